#!/bin/bash
# round-end measurement on the GPU box: GPU suite, counters of config 4 and of the shard keyed by kernel source hash, the bench line,
# pipeline kernels, whole-program times, config-5 chain check, soaks
set -u
O=gpurun_out/r04/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 900 python tools/collect_counters.py --rows 1048576 --out $O/prof_1m > $O/collect_1m.log 2>&1
timeout 600 python tools/collect_counters.py --rows 131072 --out $O/prof_131k --merge $O/prof_1m/counters.json > $O/collect_131k.log 2>&1
cp $O/prof_131k/counters.json profiles/r04_counters.json
cp $O/prof_1m/summary.txt $O/bench_eval_1m.txt 2>/dev/null; cp $O/prof_131k/summary.txt $O/bench_eval.txt 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 300 $O/bench.json
cp profiles/r04_counters.json $O/r04_counters.json
timeout 600 python tools/profile_pipeline.py --out $O/prof_pipe > $O/pipeline_kernels.txt 2>&1; head -14 $O/pipeline_kernels.txt
timeout 600 python tools/pipeline_times.py > $O/pipeline_times.jsonl 2> $O/pipeline_times.err
timeout 300 python tools/pipeline_scale.py --rows 131072 > $O/pipeline_scale_131k.json 2>&1
timeout 300 python tools/pipeline_scale.py --rows 1048576 > $O/pipeline_scale_1m.json 2>&1
timeout 900 python tools/multi_cluster.py --clusters 16 --max-rows 5000 --check > $O/config5_check.json 2> $O/config5_check.err; echo "config5 check rc=$?"; tail -c 400 $O/config5_check.json
timeout 300 python tools/soak_parity.py --seconds 150 > $O/soak.txt 2>&1; tail -2 $O/soak.txt
timeout 200 python tools/soak_primers.py --seconds 60 >> $O/soak.txt 2>&1; tail -1 $O/soak.txt
# the real step's host side: run() laps at 131072 x 1000, transfer micro-benchmark, batch throughput
(echo "# NN_degenerate.run() at 131072 x 1000, k=18 (tools/profile_run.py: second run of the process, fresh context): stats in ms, then the laps of the Python side (MP_TRACE_PY) and of the library (MP_TRACE)"
 for i in 1 2 3; do python tools/profile_run.py 131072 2>&1 | head -1; done
 echo "# MP_PLAN_STREAM=0 (blocking read-back, then planning)"; MP_PLAN_STREAM=0 python tools/profile_run.py 131072 2>&1 | head -1
 echo "# MP_NO_PIN=1 MP_NO_PREFAULT=1 MP_PLAN_STREAM=0 (the round-3 transfer path)"; MP_NO_PIN=1 MP_NO_PREFAULT=1 MP_PLAN_STREAM=0 python tools/profile_run.py 131072 2>&1 | head -1
 echo "# laps"; MP_TRACE_PY=1 MP_TRACE=1 python tools/profile_run.py 131072 2>&1 | grep "^\[core\]\|^\[mprime\]" | tail -48) > $O/run_laps.txt 2>&1
mkdir -p tools/_build && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -pthread -Wno-unused-value tools/ubench/d2h_bench.hip -o tools/_build/d2h_big 2> /dev/null
(echo "# tools/ubench/d2h_bench.hip on the MI355X box: 58 MB device -> host into memory of different kinds"; ./tools/_build/d2h_big) > $O/d2h_bench.txt 2>&1
timeout 600 python tools/batch_bench.py --clusters 64 --rows 500 --workers 4,2x2,2x4 > $O/batch64.txt 2>&1
timeout 600 python tools/batch_bench.py --clusters 256 --rows 500 --no-per-cluster --workers 4,2x4,3x4 > $O/batch256.txt 2>&1; tail -c 500 $O/batch256.txt
