#!/bin/bash
# round-5 measurement on the GPU box: counters of config 4 and of the shard keyed by kernel source hash (separate rocprofv3 --pmc passes),
# the bench line that reads them, kernels of the whole core step, the sliding kernel's phase stamps, config-5 chain check, soaks
set -u
O=gpurun_out/r05/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python tools/collect_counters.py --rows 1048576 --out $O/prof_1m > $O/collect_1m.log 2>&1
timeout 600 python tools/collect_counters.py --rows 131072 --out $O/prof_131k --merge $O/prof_1m/counters.json > $O/collect_131k.log 2>&1
cp $O/prof_131k/counters.json profiles/r05_counters.json
cp $O/prof_131k/counters.json $O/r05_counters.json
cp $O/prof_1m/summary.txt $O/bench_eval_1m.txt 2>/dev/null; cp $O/prof_131k/summary.txt $O/bench_eval.txt 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 300 $O/bench.json
timeout 600 python tools/profile_pipeline.py --out $O/prof_pipe > $O/pipeline_kernels.txt 2>&1; head -16 $O/pipeline_kernels.txt
timeout 600 python tools/pipeline_times.py > $O/pipeline_times.jsonl 2> $O/pipeline_times.err
(python tools/slide_stamps.py --rows 131072 --slices 8; python tools/slide_stamps.py --rows 1048576 --slices 64) > $O/slide_stamps.txt 2>&1
(echo "# NN_degenerate.run(), k=18 (tools/profile_run.py: second run of the process): stats in ms, then the laps of the Python side (MP_TRACE_PY) and of the library (MP_TRACE)"
 for rows in 131072 1048576; do for i in 1 2; do python tools/profile_run.py $rows 2>&1 | grep -m1 "^{"; done; done
 echo "# MP_DEVICE_GATE=0 (every window to the host)"; MP_DEVICE_GATE=0 python tools/profile_run.py 131072 2>&1 | grep -m1 "^{"
 echo "# laps at 131072"; MP_TRACE_PY=1 MP_TRACE=1 python tools/profile_run.py 131072 2>&1 | grep "^\[core\]\|^\[mprime\]" | tail -44
 echo "# laps at 1048576"; MP_TRACE_PY=1 MP_TRACE=1 python tools/profile_run.py 1048576 2>&1 | grep "^\[core\]\|^\[mprime\]" | tail -44) > $O/run_laps.txt 2>&1
timeout 900 python tools/multi_cluster.py --clusters 16 --max-rows 5000 --check > $O/config5_check.json 2> $O/config5_check.err; echo "config5 check rc=$?"; tail -c 300 $O/config5_check.json
timeout 300 python tools/soak_parity.py --seconds 120 > $O/soak.txt 2>&1; tail -2 $O/soak.txt
timeout 200 python tools/soak_primers.py --seconds 45 >> $O/soak.txt 2>&1; tail -1 $O/soak.txt
timeout 600 python tools/batch_bench.py --clusters 256 --rows 500 --no-per-cluster --workers 4,3x4 > $O/batch256.txt 2>&1; tail -c 400 $O/batch256.txt
