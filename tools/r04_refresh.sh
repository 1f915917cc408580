#!/bin/bash
# after a change to a hashed kernel source late in the round: counters of both sizes under the new source hash, the bench line,
# the pipeline kernels and the run() laps (the rest of tools/r04_final.sh does not depend on the hash)
set -u
O=gpurun_out/r04/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 python tools/collect_counters.py --rows 1048576 --out $O/prof_1m > $O/collect_1m.log 2>&1
timeout 400 python tools/collect_counters.py --rows 131072 --out $O/prof_131k --merge $O/prof_1m/counters.json > $O/collect_131k.log 2>&1
cp $O/prof_131k/counters.json profiles/r04_counters.json; cp $O/prof_131k/counters.json $O/r04_counters.json
cp $O/prof_1m/summary.txt $O/bench_eval_1m.txt 2>/dev/null; cp $O/prof_131k/summary.txt $O/bench_eval.txt 2>/dev/null
timeout 400 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 200 $O/bench.json
timeout 300 python tools/profile_pipeline.py --out $O/prof_pipe > $O/pipeline_kernels.txt 2>&1; head -8 $O/pipeline_kernels.txt | cut -c1-160
(echo "# NN_degenerate.run() at 131072 x 1000, k=18 (tools/profile_run.py: second run of the process, fresh context): stats in ms, then the laps of the Python side (MP_TRACE_PY) and of the library (MP_TRACE)"
 for i in 1 2 3; do python tools/profile_run.py 131072 2>&1 | head -1; done
 echo "# MP_PLAN_STREAM=0 (blocking read-back, then planning)"; MP_PLAN_STREAM=0 python tools/profile_run.py 131072 2>&1 | head -1
 echo "# MP_NO_PIN=1 MP_NO_PREFAULT=1 MP_PLAN_STREAM=0 (the round-3 transfer path)"; MP_NO_PIN=1 MP_NO_PREFAULT=1 MP_PLAN_STREAM=0 python tools/profile_run.py 131072 2>&1 | head -1
 echo "# laps"; MP_TRACE_PY=1 MP_TRACE=1 python tools/profile_run.py 131072 2>&1 | grep "^\[core\]\|^\[mprime\]" | tail -48) > $O/run_laps.txt 2>&1
