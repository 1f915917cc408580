#!/bin/bash
# round-4 verification on the GPU box: whole GPU suite, soak, counters of config 4 and of the shard, bench line, pipeline kernels
set -u
O=gpurun_out/r04/full
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 900 python tools/collect_counters.py --rows 1048576 --out $O/prof_1m > $O/collect_1m.log 2>&1
timeout 600 python tools/collect_counters.py --rows 131072 --out $O/prof_131k --merge $O/prof_1m/counters.json > $O/collect_131k.log 2>&1
cp $O/prof_131k/counters.json profiles/r04_counters.json
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 300 $O/bench.json
cp profiles/r04_counters.json $O/r04_counters.json
timeout 600 python tools/soak_parity.py --seconds 120 > $O/soak.txt 2>&1; tail -3 $O/soak.txt
