#!/bin/bash
set -u
O=gpurun_out/r04/full
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 python -m pytest tests/test_scale_parity.py -m gpu -x -q > $O/pytest_scale.log 2>&1; tail -2 $O/pytest_scale.log
timeout 900 python tools/collect_counters.py --rows 1048576 --out $O/prof_1m > $O/collect_1m.log 2>&1
timeout 600 python tools/collect_counters.py --rows 131072 --out $O/prof_131k --merge $O/prof_1m/counters.json > $O/collect_131k.log 2>&1
cp $O/prof_131k/counters.json profiles/r04_counters.json
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 300 $O/bench.json
cp profiles/r04_counters.json $O/r04_counters.json
