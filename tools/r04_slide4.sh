#!/bin/bash
set -u
O=gpurun_out/r04/slide4
mkdir -p $O
timeout 600 python tools/slide_bench.py --rows 1048576 --set slide=0 --set slide=1 --set slide=1,band=30 > $O/cfg4.txt 2> $O/cfg4.err
cat $O/cfg4.txt; tail -3 $O/cfg4.err
MP_EVAL_SLIDE=1 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_oracle_golden.py tests/test_core_golden.py tests/test_scale_parity.py tests/test_window_stats.py -m gpu -x -q > $O/pytest_forced.log 2>&1
echo "forced-slide pytest rc=$?"; tail -5 $O/pytest_forced.log
