#!/bin/bash
set -u
O=gpurun_out/r04/slide5
mkdir -p $O
bash tools/r04_trace.sh 2>&1 | grep -E "slide|chain|zero"
cat gpurun_out/r04/trace/run.txt
MP_EVAL_SLIDE=1 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_oracle_golden.py tests/test_core_golden.py tests/test_scale_parity.py tests/test_window_stats.py -m gpu -x -q > $O/pytest_forced.log 2>&1
echo "forced-slide pytest rc=$?"; tail -3 $O/pytest_forced.log
