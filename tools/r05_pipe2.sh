# pipeline iteration: parity of the core step on the HIP library (fixtures + synthetic depths), laps, and the bench's pipeline block
set -u
O=gpurun_out/r05
T=${1:-p1}
mkdir -p $O
timeout 900 python -m pytest tests/test_core_golden.py tests/test_scale_parity.py tests/test_bitsets.py tests/test_chain.py -m gpu -x -q -k "not sliding and not bench_workload" > $O/pytest_pipe_$T.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_pipe_$T.log
bash tools/r05_pipe.sh > $O/laps_$T.txt 2>&1
grep -E "^\{|histograms \|\||build_windows \(library\)|plan \+ eval|exception verdicts|unique: histogram|build_windows: exceptions|get_exceptions|set_extra" $O/laps_$T.txt
