# round-end check on the GPU box: full GPU suite, smoke, whole-step timings, batch mode, soaks
set -u
O=gpurun_out/r03/check
mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/smoke.log 2>&1
python tools/pipeline_times.py > $O/pipeline_times.txt 2>&1
python tools/pipeline_scale.py > $O/scale_131k.txt 2>&1
python tools/pipeline_scale.py --rows 1048576 > $O/scale_1m.txt 2>&1
python tools/soak_parity.py --seconds 150 > $O/soak_parity.txt 2>&1
python tools/soak_primers.py --seconds 60 > $O/soak_primers.txt 2>&1
tail -n 3 $O/pytest_gpu.log; tail -n 1 $O/smoke.log; tail -n 4 $O/pipeline_times.txt | cut -c1-300; tail -n 1 $O/scale_131k.txt | cut -c1-500; tail -n 1 $O/scale_1m.txt | cut -c1-500; tail -n 2 $O/soak_parity.txt | cut -c1-400; tail -n 1 $O/soak_primers.txt | cut -c1-300
