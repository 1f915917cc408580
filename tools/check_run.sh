# round-end check on the GPU box: full GPU suite, smoke, whole-step timings, dimer / PCR kernels under rocprofv3
set -u
O=gpurun_out/r02/check
mkdir -p $O
python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/smoke.log 2>&1
python tools/pipeline_times.py > $O/pipeline_times.txt 2>&1
python tools/multi_cluster.py > $O/multi_cluster.txt 2>&1
python tools/pipeline_scale.py > $O/scale_131k.txt 2>&1
python tools/pipeline_scale.py --rows 1048576 > $O/scale_1m.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_dimer -o dimer -- python $GRAFT_REPO_ROOT/tools/dimer_bench.py > $GRAFT_REPO_ROOT/$O/dimer_bench.txt 2> $GRAFT_REPO_ROOT/$O/dimer_bench.err
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_pcr -o pcr -- python $GRAFT_REPO_ROOT/tools/pcr_bench.py > $GRAFT_REPO_ROOT/$O/pcr_bench.txt 2> $GRAFT_REPO_ROOT/$O/pcr_bench.err
cd $GRAFT_REPO_ROOT
tail -n 3 $O/pytest_gpu.log; tail -n 1 $O/smoke.log; tail -n 4 $O/pipeline_times.txt | cut -c1-300; tail -n 3 $O/multi_cluster.txt | cut -c1-400; tail -n 1 $O/scale_131k.txt | cut -c1-500; tail -n 1 $O/scale_1m.txt | cut -c1-500
