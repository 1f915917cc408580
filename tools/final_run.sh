set -u
O=gpurun_out/r02/final
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python tools/collect_counters.py --out $O/prof_bench > $O/collect_131k.log 2>&1
python tools/collect_counters.py --rows 1048576 --out $O/prof_bench_1m > $O/collect_1m.log 2>&1
python bench.py --rows 1048576 > $O/bench_1m.json 2> $O/bench_1m.err
python tools/pipeline_times.py > $O/pipeline_times.txt 2>&1
python tools/pipeline_scale.py > $O/scale_131k.txt 2>&1
python tools/pipeline_scale.py --rows 1048576 > $O/scale_1m.txt 2>&1
python tools/pipeline_scale.py --rows 1048576 --file > $O/scale_1m_file.txt 2>&1
python tools/profile_pipeline.py --out $O/prof_pipe > $O/pipeline_kernels.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_dimer -o dimer -- python $GRAFT_REPO_ROOT/tools/dimer_bench.py > $GRAFT_REPO_ROOT/$O/dimer_bench.txt 2> $GRAFT_REPO_ROOT/$O/dimer_bench.err
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_pcr -o pcr -- python $GRAFT_REPO_ROOT/tools/pcr_bench.py > $GRAFT_REPO_ROOT/$O/pcr_bench.txt 2> $GRAFT_REPO_ROOT/$O/pcr_bench.err
cd $GRAFT_REPO_ROOT
tail -c 300 $O/bench.json; echo; tail -n 2 $O/scale_131k.txt | cut -c1-600; tail -n 1 $O/scale_1m.txt | cut -c1-600; tail -n 3 $O/pipeline_times.txt | cut -c1-300
