# round-end measurement on the GPU box: bench line, counters of both sizes keyed by kernel source hash, pipeline kernel profile
set -u
O=gpurun_out/r03/final
mkdir -p $O
python tools/collect_counters.py --rows 131072 --out $O/prof_bench > $O/collect_131k.log 2>&1
python tools/collect_counters.py --rows 1048576 --out $O/prof_bench_1m --merge $O/prof_bench/counters.json > $O/collect_1m.log 2>&1
cp $O/prof_bench_1m/counters.json profiles/r03_counters.json
python bench.py > $O/bench.json 2> $O/bench.err
python tools/profile_pipeline.py --out $O/prof_pipe > $O/pipeline_kernels.txt 2>&1
cp profiles/r03_counters.json $O/r03_counters.json
tail -c 400 $O/bench.json; echo; head -12 $O/pipeline_kernels.txt
