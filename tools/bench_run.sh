# headline measurement on the GPU box: counters (separate rocprofv3 --pmc passes) for both shard sizes, merged into
# profiles/r02_counters.json, then the two bench lines that read them
set -u
O=gpurun_out/r02/bench
mkdir -p $O
python tools/collect_counters.py --out $O/prof_bench > $O/collect_131k.log 2>&1
python tools/collect_counters.py --rows 1048576 --out $O/prof_bench_1m > $O/collect_1m.log 2>&1
python - <<'PY'
import json
a = json.load(open("gpurun_out/r02/bench/prof_bench/counters.json"))["entries"]
b = json.load(open("gpurun_out/r02/bench/prof_bench_1m/counters.json"))["entries"]
old = json.load(open("profiles/r02_counters.json"))
old["entries"] = a + b
json.dump(old, open("profiles/r02_counters.json", "w"), indent=1)
json.dump(old, open("gpurun_out/r02/bench/r02_counters.json", "w"), indent=1)
PY
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --rows 1048576 > $O/bench_1m.json 2> $O/bench_1m.err
cp $O/prof_bench/summary.txt $O/bench_eval.txt 2>/dev/null
cp $O/prof_bench_1m/summary.txt $O/bench_eval_1m.txt 2>/dev/null
tail -c 600 $O/bench.json; echo; tail -c 300 $O/bench_1m.json
