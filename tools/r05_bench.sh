set -u
O=gpurun_out/r05
T=${1:-b}
mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_$T.json 2> $O/bench_$T.err
echo "bench rc=$?"; tail -c 600 $O/bench_$T.err
python - $T <<'PY'
import json, sys
r = json.load(open(f"gpurun_out/r05/bench_{sys.argv[1]}.json"))
print("ms_per_step", round(r["ms_per_step"], 5), "kernel_ms", round(r["roofline"]["kernel_ms"], 5), "frac", round(r["roofline"]["frac"], 4), "shard", round(r["weak_shard"]["ms_per_step"], 5),
      "parity", r.get("parity_checked"), "proj", round(r["projected_strong_scaling"]["ceiling"], 3))
print({k: (round(v["kernel_ms"], 4), v.get("parity_checked")) for k, v in r["variants"].items()})
for k, v in r["pipeline"].items():
    if isinstance(v, dict):
        print(k, "run_ms", round(v["run_ms"], 2), "min", round(v["run_ms_min"], 2), "construct", round(v["construct_ms"], 1), "tsv == oracle", v["tsv_equal_oracle"], v["phases_ms"])
PY
