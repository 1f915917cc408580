# quick look: the bench line without CPU legs, variants and pipeline (config 4 + the shard), the shard with the sliding kernel forced; twice
set -u
O=gpurun_out/r05
T=${1:-q}
mkdir -p $O
for i in 1 2; do
timeout 600 python bench.py --steps 40 --warmup 5 --no-variants --no-pipeline --no-cpu > $O/bench_$T.json 2> $O/bench_$T.err
MP_EVAL_SLIDE=1 timeout 600 python bench.py --steps 40 --warmup 5 --no-variants --no-pipeline --rows 131072 --no-cpu --no-shard > $O/bench_${T}_shard_slide.json 2> $O/bench_${T}_shard_slide.err
python - $T <<'PY'
import json, sys
T = sys.argv[1]
r = json.load(open(f"gpurun_out/r05/bench_{T}.json"))
s = json.load(open(f"gpurun_out/r05/bench_{T}_shard_slide.json"))
print("1M ms_per_step %.5f kernel %.5f | shard (first-pass kernel) %.5f | shard (sliding) %.5f parity %s" % (r["ms_per_step"], r["roofline"]["kernel_ms"],
      r["weak_shard"]["ms_per_step"], s["ms_per_step"], r.get("parity_checked")))
PY
done
