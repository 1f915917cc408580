# timing experiments (no parity): what the patch units cost in each evaluation form
set -u
O=gpurun_out/r05
mkdir -p $O
B="python bench.py --steps 40 --warmup 5 --no-cpu --no-variants --no-pipeline"
for tag in base nopatch; do
  if [ $tag = nopatch ]; then export MP_EXPERIMENT_SKIP_PATCH=1; fi
  $B > $O/exp_${tag}.json 2>/dev/null
  MP_EVAL_SLIDE=1 $B --rows 131072 > $O/exp_${tag}_shard_slide.json 2>/dev/null
  MP_EVAL_SLIDE=1 MP_SLIDE_GW=1 $B --rows 131072 > $O/exp_${tag}_shard_slide_gw1.json 2>/dev/null
done
python - <<'PY'
import json
for tag in ("base", "nopatch"):
    r = json.load(open(f"gpurun_out/r05/exp_{tag}.json"))
    s = json.load(open(f"gpurun_out/r05/exp_{tag}_shard_slide.json"))
    g = json.load(open(f"gpurun_out/r05/exp_{tag}_shard_slide_gw1.json"))
    print(tag, "1M", round(r["ms_per_step"], 5), "shard(chain)", round(r["weak_shard"]["ms_per_step"], 5), "shard(slide)", round(s["ms_per_step"], 5), "shard(slide gw1)", round(g["ms_per_step"], 5))
PY
