# A/B on one box: variant libraries built by tools/build_variant.sh (names as arguments), twice each, interleaved
set -u
mkdir -p gpurun_out/r05
for round in 1 2; do
for v in "$@"; do
  export MPRIME_LIBRARY=$PWD/tools/_build/libmprime_hip_$v.so
  timeout 600 python bench.py --steps 40 --warmup 5 --no-variants --no-pipeline --no-cpu --no-shard > gpurun_out/r05/ab_$v.json 2> gpurun_out/r05/ab_$v.err
  MP_EVAL_SLIDE=1 timeout 600 python bench.py --steps 40 --warmup 5 --no-variants --no-pipeline --rows 131072 --no-cpu --no-shard > gpurun_out/r05/ab_${v}_shard.json 2> gpurun_out/r05/ab_${v}_shard.err
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
r = json.load(open(f"gpurun_out/r05/ab_{v}.json")); s = json.load(open(f"gpurun_out/r05/ab_{v}_shard.json"))
print("%-12s 1M ms_per_step %.5f kernel %.5f | shard (sliding) %.5f" % (v, r["ms_per_step"], r["roofline"]["kernel_ms"], s["ms_per_step"]))
PY
done
done 2>&1 | tee gpurun_out/r05/ab_$1.txt
