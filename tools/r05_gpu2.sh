# slide kernel iteration: parity of every evaluation path, then the bench line (no CPU legs) and a kernel trace
set -u
O=gpurun_out/r05
T=${1:-it}
mkdir -p $O $O/prof_$T
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_scale_parity.py -m gpu -x -q -k "grouping or rotating or kernel_shapes or scale or slide" > $O/pytest_$T.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_$T.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-variants --no-pipeline > $O/bench_$T.json 2> $O/bench_$T.err
echo "bench rc=$?"
MP_EVAL_SLIDE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-variants --no-pipeline --rows 131072 --no-cpu > $O/bench_${T}_shard_slide.json 2> $O/bench_${T}_shard_slide.err
python - $T <<'PY'
import json, sys
T = sys.argv[1]
r = json.load(open(f"gpurun_out/r05/bench_{T}.json"))
print("ms_per_step", round(r["ms_per_step"], 5), "kernel_ms", round(r["roofline"]["kernel_ms"], 5), "shard", round(r["weak_shard"]["ms_per_step"], 5),
      "shard kernel", round(r["weak_shard"]["roofline"]["kernel_ms"], 5), "parity", r.get("parity_checked"), "proj", r.get("projected_strong_scaling", {}).get("ceiling"))
r = json.load(open(f"gpurun_out/r05/bench_{T}_shard_slide.json"))
print("shard with the sliding kernel: ms_per_step", round(r["ms_per_step"], 5), "kernel_ms", round(r["roofline"]["kernel_ms"], 5), r["roofline"]["eval_mode"])
PY
R=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$T/trace -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-variants --no-pipeline > $R/$O/prof_$T/trace_bench.json 2> $R/$O/prof_$T.err)
python tools/summarize_profile.py $O/prof_$T > $O/prof_${T}_summary.txt 2>&1 || true
grep -E "eval_|zero_kernel" $O/prof_${T}_summary.txt | head
