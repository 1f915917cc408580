# round 5, first GPU call: new tests on the HIP library, the bench line with rotating launches / variants / pipeline, A/B against the fill dispatch
set -u
O=gpurun_out/r05
mkdir -p $O $O/prof1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_validate_bwt.py tests/test_hip_parity.py tests/test_abi.py -m gpu -x -q -k "bowtie2 or rotating or abi" > $O/pytest_new.log 2>&1
echo "pytest rc=$?" >> $O/pytest_new.log
tail -5 $O/pytest_new.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench1.json 2> $O/bench1.err
echo "bench rc=$?"; tail -c 1500 $O/bench1.err
MP_BENCH_ROTATE=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-variants --no-pipeline > $O/bench1_fill.json 2> $O/bench1_fill.err
python - <<'PY'
import json
for f in ("bench1", "bench1_fill"):
    try:
        r = json.load(open(f"gpurun_out/r05/{f}.json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "ms_per_step", round(r["ms_per_step"], 5), "kernel_ms", round(r["roofline"]["kernel_ms"], 5), "shard", round(r["weak_shard"]["ms_per_step"], 5),
          "shard kernel", round(r["weak_shard"]["roofline"]["kernel_ms"], 5), "parity", r.get("parity_checked"), "proj", r.get("projected_strong_scaling", {}).get("ceiling"))
    if "variants" in r:
        print({k: (round(v["kernel_ms"], 4), v.get("parity_checked")) for k, v in r["variants"].items()})
    if "pipeline" in r:
        print({k: (round(v["run_ms"], 2), round(v["construct_ms"], 1), v["tsv_equal_oracle"], v["phases_ms"]) for k, v in r["pipeline"].items() if isinstance(v, dict)})
PY
R=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof1/trace -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-variants --no-pipeline > $R/$O/prof1/trace_bench.json 2> $R/$O/prof1.err)
python tools/summarize_profile.py $O/prof1 > $O/prof1_summary.txt 2>&1 || true
head -30 $O/prof1_summary.txt
