#!/bin/bash
# the counter passes and the bench line again (a hashed source changed after tools/r05_final.sh ran: common.hpp, host side only)
set -u
O=gpurun_out/r05/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python tools/collect_counters.py --rows 1048576 --out $O/prof_1m > $O/collect_1m.log 2>&1
timeout 600 python tools/collect_counters.py --rows 131072 --out $O/prof_131k --merge $O/prof_1m/counters.json > $O/collect_131k.log 2>&1
cp $O/prof_131k/counters.json profiles/r05_counters.json
cp $O/prof_131k/counters.json $O/r05_counters.json
cp $O/prof_1m/summary.txt $O/bench_eval_1m.txt 2>/dev/null; cp $O/prof_131k/summary.txt $O/bench_eval.txt 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; python - <<'PY'
import json
r = json.load(open("gpurun_out/r05/final/bench.json"))
print("value %.4g ms_per_step %.5f frac %.3f hbm_frac %.3f stale %s shard %.5f pipeline %.2f / %.2f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"].get("hbm_frac") or -1,
      r["roofline"].get("counters_stale"), r["weak_shard"]["ms_per_step"], r["pipeline"]["rows_131072"]["run_ms"], r["pipeline"]["rows_1048576"]["run_ms"]))
PY
