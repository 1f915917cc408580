#!/bin/bash
# GPU run of the LDS-tiled evaluation: A/B on the bench workload, kernel trace and counters of the variants
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O
VARS=${1:-"c7 t4 t2"}
PMC=${2:-""}
timeout 300 python tools/variant_bench.py --mode bits --variants $VARS > $O/tile_variants_131k.txt 2>&1
cat $O/tile_variants_131k.txt
timeout 600 python tools/variant_bench.py --mode bits --rows 1048576 --variants $VARS > $O/tile_variants_1m.txt 2>&1
cat $O/tile_variants_1m.txt
bash tools/trace_variants.sh r03tile "$VARS" > /dev/null 2>&1
python - <<'PY'
import glob, sqlite3
for f in glob.glob("gpurun_out/prof_r03tile/trace/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for row in db.execute("select name, grid_x, workgroup_x, lds_size, count(*), avg(duration), min(duration), max(duration) from kernels where name like '%eval_%' group by name, grid_x order by name"):
        print(row[0][28:70], row[1:5], [round(x / 1e3, 2) for x in row[5:]])
PY
if [ -n "$PMC" ]; then
  bash tools/pmc_variants.sh r03tile "$PMC" > /dev/null 2>&1
  python tools/pmc_report.py gpurun_out/prof_r03tile > $O/tile_pmc.txt
  cat $O/tile_pmc.txt
fi
