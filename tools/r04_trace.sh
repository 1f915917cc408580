#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04/trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -o sb -- python $ROOT/tools/slide_bench.py --rows ${1:-1048576} --launches 20 --set slide=1 > $O/run.txt 2> $O/run.err
cd $ROOT
python - <<'PY'
import glob, sqlite3
f = glob.glob("gpurun_out/r04/trace/t/**/*.db", recursive=True)[0]
db = sqlite3.connect(f)
for row in db.execute("select name, count(*), avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3 from kernels group by name order by sum(duration) desc limit 12"):
    print("%-90s n=%4d avg %.1f us  min %.1f  max %.1f" % (row[0][:90], row[1], row[2], row[3], row[4]))
PY
