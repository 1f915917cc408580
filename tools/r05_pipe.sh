# the real step: laps of the library (MP_TRACE) and of the Python side (MP_TRACE_PY), second run of the process, both depths; kernel trace of the 1M run
set -u
O=gpurun_out/r05
mkdir -p $O
for rows in 131072 1048576; do
  MP_TRACE=1 MP_TRACE_PY=1 python tools/profile_run.py $rows > $O/laps_$rows.out 2> $O/laps_$rows.err
  # the laps of the LAST run() only
  python - $O/laps_$rows.err <<'PY'
import sys
lines = open(sys.argv[1]).read().splitlines()
starts = [i for i, l in enumerate(lines) if "run: start" in l]
print("\n".join(lines[starts[-1]:]))
PY
  grep -m1 "parse_s" $O/laps_$rows.out
done
