# histogram kernel at 10^6 rows: row slices per window (MP_HIST_SLICES) against L2 locality
set -u
for s in 0 9 16 24 32 48 64 96; do
  if [ $s = 0 ]; then unset MP_HIST_SLICES; else export MP_HIST_SLICES=$s; fi
  MP_TRACE=1 python tools/profile_run.py ${1:-1048576} 2>&1 | grep "unique: histogram" | tail -1 | sed "s/^/slices $s: /"
done
