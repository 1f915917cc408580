set -u
for thr in 0 48 64 96 128 192; do
  if [ $thr = 0 ]; then unset MP_HOST_THREADS; else export MP_HOST_THREADS=$thr; fi
  for rows in 131072 1048576; do
    MP_TRACE=1 python tools/profile_run.py $rows 2>&1 | grep -E "planning done at|^\{" | tail -2 | tr '\n' ' ' | sed "s/^/threads $thr rows $rows: /"; echo
  done
done
