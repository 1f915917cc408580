# planning threads with kept host threads: MP_HOST_THREADS sweep, run() of the bench's pipeline block at 131072 and 10^6 rows
set -u
mkdir -p gpurun_out/r05
(for thr in default 48 64 96 default 64; do
  if [ $thr = default ]; then unset MP_HOST_THREADS; else export MP_HOST_THREADS=$thr; fi
  echo "== MP_HOST_THREADS=$thr"
  python tools/pipeline_ab.py 131072 1048576 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); p = r['phases_ms']
    print('rows %8d run %.2f (min %.2f) plan %.2f unique %.2f build_windows %.2f finish %.2f' % (r['rows'], r['run_ms'], r['min'], p['plan'], p['unique'], p['build_windows'], p['finish']))"
done) 2>&1 | tee gpurun_out/r05/exp_threads2.txt
