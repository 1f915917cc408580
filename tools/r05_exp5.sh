set -u
B="python bench.py --steps 40 --warmup 5 --no-cpu --no-variants --no-pipeline --no-shard"
for band in 0 80 64 48; do
  if [ $band = 0 ]; then unset MP_SLIDE_BAND; else export MP_SLIDE_BAND=$band; fi
  for i in 1 2; do
  $B 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('band $band run $i: 1M ms_per_step %.5f kernel %.5f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
  done
done
