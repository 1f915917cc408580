# timing experiments (no parity): words per lane of the sliding kernel at both sizes
set -u
O=gpurun_out/r05
mkdir -p $O
for rows in 131072 1048576; do
  B="python bench.py --steps 40 --warmup 5 --no-cpu --no-variants --no-pipeline --no-shard --rows $rows"
  for gw in 1 2 4; do
    for band in 0 $1; do
      if [ $band = 0 ]; then unset MP_SLIDE_BAND; else export MP_SLIDE_BAND=$band; fi
      MP_EVAL_SLIDE=1 MP_SLIDE_GW=$gw $B 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows $rows gw $gw band $band ms_per_step %.5f kernel_ms %.5f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
    done
  done
done | tee $O/exp3_gw.txt
