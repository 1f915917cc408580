# timing experiments (no parity): band length of the sliding kernel at the 131072-row shard, with and without the patch units
set -u
O=gpurun_out/r05
mkdir -p $O
B="python bench.py --steps 40 --warmup 5 --no-cpu --no-variants --no-pipeline --rows 131072"
for band in 4 6 8 12 16 24 32; do
  for skip in 0 1; do
    if [ $skip = 1 ]; then export MP_EXPERIMENT_SKIP_PATCH=1; else unset MP_EXPERIMENT_SKIP_PATCH; fi
    MP_EVAL_SLIDE=1 MP_SLIDE_BAND=$band $B 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('band $band skip_patch $skip ms_per_step %.5f kernel_ms %.5f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
  done
done | tee $O/exp2_bands.txt
