# timing experiment (no parity): the sliding kernel with every plane fetch pointed at ONE plane row (cache hits only) — what is left is issue + latency of hits
set -u
for rows in 131072 1048576; do
  B="python bench.py --steps 40 --warmup 5 --no-cpu --no-variants --no-pipeline --no-shard --rows $rows"
  for one in 0 1; do
    if [ $one = 1 ]; then export MP_EXPERIMENT_ONE_ROW=1 MP_EXPERIMENT_SKIP_PATCH=1; else unset MP_EXPERIMENT_ONE_ROW; export MP_EXPERIMENT_SKIP_PATCH=1; fi
    MP_EVAL_SLIDE=1 $B 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows $rows one_row $one (no patch units) ms_per_step %.5f kernel_ms %.5f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
  done
done | tee gpurun_out/r05/exp4_onerow.txt
