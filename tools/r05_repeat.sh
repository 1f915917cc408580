set -u
for i in 1 2 3; do
python bench.py --steps 40 --warmup 5 --no-cpu --no-variants --no-pipeline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run $i: 1M ms_per_step %.5f kernel %.5f  shard %.5f kernel %.5f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['weak_shard']['ms_per_step'], r['weak_shard']['roofline']['kernel_ms']))"
done
