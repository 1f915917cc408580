# last verification of the round's tree on the GPU box: the whole GPU suite, bench.py's N > 1 control flow on the one GPU, smoke(), and the
# release sizes the call sites claim against the sizes the blocks were requested with (MP_TRACE)
set -u
O=gpurun_out/r05/final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
MP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 12 --warmup 3 --rows 65536 > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err
echo "bench 2 ranks rc=$?"; tail -c 300 $O/bench_2ranks_gloo.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
MP_TRACE=1 timeout 300 python tools/pipeline_ab.py 131072 2>&1 | grep "dev_free\|device blocks" | sort | uniq -c | head -20 > $O/release_sizes.txt; cat $O/release_sizes.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-variants --no-pipeline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench: ms_per_step %.5f shard %.5f counters_stale %s' % (r['ms_per_step'], r['weak_shard']['ms_per_step'], r['roofline'].get('counters_stale')))"
