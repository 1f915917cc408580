set -u
O=gpurun_out/r05
mkdir -p $O
timeout 1200 python -m pytest tests/test_comm_ranks.py tests/test_multirank.py tests/test_abi.py -m gpu -x -q > $O/pytest_comm.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_comm.log
# bench.py's N > 1 control flow on the one GPU (collectives through the host: RCCL refuses two ranks on one device)
MP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 12 --warmup 3 --rows 65536 > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err
echo "bench 2 ranks rc=$?"; tail -c 400 $O/bench_2ranks_gloo.json; tail -3 $O/bench_2ranks_gloo.err
