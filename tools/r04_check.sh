#!/bin/bash
# round-4 check on the GPU box: the whole GPU suite (new: test_comm_ranks, test_chain), then the bench line in its new form
set -u
O=gpurun_out/r04/check
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
tail -c 1500 $O/bench.json; tail -5 $O/bench.err
