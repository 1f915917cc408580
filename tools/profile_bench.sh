#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# Summaries land in gpurun_out/prof_<tag>/; tools/summarize_profile.py condenses them for profiles/.
set -u
TAG=${1:-r1}
ARGS=${2:-"--steps 20 --warmup 3 --no-cpu"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT $OUT/trace
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $ROOT/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o bench -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_l2 -o bench -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/pmc_l2_bench.json 2> $OUT/pmc_l2.err
find $OUT -name '*.csv' | head -40
ls -la $OUT/*
