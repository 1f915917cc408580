#!/bin/bash
# Runs on the GPU box: SQ counter passes over tools/variant_bench.py (bit-sliced eval shapes), one pass per counter group.
# usage: tools/pmc_variants.sh <tag> "<variant list>" [extra variant_bench args]
set -u
TAG=${1:-pv}
VARS=${2:-"1 0"}
EXTRA=${3:-""}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_IFETCH -d $OUT/pmc_insts -o vb -- python $ROOT/tools/variant_bench.py --mode bits --steps 3 --variants $VARS $EXTRA > $OUT/pmc_insts.json 2> $OUT/pmc_insts.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY -d $OUT/pmc_cycles -o vb -- python $ROOT/tools/variant_bench.py --mode bits --steps 3 --variants $VARS $EXTRA > $OUT/pmc_cycles.json 2> $OUT/pmc_cycles.err
timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_INST_CYCLES_SMEM SQ_IFETCH_LEVEL SQ_CYCLES -d $OUT/pmc_mem -o vb -- python $ROOT/tools/variant_bench.py --mode bits --steps 3 --variants $VARS $EXTRA > $OUT/pmc_mem.json 2> $OUT/pmc_mem.err
ls $OUT/*
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR -d $OUT/pmc_lds -o vb -- python $ROOT/tools/variant_bench.py --mode bits --steps 3 --variants $VARS $EXTRA > $OUT/pmc_lds.json 2> $OUT/pmc_lds.err
