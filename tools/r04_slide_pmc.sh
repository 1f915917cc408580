#!/bin/bash
# SQ counters of the sliding kernel vs the first-pass kernel on config 4 (one rocprofv3 pass per counter group)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04/slide_pmc
mkdir -p $O
SETS=${1:-"--set slide=0 --set slide=1,gw=2,band=32"}
ROWS=${2:-1048576}
cd /tmp && export TMPDIR=/tmp
run() { timeout 300 rocprofv3 --pmc $2 -d $O/pmc_$1 -o sb -- python $ROOT/tools/slide_bench.py --rows $ROWS --launches 3 $SETS > $O/pmc_$1.json 2> $O/pmc_$1.err; }
run insts "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_IFETCH"
run cycles "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY"
run mem "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_INST_CYCLES_SMEM SQ_IFETCH_LEVEL SQ_CYCLES"
run lds "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR"
cd $ROOT
python tools/pmc_report.py $O eval_ > $O/report.txt 2>&1
cat $O/report.txt | head -120
