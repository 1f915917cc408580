#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r04/slide_pmc2
rm -rf $O; mkdir -p $O
SETS=${1:-"--set slide=1"}
ROWS=${2:-1048576}
cd /tmp && export TMPDIR=/tmp
run() { timeout 300 rocprofv3 --pmc $2 -d $O/pmc_$1 -o sb -- python $ROOT/tools/slide_bench.py --rows $ROWS --launches 3 $SETS > $O/pmc_$1.json 2> $O/pmc_$1.err; }
run insts "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_IFETCH"
run cycles "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run l2 "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
cd $ROOT
python tools/pmc_report.py $O eval_ > $O/report.txt 2>&1
cat $O/report.txt
