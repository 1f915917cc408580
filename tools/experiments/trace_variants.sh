#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of tools/variant_bench.py for a list of bit-sliced shapes.
# usage: tools/trace_variants.sh <tag> "<variant list>" [extra variant_bench args]
set -u
TAG=${1:-tv}
VARS=${2:-"c0"}
EXTRA=${3:-""}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT/trace
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o vb -- python $ROOT/tools/variant_bench.py --mode bits --steps 20 --variants $VARS $EXTRA > $OUT/trace.json 2> $OUT/trace.err
ls $OUT/trace
