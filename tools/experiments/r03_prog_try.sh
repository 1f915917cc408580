#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03
mkdir -p $O
VARS=${1:-"c7 p7 p6 p8 p3 p4"}
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "grouping_paths or kernel_shapes" > $O/prog_pytest.log 2>&1
tail -3 $O/prog_pytest.log
timeout 300 python tools/variant_bench.py --mode bits --variants $VARS > $O/prog_variants_131k.txt 2>&1
cat $O/prog_variants_131k.txt
timeout 600 python tools/variant_bench.py --mode bits --rows 1048576 --variants $VARS > $O/prog_variants_1m.txt 2>&1
cat $O/prog_variants_1m.txt
