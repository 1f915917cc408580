#!/bin/bash
set -u
O=gpurun_out/r04/slide1
mkdir -p $O
timeout 600 python tools/slide_bench.py --rows 1048576 --set slide=0 --set slide=1,gw=2,band=16 --set slide=1,gw=2,band=32 --set slide=1,gw=2,band=8 --set slide=1,gw=1,band=16 --set slide=1,gw=4,band=16 --set slide=1,gw=1,band=32 > $O/cfg4.txt 2> $O/cfg4.err
cat $O/cfg4.txt; tail -3 $O/cfg4.err
timeout 300 python tools/slide_bench.py --rows 131072 --set slide=0 --set slide=1,gw=1,band=8 --set slide=1,gw=1,band=4 --set slide=1,gw=2,band=8 --set slide=1,gw=2,band=4 --set slide=1,gw=1,band=16 > $O/shard.txt 2> $O/shard.err
cat $O/shard.txt; tail -3 $O/shard.err
