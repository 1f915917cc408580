#!/bin/bash
set -u
O=gpurun_out/r04/sizes
mkdir -p $O; rm -f $O/sizes.txt
for R in 131072 196608 262144 327680 393216 524288 786432 1048576; do
  timeout 300 python tools/slide_bench.py --rows $R --set slide=0 --set slide=1 --set slide=1,gw=1 >> $O/sizes.txt 2>> $O/sizes.err
done
cut -c1-140 $O/sizes.txt
