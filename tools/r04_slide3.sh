#!/bin/bash
set -u
O=gpurun_out/r04/slide3
mkdir -p $O
for R in 262144 393216 524288 786432; do
  timeout 300 python tools/slide_bench.py --rows $R --set slide=0 --set slide=1 >> $O/sizes.txt 2>> $O/sizes.err
done
cat $O/sizes.txt
timeout 900 python tools/collect_counters.py --rows 1048576 --out $O/prof_1m > $O/collect_1m.log 2>&1
tail -40 $O/collect_1m.log
cp $O/prof_1m/counters.json profiles/r04_counters.json
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -c 600 $O/bench.json
cp profiles/r04_counters.json $O/r04_counters.json
