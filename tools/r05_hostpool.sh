# kept host threads on / off (MP_HOST_POOL): run() of the bench's pipeline block, two rounds on one box
set -u
mkdir -p gpurun_out/r05
(for rep in 1 2; do
for pool in 0 1; do
  echo "== MP_HOST_POOL=$pool"
  MP_HOST_POOL=$pool python tools/pipeline_ab.py 131072 1048576 2>&1 | grep "^{" | cut -c1-420
done
done) 2>&1 | tee gpurun_out/r05/exp_hostpool.txt
