# device block pool on / off: run() of the bench's pipeline block (context kept), then parity of the whole pipeline on the GPU
set -u
mkdir -p gpurun_out/r05
(for rep in 1 2; do
for pool in 0 1; do
  echo "== MP_DEVICE_POOL=$pool"
  MP_DEVICE_POOL=$pool python tools/pipeline_ab.py 131072 1048576 2>&1 | grep "^{" | cut -c1-420
done
done
MP_TRACE=1 python tools/pipeline_ab.py 131072 2>&1 | grep "device blocks") 2>&1 | tee gpurun_out/r05/exp_pool.txt
timeout 1200 python -m pytest tests/test_core_golden.py tests/test_scale_parity.py tests/test_bitsets.py tests/test_pairing.py -m gpu -x -q > gpurun_out/r05/pytest_pool.log 2>&1; tail -3 gpurun_out/r05/pytest_pool.log
