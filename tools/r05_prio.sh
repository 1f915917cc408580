# issue-priority experiment: product build (progress priority everywhere, patch waves at priority 3) against variant builds
set -u
mkdir -p gpurun_out/r05
for v in product pp0 minwin minwin_pp0 noprio; do
  if [ $v = product ]; then unset MPRIME_LIBRARY; else export MPRIME_LIBRARY=$PWD/tools/_build/libmprime_hip_$v.so; fi
  echo "== $v"
  bash tools/r05_gpu3.sh prio_$v
done 2>&1 | tee gpurun_out/r05/exp_prio.txt
