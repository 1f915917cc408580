#!/bin/bash
# One entry point for every measurement that ends up under profiles/ (round 6 on; the per-round r04_*/r05_* command logs are gone).
#   tools/reproduce.sh <target> [args]      — run ON the GPU box (through gpurun), writes under gpurun_out/r06/<target>/
# targets: see the case statement; each names the profiles/ file(s) it produces.
set -u
T=${1:?target}; shift || true
O=gpurun_out/r06/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
laps() {   # laps <rows> <k>: stats line + library / Python laps of the second run() of a process
  MP_TRACE_PY=1 MP_TRACE=1 python tools/profile_run.py $1 $2 2>&1 | grep "^{\|^\[core\]\|^\[mprime\]" | tail -48
}
case $T in
hist_tests)      # the histogram paths + the suites that go through them
  timeout 1500 python -m pytest tests/test_hist_paths.py tests/test_hip_parity.py tests/test_core_golden.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt ;;
hist_ab)         # profiles/r06_hist_folds.txt: wave-level folding rounds of hist_kernel, k = 18 / 22, both depths
  for rows in 131072 1048576; do for k in 18 22; do for f in 1 2 3; do
    echo "# rows $rows k $k MP_HIST_FOLDS=$f"
    MP_HIST_FOLDS=$f MP_TRACE=1 python tools/profile_run.py $rows $k 2>&1 | grep "unique:\|^{" | tail -8
  done; done; done 2>&1 | tee $O/hist_folds.txt ;;
run_laps)        # profiles/r06_run_laps.txt
  (for rows in 131072 1048576; do for k in 18 22; do echo "# rows $rows k $k"; laps $rows $k; done; done) 2>&1 | tee $O/run_laps.txt ;;
hist_where)      # where hist_kernel's time goes: phase stamps of every workgroup, kernel trace of the core step, LDS / wait counters
  for rows in 131072 1048576; do
    echo "# rows $rows: MP_HIST_PROF stamps"
    MP_HIST_PROF=$O/stamps_$rows.bin python tools/pipeline_scale.py --rows $rows > /dev/null 2>&1; python tools/hist_prof.py $O/stamps_$rows.bin
    echo "# rows $rows: rocprofv3 --kernel-trace --stats of the core step"
    rocprofv3 --kernel-trace --stats -d $O/prof_$rows/trace -o t -- python tools/pipeline_scale.py --rows $rows > /dev/null 2>&1
    python tools/summarize_profile.py $O/prof_$rows 2>/dev/null | head -30
    echo "# rows $rows: counters of hist_kernel"
    python tools/pmc_kernel.py --kernel hist_kernel --rows $rows --out $O/pmc_$rows
  done 2>&1 | tee $O/hist_where.txt ;;
hist_variants)   # A/B of compile-time knobs of unique.hip: tools/build_variant.sh <tag> unique.hip -D... beforehand, tags as arguments ("product" = the shipped build)
  for round in 1 2; do for v in "$@"; do for rows in 131072 1048576; do
    if [ $v = product ]; then unset MPRIME_LIBRARY MP_HOST_LIB; else export MPRIME_LIBRARY=$PWD/tools/_build/libmprime_hip_$v.so MP_HOST_LIB=$PWD/tools/_build/libmprime_hip_$v.so; fi
    echo "# $v rows $rows"; MP_TRACE=1 python tools/profile_run.py $rows 18 2>&1 | grep "histogram + sums\|^{" | tail -2
  done; done; done 2>&1 | tee $O/hist_variants.txt ;;
hist_v1v2)       # hist2_kernel (default) against hist_kernel (MP_HIST_V1=1), k = 18 / 22, both depths
  for round in 1 2; do for rows in 131072 1048576; do for k in 18 22; do for v1 in "" 1; do
    echo "# rows $rows k $k MP_HIST_V1=$v1"; MP_HIST_V1=$v1 MP_TRACE=1 python tools/profile_run.py $rows $k 2>&1 | grep "histogram + sums\|^{" | tail -2
  done; done; done; done 2>&1 | tee $O/hist_v1v2.txt ;;
hist_exp)        # experiment builds of unique.hip (results are WRONG: timing only): kernel time of hist2_kernel under rocprofv3, tags as arguments
  for v in "$@"; do for rows in 131072 1048576; do
    if [ $v = product ]; then unset MPRIME_LIBRARY MP_HOST_LIB; else export MPRIME_LIBRARY=$PWD/tools/_build/libmprime_hip_$v.so MP_HOST_LIB=$PWD/tools/_build/libmprime_hip_$v.so; fi
    rm -rf $O/t_${v}_$rows; rocprofv3 --kernel-trace --stats -d $O/t_${v}_$rows/trace -o t -- python tools/hist_only.py $rows 18 > /dev/null 2>&1
    echo "# $v rows $rows"; python tools/summarize_profile.py $O/t_${v}_$rows 2>/dev/null | grep "hist\|table_sums\|compact"
  done; done 2>&1 | tee $O/hist_exp.txt ;;
side)            # rows D / M, f-2, f-3, f-4: tests of the resident store, the side blocks of bench.py, kernel trace of the same calls
  timeout 900 python -m pytest tests/test_seq_store.py tests/test_pcr.py tests/test_validate.py tests/test_validate_bwt.py tests/test_dimer.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.txt
  timeout 900 python tools/side_bench.py > $O/side_bench.json 2> $O/side_bench.err; tail -c 600 $O/side_bench.err; python -c "
import json; d = json.load(open('$O/side_bench.json'))
for k, v in d.items():
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, (dict, str))})
    for a, b in v.items():
        if isinstance(b, dict): print('   ', a, {x: (float('%.4g' % y) if isinstance(y, float) else y) for x, y in b.items() if x != 'sample'})
"
  rocprofv3 --kernel-trace --stats -d $O/prof/trace -o t -- python tools/side_bench.py > /dev/null 2>&1
  python tools/summarize_profile.py $O/prof 2>/dev/null | head -24 | tee $O/side_kernels.txt ;;
shapes)          # bench.py's shard_shapes block alone (headline + 8x1 shard + 4x2 / 2x4 / 1x8), no side measurements
  timeout 1200 python bench.py --steps 40 --warmup 5 --no-variants --no-pipeline --no-side --no-ksweep "$@" > $O/bench.txt 2> $O/bench.err; tail -c 400 $O/bench.err; tail -1 $O/bench.txt
  python -c "
import json
d = json.load(open('bench_detail.json'))
for k, v in d.get('shard_shapes', {}).items(): print(k, v)
print('8x1', d['weak_shard']['ms_per_step'], d.get('projected_strong_scaling'))
" | tee $O/shapes.txt ;;
bench_ranks)     # bench.py's N > 1 control flow on the ONE GPU of a test box: ranks over gloo (RCCL refuses two ranks on one device), every shape
  for spec in "2 1x2" "2 2x1" "4 2x2" "4 1x4" "4 auto"; do set -- $spec
    echo "# N=$1 --shape $2"
    MP_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 29533 \
      bench.py --gpus $1 --steps 10 --warmup 2 --shape $2 2> $O/err_$1_$2.txt | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('n_gpus', 'ms_per_step', 'value', 'parity_checked')}, d['config']['shape'], d['config']['parallelism'])"
  done 2>&1 | tee $O/bench_ranks.txt ;;
share_sweep)     # profiles/r06_share_sweep.txt: one rank's share of the 1x8 / 2x4 / 8x1 job under band lengths and row words per lane of the sliding kernel
  for share in 1x8 2x4 8x1; do for gw in 1 2 4; do for band in 0 4 6 8 10 12 15 20 30; do
    if [ $band = 0 ]; then unset MP_SLIDE_BAND; else export MP_SLIDE_BAND=$band; fi
    MP_EVAL_SLIDE=1 MP_SLIDE_GW=$gw timeout 300 python bench.py --steps 60 --warmup 5 --share $share --no-variants --no-pipeline --no-side --no-ksweep --no-shard --no-cpu 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('share $share gw $gw band $band ms_per_step %.5f kernel_ms %.5f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done; done; done 2>&1 | tee $O/share_sweep.txt ;;
streams)         # one stream (rotating launches) against two (mp_eval_launch_alt): the whole workload and one rank's shares
  for share in "" 1x8 2x4 8x1; do for st in 1 2; do for rep in 1 2; do
    MP_BENCH_STREAMS=$st timeout 300 python bench.py --steps 80 --warmup 6 ${share:+--share $share} --no-variants --no-pipeline --no-side --no-ksweep --no-shard --no-cpu 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('share ${share:-1x1} streams $st ms_per_step %.5f kernel_ms %.5f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done; done; done 2>&1 | tee $O/streams.txt ;;
strict_ab)       # profiles/r06_strict_forms.txt: the sliding kernel's strict positions as two-bit counts per side against position by position (MP_SLIDE_STRICT=0)
  timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "strict_position or grouping_paths or kernel_shapes" 2>&1 | tail -4 | tee $O/pytest.txt
  for rep in 1 2 3; do for st in 0 1; do
    MP_SLIDE_STRICT=$st timeout 300 python bench.py --steps 80 --warmup 6 --no-variants --no-pipeline --no-side --no-ksweep --no-shard --no-cpu 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('MP_SLIDE_STRICT=$st ms_per_step %.5f one_stream %.5f kernel_ms %.5f' % (d['ms_per_step'], d.get('ms_per_step_one_stream', 0), d['roofline']['kernel_ms']))"
  done; done 2>&1 | tee $O/strict_forms.txt
  timeout 600 python bench.py --steps 20 --warmup 5 --no-variants --no-pipeline --no-side --no-ksweep --no-shapes 2> $O/bench.err | tail -1 | tee $O/bench_parity.json ;;
exp_libs)        # A/B of experiment builds: tools/reproduce.sh exp_libs TAG... (tools/_build/libmprime_hip_TAG.so; `product` = the tree's library), 3 interleaved repetitions
  for rep in 1 2 3; do for tag in "$@"; do
    lib=$PWD/tools/_build/libmprime_hip_$tag.so; [ $tag = product ] && lib=$PWD/multiprime_amd/csrc/libmprime_hip.so
    MPRIME_LIBRARY=$lib MP_HOST_LIB=$lib MP_BENCH_STREAMS=1 timeout 300 python bench.py --steps 80 --warmup 6 --no-variants --no-pipeline --no-side --no-ksweep --no-shard --no-cpu 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('$tag ms_per_step %.5f kernel_ms %.5f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done; done 2>&1 | tee $O/exp_libs.txt
  timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "strict_position or grouping_paths" 2>&1 | tail -2 | tee -a $O/exp_libs.txt ;;
stats_ab)        # window statistics: G consecutive windows per workgroup (MP_STATS_GROUP=4 default, 8) against the per-window kernel (=0): tests, kernel trace, run() laps
  timeout 900 python -m pytest tests/test_window_stats.py tests/test_core_golden.py -q -m gpu -x 2>&1 | tail -3 | tee $O/pytest.txt
  for g in 0 4 8; do for rows in 131072 1048576; do
    echo "# MP_STATS_GROUP=$g rows $rows"
    MP_STATS_GROUP=$g python tools/profile_pipeline.py --rows $rows --out $O/prof_${g}_$rows 2>&1 | grep "window_stats\|^kernel"
    MP_STATS_GROUP=$g MP_TRACE_PY=1 python tools/profile_run.py $rows 18 2>&1 | grep "^{" | tail -1 | cut -c1-600
  done; done 2>&1 | tee $O/stats_ab.txt; rm -rf $O/prof_* ;;
compact_ab)      # compact_kernel per piece of a window's table: the suites that read its entries, then the kernel rows of the core step at both depths
  timeout 1500 python -m pytest tests/test_hist_paths.py tests/test_hip_parity.py tests/test_core_golden.py tests/test_scale_parity.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
  for rows in 131072 1048576; do python tools/profile_pipeline.py --rows $rows --out $O/prof_$rows 2>&1 | grep "compact\|table_sums\|hist2\|^kernel"; done 2>&1 | tee $O/compact.txt
  MP_TRACE=1 python tools/profile_run.py 1048576 18 2>&1 | grep "unique:\|^{" | tail -8 | cut -c1-400 | tee -a $O/compact.txt; rm -rf $O/prof_* ;;
k_sweep)         # bench.py's k_sweep block alone (the headline workload at k = 20, 22, 36), after the headline
  timeout 900 python bench.py --steps 20 --warmup 5 --no-variants --no-pipeline --no-side --no-shapes 2> $O/bench.err | tail -1 > $O/bench.json; python -c "
import json
d = json.load(open('bench_detail.json'))
print('k_18', d['ms_per_step'], d['roofline']['kernel_ms'], d['parity_checked'])
for k, v in d.get('k_sweep', {}).items(): print(k, v if not isinstance(v, dict) else {x: v[x] for x in ('kernel_ms', 'evals_per_s', 'eval_mode', 'compulsory_frac', 'parity_checked') if x in v})
" | tee $O/k_sweep.txt ;;
exp_rows)        # profiles/r06_exp_one_row.txt: experiment builds of the sliding kernel (tools/build_variant.sh rowN evalslide.hip -DSLIDE_EXP_ONE_ROW=N; wrong results, timing only)
  for rep in 1 2; do for tag in product row1 row2 row3; do
    lib=$PWD/tools/_build/libmprime_hip_$tag.so; [ $tag = product ] && lib=$PWD/multiprime_amd/csrc/libmprime_hip.so
    MPRIME_LIBRARY=$lib MP_HOST_LIB=$lib MP_BENCH_STREAMS=1 timeout 300 python bench.py --steps 80 --warmup 6 --no-variants --no-pipeline --no-side --no-ksweep --no-shard --no-cpu 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('$tag ms_per_step %.5f kernel_ms %.5f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
  done; done 2>&1 | tee $O/exp_one_row.txt
  for st in 1 0; do
    MP_SLIDE_STRICT=$st timeout 900 python tools/collect_counters.py --rows 1048576 --out $O/prof_strict$st > $O/collect_strict$st.log 2>&1
    python -c "
import json
d = json.load(open('$O/prof_strict$st/counters.json'))
for e in d['entries']:
    print('MP_SLIDE_STRICT=$st', 'trace', e.get('trace'), 'raw', {k: round(v) for k, v in e.get('raw_counters_per_launch', {}).items()})
" 2>&1 | tee -a $O/exp_one_row.txt
    rm -rf $O/prof_strict$st/pmc_* $O/prof_strict$st/cal_* $O/prof_strict$st/trace
  done ;;
bench_default)   # profiles/r06_bench.json + r06_bench_detail.json: `python bench.py` with the driver's flags, wall time, the blocks of the detail file in short
  SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.txt 2> $O/bench.err; echo "rc=$? wall=${SECONDS}s"; tail -c 300 $O/bench.err
  tail -1 $O/bench.txt > $O/bench.json; cp bench_detail.json $O/bench_detail.json; wc -c $O/bench.json; cat $O/bench.json
  python -c "
import json
d = json.load(open('bench_detail.json'))
print('k_sweep', json.dumps(d.get('k_sweep'))[:1800])
print('pipeline', {k: (v.get('run_ms'), v.get('construct_ms'), v.get('tsv_equal_oracle')) for k, v in d['pipeline'].items() if isinstance(v, dict)})
for k, v in d['pipeline'].items():
    if isinstance(v, dict): print('phases', k, v.get('phases_ms'))
print('side', {k: (v.get('parity_checked') if isinstance(v, dict) else v) for k, v in d['side_steps'].items()})
print('shapes', {k: (v.get('ms_per_step'), v.get('parity_checked')) for k, v in d.get('shard_shapes', {}).items() if isinstance(v, dict)})
print('variants', {k: (v.get('evals_per_s'), v.get('parity_checked')) for k, v in d.get('variants', {}).items() if isinstance(v, dict)})
" | tee $O/blocks.txt ;;
pipeline_kernels) # profiles/r06_pipeline_kernels.txt + .json: every kernel of the core step at both depths (trace + one SQ counter pass), bytes model, frac
  rm -f $O/pipeline_kernels.json
  for rows in 131072 1048576; do python tools/profile_pipeline.py --rows $rows --out $O/prof_$rows --json $O/pipeline_kernels.json; echo; done 2>&1 | tee $O/pipeline_kernels.txt ;;
final)           # everything the round's profiles/ files come from, in the order their consumers need them (copy gpurun_out/r06/final/* to profiles/r06_*)
  timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
  rm -f $O/pipeline_kernels.json
  for rows in 131072 1048576; do python tools/profile_pipeline.py --rows $rows --out $O/prof_pipe_$rows --json $O/pipeline_kernels.json; echo; done > $O/pipeline_kernels.txt 2>&1
  cp $O/pipeline_kernels.json profiles/r06_pipeline_kernels.json
  timeout 900 python tools/collect_counters.py --rows 1048576 --out $O/prof_1m > $O/collect_1m.log 2>&1
  timeout 600 python tools/collect_counters.py --rows 131072 --out $O/prof_131k --merge $O/prof_1m/counters.json > $O/collect_131k.log 2>&1
  cp $O/prof_131k/counters.json profiles/r06_counters.json; cp $O/prof_131k/counters.json $O/counters.json
  cp $O/prof_1m/summary.txt $O/bench_eval_1m.txt 2>/dev/null; cp $O/prof_131k/summary.txt $O/bench_eval.txt 2>/dev/null
  rocprofv3 --kernel-trace --stats -d $O/prof_side/trace -o t -- python tools/side_bench.py > /dev/null 2>&1
  python tools/summarize_profile.py $O/prof_side 2>/dev/null | head -24 > $O/side_kernels.txt
  SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.txt 2> $O/bench.err; echo "bench rc=$? wall=${SECONDS}s"
  tail -1 $O/bench.txt > $O/bench.json; cp bench_detail.json $O/bench_detail.json; wc -c $O/bench.json; cat $O/bench.json
  (for rows in 131072 1048576; do for k in 18 22; do echo "# rows $rows k $k"; laps $rows $k; done; done) > $O/run_laps.txt 2>&1
  du -sh $O; rm -rf $O/prof_pipe_* $O/prof_1m/pmc_* $O/prof_131k/pmc_* $O/prof_1m/cal_* $O/prof_131k/cal_* $O/prof_side ;;
soak)            # profiles/r06_soak.txt: randomised HIP-vs-oracle soaks through every kernel setting (incl. hist2 / hist / rep-rows, eval_chain_x shapes, v <= 5)
  (timeout 500 python tools/soak_parity.py --seconds ${1:-240} --seed 61; timeout 200 python tools/soak_parity.py --seconds 60 --seed 62; timeout 200 python tools/soak_primers.py --seconds 45) 2>&1 | tee $O/soak.txt ;;
side_counters)   # profiles/r06_side_kernels.json (+ .txt): kernel trace and FETCH_SIZE / WRITE_SIZE passes of tools/side_bench.py, per kernel
  python tools/side_counters.py --out $O 2>&1 | tee $O/side_kernels_counters.txt ;;
*) echo "unknown target $T"; exit 2 ;;
esac
