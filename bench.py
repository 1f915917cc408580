#!/usr/bin/env python3
"""Benchmark of the hot path: batched candidate x sequence coverage evaluation (mp_eval_launch) on the synthetic alignment of
SURVEY §8d input 4 / BASELINE.json configs[3]: 1M x 1 kb sequences, sharded over the GPUs of the job.

One step = one pass of the evaluation over every window with C candidates per window, plus (N > 1) the RCCL all-reduce of the
per-candidate coverage counters.  Inputs are resident in HBM (planes built once, untimed).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

The workload is config 4 at every N: 1 048 576 x 1000 rows in all, 1 048 576 / N per GPU (`scaling: "strong"`).  At N = 1 the
headline (`value`, `ms_per_step`, `config.workload`, `roofline`, `cpu_baseline`) is config 4 itself on one GPU — the one size
where HBM can bind (planes 0.5 GB > Infinity Cache).  `--rows R` fixes the rows per GPU instead (weak scaling, `scaling: "weak"`).
At N > 1 the headline step does ONE collective per step, as the drop-in pipeline does per alignment; the bucketed, overlapped
form (`--bucket` steps per collective, dist.StepBuckets) is the side key `ms_per_step_bucketed`.

  weak_shard    N = 1 only: the 131072 x 1000 shard one GPU holds in the 8-GPU job (81 MB of planes: lives in the Infinity Cache),
                timed the same way in this same run (skip with --no-shard) — what the per-GPU kernel time of an N = 8 run should be.
  roofline      `frac` = `compulsory_frac` = the USEFUL HBM fraction: compulsory bytes of a pass (the packed planes read once, N L 3/8 —
                SURVEY 8d (ii)) / kernel time / 8 TB/s.  `hbm_frac` = what the kernel actually moved: measured fabric/HBM bytes per
                launch (`traffic`, rocprofv3 counters) / kernel time / 8 TB/s; `over_fetch` = traffic / compulsory bytes.  Beside them,
                every fraction that can bind (all <= 1 by construction):
                  valu_frac = VALU wave-instructions per launch / (kernel time x SIMDs x measured issue rate)
                  l2_frac   = bytes returned to registers per launch / (kernel time x 31.4 TB/s = 256 CUs x 64 B/clk)
                `bound` names the largest.  Kernel time is measured live (HIP events on the library's stream); instruction /
                request / byte counts come from separate rocprofv3 --pmc passes of this same command (tools/collect_counters.py
                -> profiles/r04_counters.json).  An entry is used only if it was collected for this exact configuration AND
                this exact kernel source (`source_hash` = sha256 of the evaluation kernels' sources): after a kernel change the
                fractions are dropped (`counters_stale`), never silently reused.  SURVEY §8d's algorithmic figure (3k/8 bytes
                per evaluation / kernel time / 8 TB/s) is kept as `algorithmic_frac`, labelled: it exceeds 1 (8 nested
                candidates and 18 overlapping windows share every loaded plane word) and is NOT a roofline fraction.
  variants      SURVEY 8d's micro-benchmark, at the headline size AND at the shard: the same kernel library on other candidate sets —
                C = 1, C = 8 unrelated (no nesting), the symbol-table kernel forced on the nested set, C = 64 (eight nested chains
                per window); at the headline size every variant's counters are checked against the oracle (every 16th window);
                at the shard also one cold launch (caches flushed, no warm-up).
  pipeline      the REAL step the kernel belongs to: NN_degenerate(...).run() (the drop-in class, --no-json, bitsets on the device) on
                the same rows at both sizes, median of 5, phase split, TSV SHA-256 against the checker's (tests/golden/synth_pipeline.json).
  shard_shapes  N = 1 only: one rank's share under the 2-D shapes 4x2, 2x4, 1x8 (row shards x window groups, dist.ShardGrid), timed and checked the same way.
  projected_strong_scaling   ms_per_step (whole workload, one GPU) / the best ms_per_step of one rank's share over the shapes 8x1 (`weak_shard`), 4x2, 2x4, 1x8:
                what 8 GPUs reach with a free all-reduce.
  cpu_baseline  the plain-C oracle on EVERY host core (threads over row blocks of the whole workload) and on one core (bounded
                sample); its counters are compared with the GPU's candidate by candidate — `parity_checked` true, or the run
                exits non-zero.  `python_reference`: the reference's own algorithm (dict of k-mers, numpy score-table
                differences, V20:229-233 / 1103-1130) restated in Python, one core, bounded sample, checked against the oracle.
  comm          N > 1: what the communicator reports (`rccl_ranks_seen`, the librccl in use), per-rank step time min / max.
"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
# the workload, the roofline arithmetic (tools/bench_common.py) and the side measurements that fill bench_detail.json (tools/bench_blocks.py);
# this file keeps the contract: the timed region and the one line the driver parses
# (the names tests/ and tools/ use through `import bench` stay importable from here)
from bench_common import (CONFIG4_CHECKSUM, FULL_ROWS, SHARD_ROWS, Workload, eval_mode, expand_exceptions, kernel_source_hash, load_json,  # noqa: E402,F401
                          make_candidates, roofline_block, synth_rows, time_launches)
from bench_blocks import (OracleBlocks, cpu_baseline, k_sweep, pipeline_block, python_reference_leg, run_variants, shard_shapes,  # noqa: E402,F401
                          weak_shard)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=0, help="sequences per GPU (weak scaling); 0 = config 4 split over the GPUs: 1048576 / N each (strong scaling)")
    ap.add_argument("--cols", type=int, default=1000)
    ap.add_argument("--k", type=int, default=18)
    ap.add_argument("--v", type=int, default=1)
    ap.add_argument("--cands", type=int, default=8, help="candidates per window")
    ap.add_argument("--seed", type=int, default=20250303)
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the workload the CPU oracle evaluates (0 = all of them)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads of the all-core CPU leg (0 = every core)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="N = 1: skip the `pipeline` block (NN_degenerate.run() on the same rows)")
    ap.add_argument("--no-shapes", action="store_true", help="N = 1: skip the `shard_shapes` block (one rank's share of the 2-D shards 4x2, 2x4, 1x8)")
    ap.add_argument("--no-ksweep", action="store_true", help="N = 1: skip the `k_sweep` block (the headline workload at k = 20, 22, 36)")
    ap.add_argument("--no-side", action="store_true", help="N = 1: skip the `side_steps` block (dimer scan, in-silico PCR, k-mismatch scan: tools/side_bench.py)")
    ap.add_argument("--no-shard", "--no-full", dest="no_shard", action="store_true",
                    help="N = 1: skip the weak_shard block (the 131072-row shard of the 8-GPU job)")
    ap.add_argument("--shape", default=None, metavar="RxG|auto", help="N > 1: R row shards x G window groups (R x G = N); default auto = as many window "
                    "groups as the device memory allows (dist.ShardGrid.best_shape); Nx1 = row shards only")
    ap.add_argument("--share", default=None, metavar="RxG", help="N = 1 experiment: time one rank's share of an R x G job instead of the whole workload")
    ap.add_argument("--bucket", type=int, default=4, help="steps per collective of the bucketed side measurement (N > 1)")
    a = ap.parse_args()
    import faulthandler
    faulthandler.enable()           # a fatal signal leaves every thread's Python stack on stderr (the driver keeps stderr)
    t_main = time.time()

    def progress(block):
        """MP_BENCH_PROGRESS=1: which block starts when, on stderr."""
        if os.environ.get("MP_BENCH_PROGRESS"):
            print("[bench] %7.2f s  %s" % (time.time() - t_main, block), file=sys.stderr, flush=True)

    from multiprime_amd._abi import Library, prefer_staged_copies
    prefer_staged_copies()          # a program's decision, before the HIP runtime starts: what the drop-in command lines do (the `pipeline` block times their code path)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    strong = a.rows == 0
    # N > 1: the shape of the job — R row shards x G window groups (multiprime_amd.dist.ShardGrid): rank r holds row shard r // G and evaluates
    # window group r % G; the counters' all-reduce runs inside a row group (the R ranks with the same r % G).  Default: as many window groups as
    # the device memory allows (config 4: 1 x N — every GPU holds every row, 5 GB, and there is nothing to reduce); --shape Nx1 = row shards only.
    from multiprime_amd.dist import ShardGrid
    R, G = (ShardGrid.best_shape(world, FULL_ROWS if strong else a.rows * world, a.cols) if a.shape in (None, "auto") else ShardGrid.parse(a.shape, world))
    if not strong:
        R, G = world, 1                                 # --rows fixes the rows per GPU: row shards
    if strong and FULL_ROWS % R:
        raise SystemExit(f"config 4 ({FULL_ROWS} rows) does not split evenly into {R} row shards")
    ri, gi = divmod(rank, G)
    rows_per_gpu = FULL_ROWS // R if strong else a.rows
    if a.share:                                         # N = 1 experiments: the main workload is ONE rank's share of an R x G job (tools/reproduce.sh share_sweep)
        if world != 1:
            raise SystemExit("--share is a one-GPU experiment")
        sR, sG = (int(x) for x in a.share.lower().split("x"))
        rows_per_gpu, G, strong = FULL_ROWS // sR, sG, False
    # MP_BENCH_BACKEND=gloo (testing only): several ranks on whatever GPUs the box has, collectives through the host —
    # exercises the N > 1 control flow of this file on a 1-GPU box; RCCL itself refuses two ranks on one device
    backend = os.environ.get("MP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    k, C, L = a.k, a.cands, a.cols
    # the library brackets every n-th mp_eval_launch with a HIP-event pair on its stream (live kernel time for the
    # roofline); a pair idles the stream for ~6 us, so the timed region samples one launch in four
    every = int(os.environ.setdefault("MP_EVAL_TIMING_EVERY", "4"))
    lib = Library()                                   # the HIP library or an error: no fallback
    w = Workload(lib, local, torch, ri * rows_per_gpu, rows_per_gpu, a, win_part=(gi, G))
    ctx, n_cand = w.ctx, w.n_cand
    row_group = None                                    # the ranks this rank's counters are summed with: its row group
    if world > 1 and G > 1 and R > 1:
        for g in range(G):                              # (every rank creates every group, in the same order)
            grp = dist.new_group([r * G + g for r in range(R)])
            if g == gi:
                row_group = grp

    # N > 1: the counters of every step are all-reduced over RCCL inside the timed region.  Headline: ONE collective per step
    # (what the drop-in pipeline does per alignment).  Side key: `--bucket` consecutive steps write into one [bucket][n_cand][3]
    # buffer that is reduced with one collective on RCCL's stream while the next bucket's steps evaluate into the other buffer
    # (dist.StepBuckets; xGMI rings are latency-bound at 180 KB per step).
    from multiprime_amd.dist import StepBuckets
    rotate = os.environ.get("MP_BENCH_ROTATE", "1") != "0"
    # Steps are independent evaluations of one staged candidate set into alternating counter blocks.  Where no collective follows a step
    # (one GPU, or window groups of one row shard) they are issued alternately on the context's TWO streams (mp_eval_launch /
    # mp_eval_launch_alt): the next kernel's dispatch, first misses and warm-up columns overlap the last workgroups of the one before it —
    # worth 2 % on the whole workload and 30 % on a 1/8 share, whose 27 us kernel holds 17 us of window work (profiles/r06_streams.txt).
    # MP_BENCH_STREAMS=1 issues them on one stream (rotating launches), as rounds 1-5 did; the roofline's kernel time ALWAYS comes from
    # such a pass (a launch alone on the chip), reported beside the headline as `ms_per_step_one_stream`.
    two_streams = os.environ.get("MP_BENCH_STREAMS", "2") == "2"

    def timed_region(wl, bucket, n_world, n_reduce=None, group=None, streams=None, steps=None):
        """`n_world` ranks take part in the barriers and the max over ranks; `n_reduce` of them (a row group) in the counters' all-reduce.
        Returns (..., streams used)."""
        sb = StepBuckets(wl.n_cand, bucket, dev, n_world if n_reduce is None else n_reduce, group)
        use2 = (two_streams if streams is None else streams == 2) and sb.world == 1
        keep_every = os.environ.get("MP_EVAL_TIMING_EVERY")
        if use2:
            os.environ["MP_EVAL_TIMING_EVERY"] = "0"            # (an event pair between two overlapping kernels measures neither)

        if use2:
            def step():             # steps alternate between the context's two streams and between two counter blocks (mp_eval_launch_alt)
                blk = sb.begin_step()
                (wl.ctx.eval_launch if sb.i % 2 == 0 else wl.ctx.eval_launch_alt)(blk.data_ptr())
                sb.end_step()
        elif rotate:
            def step():             # the launch that fills a step's block clears the next step's inside its own grid (mp_eval_launch_rotating)
                out, nxt = sb.begin_rotating()
                wl.ctx.eval_launch_rotating(out.data_ptr(), nxt.data_ptr())
                sb.end_step()
        else:
            def step():             # MP_BENCH_ROTATE=0: a fill dispatch in front of every evaluation (mp_eval_launch)
                wl.ctx.eval_launch(sb.begin_step().data_ptr())
                sb.end_step()

        for _ in range(a.warmup):
            step()
        sb.drain()
        wl.ctx.eval_timing(reset=True)
        torch.cuda.synchronize()
        if n_world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps if steps is None else steps):
            step()
        sb.drain()
        torch.cuda.synchronize()
        if n_world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kern_ms, kern_n = wl.ctx.eval_timing(reset=True)
        samples = np.sort(wl.ctx.eval_timing_samples())
        if keep_every is None:
            os.environ.pop("MP_EVAL_TIMING_EVERY", None)
        else:
            os.environ["MP_EVAL_TIMING_EVERY"] = keep_every
        tt = torch.tensor([dt, -dt], dtype=torch.float64, device=dev)
        if n_world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt[0].item()), kern_ms, kern_n, samples, sb, (-float(tt[1].item()), float(tt[0].item())), 2 if use2 else 1

    def timed_pair(wl, n_world, n_reduce=None, group=None):
        """The headline's timed region and, when that overlapped its steps on two streams, a second one on ONE stream for the kernel's own
        duration (HIP events around single launches: the roofline's kernel time) — (elapsed, kernel ms, launches timed, samples, buckets,
        spread, streams of the first region, ms per step on one stream)."""
        pick = None
        if n_world == 1 and two_streams and "MP_BENCH_STREAMS" not in os.environ:
            # one GPU: the launch strategy is chosen by a short trial of both (a kernel that fills the chip for 140 us gains nothing from a
            # second stream and may lose to it; a 27 us share gains 30 %) — like any launch parameter, and reported as `steps_in_flight`
            trial = {st: min(timed_region(wl, 1, 1, streams=st, steps=12)[0] for _ in range(2)) for st in (1, 2)}
            pick = min(trial, key=trial.get)
        elapsed, kern_ms, kern_n, samples, sb, spread, used = timed_region(wl, 1, n_world, n_reduce, group, streams=pick)
        one_ms = elapsed / a.steps * 1e3
        if used == 2:
            e1, kern_ms, kern_n, samples, _, _, _ = timed_region(wl, 1, 1, streams=1)
            one_ms = e1 / a.steps * 1e3
        return elapsed, kern_ms, kern_n, samples, sb, spread, used, one_ms

    elapsed, kern_ms, kern_n, samples, sb, spread, streams_used, one_stream_ms = timed_pair(w, world, R, row_group)
    gpu_counters = sb.block_of(a.steps - 1).cpu().numpy().copy()       # [n_cand][3], summed over ranks when N > 1
    bucketed = None
    if world > 1 and a.bucket > 1:
        e1, *_ = timed_region(w, a.bucket, world, R, row_group, streams=1)
        bucketed = e1

    ev = torch.tensor([w.evals], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(ev, op=dist.ReduceOp.SUM)
    evals_total = int(ev.item())
    # the counters of every candidate summed: over the window groups (their leaders, row shard 0, hold a group's reduced counters) this is the
    # checksum of the whole workload — at N = 1 it comes out of the run whose counters are compared with the oracle's one by one, so a
    # committed value lets an N > 1 run (no oracle leg there) say whether ITS counters are the same
    cs = torch.tensor(gpu_counters.sum(axis=0) if ri == 0 else np.zeros(3, np.int64), dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(cs, op=dist.ReduceOp.SUM)
    checksum = [int(x) for x in cs.tolist()]

    # what the communicator itself says (N > 1): the ranks RCCL saw, through a communicator of the library's own
    comm_info = None
    if world > 1:
        comm_info = {"backend": backend, "torch_world": dist.get_world_size()}
    # measured device-copy bandwidth (SURVEY §8d asks for it next to the spec peak): 1 GiB d2d, read + write
    copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    res = None
    if rank == 0:
        mode = eval_mode(w.n_rows, ctx)
        per_launch_ms = kern_ms / max(kern_n, 1)
        whole = "BASELINE configs[3] itself" if strong else f"{world * rows_per_gpu} rows in all"
        res = {
            "metric": "candidate x sequence coverage evals/s",
            "value": evals_total * a.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "steps_in_flight": streams_used, "ms_per_step_one_stream": one_stream_ms,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{w.describe()} per GPU x {world} GPU(s) = {whole} (SURVEY 8d input 4; synthetic 1M x 1 kb; shape {R}x{G}: row shards x window groups)",
                       "rows_per_gpu": rows_per_gpu, "rows_total": R * rows_per_gpu, "cols": L, "k": k, "variation": a.v,
                       "candidates_per_window": C, "windows": w.W, "evals_per_step_per_gpu": w.evals, "iupac_extra_rows": w.n_extra,
                       "shape": f"{R}x{G}",
                       "parallelism": ("one GPU holds every row" if world == 1 else
                                       f"{G} window groups x 1 row shard: every GPU holds every row and evaluates 1/{G} of the windows, no collective" if R == 1 else
                                       f"{R} row shards x {G} window group(s): one RCCL all-reduce of the [{n_cand}x3] int64 counters per step inside a row group of {R} ranks")},
            "roofline": roofline_block(w, per_launch_ms, samples, kern_n, every, mode),
            "measured_copy_GBs": copy_gbs, "setup_s": w.setup_s, "device_bytes": ctx.device_bytes(), "counter_checksum": checksum,
        }
        default_workload = strong and (a.k, a.v, a.cands, a.cols, a.seed) == (18, 1, 8, 1000, 20250303)
        if world > 1 and default_workload:
            res["parity_checked"] = checksum == CONFIG4_CHECKSUM
            res["parity_note"] = "the sum of all candidates' counters equals the N = 1 run's, whose counters were compared with the oracle's one by one"
        if world > 1:
            res["step_time_ranks_ms"] = {"min": spread[0] / a.steps * 1e3, "max": spread[1] / a.steps * 1e3}
            res["comm"] = comm_info
        if bucketed is not None:
            res["ms_per_step_bucketed"] = bucketed / a.steps * 1e3
            res["bucket"] = a.bucket
        if world == 1:
            res["step_form"] = ("rotating: the launch that fills a step's counter block clears the next step's inside its own grid (mp_eval_launch_rotating)"
                                if rotate else "a fill dispatch in front of every evaluation (mp_eval_launch; MP_BENCH_ROTATE=0)")
            progress("cpu_baseline")
            blocks = None
            if not a.no_cpu:
                n_cpu = a.cpu_rows or rows_per_gpu
                blocks = OracleBlocks(w, w.rows[:n_cpu], a.cpu_threads)
                res["cpu_baseline"] = cpu_baseline(w, blocks, gpu_counters if n_cpu == rows_per_gpu else None, a.seed)
                res["parity_checked"] = res["cpu_baseline"].get("parity_checked")
            # The contract's line as soon as its fields exist (value, roofline, cpu_baseline, parity): should one of the side measurements below take
            # the process down — a fatal signal cannot be caught — the LAST stdout line is still a complete headline.  The final line replaces it.
            print(headline({**res, "provisional": True}), flush=True)
            progress("variants")
            if not a.no_variants:
                # SURVEY 8d's micro-benchmark at the headline size: C = 1, unrelated C = 8, nested C = 64, each checked against the oracle
                try:
                    res["variants"] = run_variants(w, torch, dev, None, None, a.seed, blocks if blocks is not None and blocks.n == rows_per_gpu else None, cold=False)
                except Exception as e:          # noqa: BLE001 — a side measurement must not take the headline line with it
                    res["variants"] = {"error": {"error": f"{type(e).__name__}: {e}"}}
                if any(v.get("parity_checked") is False for v in res["variants"].values()):
                    res["parity_checked"] = False
            if blocks is not None:
                blocks.close()
                del blocks
            progress("pipeline")
            if not a.no_pipeline:
                # the headline's context goes first: a process of the drop-in holds ONE context, and the library divides its pool of released
                # device blocks by the contexts alive on the device — beside the headline's 5 GB context the pipeline's constructor got every
                # block of its 1 GB alignment from hipMalloc again (0.16 s instead of 0.05 s; tools/construct_probe.py)
                sb = None
                if w.ctx is not None:
                    w.ctx.close()
                w.ctx = ctx = None
                torch.cuda.empty_cache()
                res["pipeline"] = {"what": "the REAL step: the drop-in class NN_degenerate(...).run() (multiprime_amd/core.py; --no-json, coverage bitsets kept on the "
                                           "device) on the same synthetic rows, median of 5 after one warm-up; `run_ms` is run() alone (the context is kept across repetitions, as the "
                                           "--batch workers keep theirs across alignments), `construct_ms` the constructor before it (FASTA parse of a file in /dev/shm + mp_load_msa); TSV compared with the CHECKER's "
                                           "(oracle/core_ref.py over the plain-C oracle, tests/golden/synth_pipeline.json)",
                                   f"rows_{rows_per_gpu}": pipeline_block(lib, local, w.rows, a)}
    # N = 1: the shard one GPU holds in the 8-GPU job, timed the same way (with the variant measurements)
    progress("weak_shard")
    if rank == 0 and world == 1 and not a.no_shard and rows_per_gpu != SHARD_ROWS:
        sb = None
        if w.ctx is not None:
            w.ctx.close()
        w.ctx = ctx = None
        rows_full, w.rows = w.rows, None
        torch.cuda.empty_cache()
        try:
            res["weak_shard"] = weak_shard(lib, local, torch, dev, a, timed_pair, every, not a.no_cpu)
        except Exception as e:                  # noqa: BLE001 — as above
            res["weak_shard"] = {"error": f"{type(e).__name__}: {e}"}
        if res["weak_shard"].get("parity_checked") is False:
            res["parity_checked"] = False
        if "pipeline" in res and "pipeline" in res["weak_shard"]:
            res["pipeline"][f"rows_{SHARD_ROWS}"] = res["weak_shard"].pop("pipeline")
        # what row shards over 8 GPUs can reach at best: the whole workload's step over the shard's step (no collective counted)
        if "ms_per_step" in res["weak_shard"]:
            res["projected_strong_scaling"] = {
                "n_gpus": FULL_ROWS // SHARD_ROWS, "ceiling": res["ms_per_step"] / res["weak_shard"]["ms_per_step"],
                "ms_per_step_one_gpu": res["ms_per_step"], "ms_per_step_shard": res["weak_shard"]["ms_per_step"],
                "shape": "8x1",
                "ceiling_one_stream": res["ms_per_step_one_stream"] / res["weak_shard"]["ms_per_step_one_stream"],
                "note": "ms_per_step of the whole workload on one GPU over ms_per_step of one rank's share of the 8-GPU job on one GPU, both measured in this run "
                        "the same way (steps on two streams; `ceiling_one_stream`: both on one stream): the speed-up 8 GPUs reach when the exchange of the counters "
                        "costs nothing — true by construction for window groups of one row shard (shape 1x8: no collective), overlapped with the next step's "
                        "kernel for row shards (dist.StepBuckets); north_star asks >= 6"}
        # ... and what the 2-D shards reach (dist.ShardGrid: R row shards x G window groups, the alignment's rows replicated along the window axis,
        # the all-reduce inside a row group only): the share of ONE rank of every shape, on this GPU, in this run, checked against the oracle
        progress("shard_shapes")
        if not a.no_shapes and rows_per_gpu == FULL_ROWS:
            try:
                res["shard_shapes"] = shard_shapes(lib, local, torch, a, timed_pair, rows_full, not a.no_cpu)
                best = min((s for s in res["shard_shapes"].values() if isinstance(s, dict) and "ms_per_step" in s), key=lambda s: s["ms_per_step"], default=None)
                ps = res.get("projected_strong_scaling")
                if best is not None and ps is not None and best["ms_per_step"] < ps["ms_per_step_shard"]:
                    ps.update({"ceiling": res["ms_per_step"] / best["ms_per_step"], "ms_per_step_shard": best["ms_per_step"], "shape": best["shape"],
                               "ceiling_one_stream": res["ms_per_step_one_stream"] / best["ms_per_step_one_stream"]})
                if any(isinstance(s, dict) and s.get("parity_checked") is False for s in res["shard_shapes"].values()):
                    res["parity_checked"] = False
            except Exception as e:              # noqa: BLE001 — as above
                res["shard_shapes"] = {"error": f"{type(e).__name__}: {e}"}
        # the same rows at other primer lengths (k = 20, 22: BASELINE configs[1]; 36: 64-bit window words)
        progress("k_sweep")
        if not a.no_ksweep and rows_per_gpu == FULL_ROWS and a.k == 18:
            try:
                res["k_sweep"] = k_sweep(lib, local, torch, dev, a, rows_full, not a.no_cpu)
                if any(isinstance(b, dict) and b.get("parity_checked") is False for b in res["k_sweep"].values()):
                    res["parity_checked"] = False
            except Exception as e:              # noqa: BLE001 — as above
                res["k_sweep"] = {"error": f"{type(e).__name__}: {e}"}
        del rows_full
    if rank == 0 and world == 1 and any(isinstance(p, dict) and p.get("tsv_equal_oracle") is False for p in res.get("pipeline", {}).values()):
        res["parity_checked"] = False
    # N = 1: the steps either side of the core step, each with its checker leg — detail file only
    progress("side_steps")
    if rank == 0 and world == 1 and not a.no_side:
        try:
            sys.path.insert(0, os.path.join(REPO, "tools"))
            import side_bench
            res["side_steps"] = side_bench.run(lib, local)
            if any(isinstance(b, dict) and b.get("parity_checked") is False for b in res["side_steps"].values()):
                res["parity_checked"] = False
        except Exception as e:                  # noqa: BLE001 — a side measurement must not take the headline line with it
            res["side_steps"] = {"error": f"{type(e).__name__}: {e}"}
    # N > 1: what the communicator itself says, through a communicator of the library's own (csrc/comm.hip) — AFTER everything the headline needs
    # exists, and under a watchdog: this path has only ever run on stand-ins (gloo ranks, the shared-memory stub of librccl), a communicator
    # that never answers must not take the measured line with it.  After 120 s rank 0 prints the line without it and every rank leaves.
    if world > 1 and backend == "nccl" and os.environ.get("MP_BENCH_LIBCOMM", "1") != "0":
        def give_up():
            if rank == 0:
                comm_info["library_communicator_error"] = "no answer within 120 s"
                emit(res)
            os._exit(0)
        dog = threading.Timer(120.0, give_up)
        dog.daemon = True
        dog.start()
        try:
            raw = ctx.comm_unique_id() if rank == 0 else bytes(128)
            box = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
            dist.broadcast(box, src=0)
            ctx.comm_init(world, rank, bytes(box.cpu().numpy().tobytes()))
            seen = ctx.comm_describe()
            one = ctx.comm_sum(np.ones(1, np.int64))
            comm_info.update({"rccl_ranks_seen": seen[0], "rccl_rank": seen[1], "rccl_library": seen[2], "sum_of_ones": int(one[0])})
            ctx.comm_destroy()
        except Exception as e:                                      # noqa: BLE001 — reported, not fatal: the timed steps ran on torch's RCCL
            comm_info["library_communicator_error"] = str(e)
        dog.cancel()
    progress("emit")
    if rank == 0:
        emit(res)
    if world > 1:
        # the line is out: a rank that left through the watchdog above must not keep the others in this barrier for the collective's own timeout
        bad = bool(res.get("parity_checked") is False) if rank == 0 else False
        last = threading.Timer(60.0, lambda: os._exit(1 if bad else 0))
        last.daemon = True
        last.start()
        dist.barrier()
        dist.destroy_process_group()
        last.cancel()
    if rank == 0 and res.get("parity_checked") is False:
        raise SystemExit("bench.py: GPU counters differ from the CPU oracle's")


HEADLINE_LIMIT = 4096        # bytes of the final stdout line: the driver keeps a tail of stdout and parses its last line


def _sig(x, digits=6):
    """Floats of the headline at `digits` significant digits (the detail file keeps full precision)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {key: _sig(val, digits) for key, val in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(val, digits) for val in x]
    return x


def headline(res):
    """The ONE line the driver parses: the contract's keys, `roofline` and `cpu_baseline` of the headline kernel, the parity verdict and
    the projected scaling ceiling — no notes, no tables, no nested side measurements (those go to bench_detail.json and to an
    earlier, prefixed stdout line).  Always shorter than HEADLINE_LIMIT (tests/test_bench_line.py)."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
           "parity_checked", "ms_per_step_bucketed", "bucket", "steps_in_flight", "ms_per_step_one_stream", "provisional")
    out = {key: res[key] for key in top if key in res}
    cfg = res.get("config", {})
    out["config"] = {key: cfg[key] for key in ("workload", "rows_per_gpu", "rows_total", "cols", "k", "variation", "candidates_per_window", "windows",
                                               "evals_per_step_per_gpu", "shape", "parallelism") if key in cfg}
    out["config"]["workload"] = str(out["config"].get("workload", ""))[:240]
    rf = res.get("roofline") or {}
    # `bound` of the contract is the roofline `peak` belongs to (HBM; integer bit-mask work has no MFMA roofline); `limiter` is the
    # largest of the measured fractions (valu / l2 / hbm), i.e. what actually binds the kernel
    out["roofline"] = {"bound": "hbm", "achieved": rf.get("achieved"), "peak": rf.get("peak"), "unit": rf.get("unit"), "frac": rf.get("frac"),
                       "traffic": rf.get("traffic"), "hbm_frac": rf.get("hbm_frac"), "over_fetch": rf.get("over_fetch"),
                       "valu_frac": rf.get("valu_frac"), "l2_frac": rf.get("l2_frac"), "limiter": rf.get("bound"),
                       "algorithmic_frac": rf.get("algorithmic_frac"), "compulsory_bytes": rf.get("compulsory_bytes"),
                       "kernel": str(rf.get("kernel", "")).split(" (")[0], "eval_mode": rf.get("eval_mode"), "kernel_ms": rf.get("kernel_ms"),
                       "counters_stale": rf.get("counters_stale")}
    cb = res.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": str(cb.get("sample", ""))[:200], "parity_checked": cb.get("parity_checked")}
        if isinstance(cb.get("one_core"), dict):
            out["cpu_baseline"]["one_core_value"] = cb["one_core"].get("value")
    ps = res.get("projected_strong_scaling")
    if ps:
        out["projected_strong_scaling"] = {key: ps[key] for key in ("n_gpus", "ceiling", "shape", "ms_per_step_one_gpu", "ms_per_step_shard",
                                                                    "ceiling_one_stream") if key in ps}
    pl = res.get("pipeline") or {}
    runs = {key: {"run_ms": val.get("run_ms"), "construct_ms": val.get("construct_ms"), "tsv_equal_oracle": val.get("tsv_equal_oracle")}
            for key, val in pl.items() if isinstance(val, dict) and "run_ms" in val}
    if runs:
        out["pipeline"] = runs
    if "comm" in res:
        out["comm"] = {key: res["comm"][key] for key in ("backend", "rccl_ranks_seen", "sum_of_ones") if key in res["comm"]}
    if "step_time_ranks_ms" in res:
        out["step_time_ranks_ms"] = res["step_time_ranks_ms"]
    out["detail"] = "bench_detail.json"
    out = _sig(out)
    line = json.dumps(out, separators=(",", ":"))
    for drop in ("pipeline", "comm", "step_time_ranks_ms", "projected_strong_scaling"):       # never reached with the fields above; the limit holds regardless
        if len(line) < HEADLINE_LIMIT:
            break
        out.pop(drop, None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line) >= HEADLINE_LIMIT:
        raise SystemExit(f"bench.py: headline line of {len(line)} bytes")
    return line


def emit(res):
    """Everything measured -> bench_detail.json beside this file (and gpurun_out/ when that exists) and ONE earlier stdout line prefixed
    `bench_detail: ` (not a JSON line by itself); then the headline as the LAST stdout line."""
    detail = json.dumps(res)
    for d in (REPO, os.path.join(REPO, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(detail + "\n")
            except OSError:
                pass
    print("bench_detail: " + detail, flush=True)
    print(headline(res), flush=True)


def supervise(cmd, env=None, out=None, attempts=2):
    """N = 1: the measuring runs in a CHILD process and this one only relays its stdout.  A fatal signal inside the runtime cannot be caught
    in the process it hits (round 6: two of 57 runs died in a side measurement, never reproduced in isolation — DESIGN 9.-2); from here it can:
      * the child's last line is a final headline           -> nothing to add, its exit status is ours (non-zero when parity failed);
      * the child ended after the PROVISIONAL headline      -> that line again as the last line, `side_measurements` saying what happened, exit 0
        (the timed region, the roofline, the CPU leg and the parity check of the headline were complete when it was printed);
      * the child was ended by a signal before any headline -> one more attempt, then its exit status.
    Nothing is measured here: the timed region is the child's.  `cmd` / `out` exist for tests/test_bench_line.py."""
    import subprocess
    out = out or sys.stdout
    rc = 1
    for attempt in range(1, attempts + 1):
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
        try:                                        # whoever ends this process ends the measuring one with it
            import signal
            for sig in (signal.SIGTERM, signal.SIGINT):
                signal.signal(sig, lambda signum, frame, c=child: (c.terminate(), c.wait(10), sys.exit(128 + signum)))
        except ValueError:                          # (not the main thread: tests)
            pass
        last, line = None, "\n"
        for line in child.stdout:
            out.write(line)
            out.flush()
            if line.startswith("{"):
                try:
                    last = json.loads(line)
                except ValueError:
                    last = None
        rc = child.wait()
        if not line.endswith("\n"):
            out.write("\n")                         # the child ended in the middle of a line
        if isinstance(last, dict) and "metric" in last:
            if not last.get("provisional"):
                return rc
            last.pop("provisional")
            last["side_measurements"] = f"the measuring process ended with status {rc} after the headline was complete; side blocks missing (attempt {attempt})"
            out.write(json.dumps(last, separators=(",", ":")) + "\n")
            out.flush()
            return 0
        print(f"[bench] attempt {attempt}: the measuring process ended with status {rc} before a headline", file=sys.stderr, flush=True)
        if rc >= 0:
            break                                   # its own exit (no GPU, bad flags, an exception): a second attempt would end the same way
    return rc


if __name__ == "__main__":
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "") or "ROCP_TOOL_LIBRARIES" in os.environ      # rocprofv3: one process, one trace
    if "RANK" in os.environ or profiled or os.environ.get("MP_BENCH_CHILD") or os.environ.get("MP_BENCH_NO_SUPERVISOR"):
        main()                      # a rank of torch.distributed.run, a profiled run, or the child below
    else:
        sys.exit(supervise([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env={**os.environ, "MP_BENCH_CHILD": "1"}))

