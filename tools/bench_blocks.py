#!/usr/bin/env python3
"""bench.py's side measurements (everything that goes to bench_detail.json, nothing of the headline): the 131072-row shard, the 2-D
shard shapes, the k sweep, the section-8d variants, the real step (NN_degenerate.run()), and the CPU legs on the oracle — the only
places besides tests/ and smoke() that touch oracle/."""
import hashlib
import os
import shutil
import tempfile
import threading
import time

import numpy as np

from bench_common import (FULL_ROWS, HBM_PEAK_GBS, REPO, SHARD_ROWS, Workload, eval_mode, expand_exceptions, load_json, make_candidates,  # noqa: F401
                          roofline_block, synth_rows, time_launches)

def weak_shard(lib, local, torch, dev, a, timed_pair, every, with_cpu):
    """The 131072 x 1000 shard one GPU holds when config 4 is spread over 8 GPUs: the same steps, timed the same way, on one GPU."""
    w = Workload(lib, local, torch, 0, SHARD_ROWS, a)
    elapsed, kern_ms, kern_n, samples, sb, _, used, one_ms = timed_pair(w, 1)
    counters = sb.block_of(a.steps - 1).cpu().numpy().copy()
    per_launch_ms = kern_ms / max(kern_n, 1)
    out = {"workload": w.describe() + " on ONE GPU (the per-GPU shard of BASELINE configs[3] at N = 8; planes 81 MB: inside the Infinity Cache)",
           "value": w.evals * a.steps / elapsed, "unit": "evals/s", "ms_per_step": elapsed / a.steps * 1e3, "steps": a.steps,
           "steps_in_flight": used, "ms_per_step_one_stream": one_ms,
           "evals_per_step": w.evals, "iupac_extra_rows": w.n_extra, "setup_s": w.setup_s, "device_bytes": w.ctx.device_bytes(),
           "roofline": roofline_block(w, per_launch_ms, samples, kern_n, every, eval_mode(w.n_rows, w.ctx)),
           "counter_checksum": counters.sum(axis=0).tolist()}
    if not a.no_variants:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        out["variants"] = run_variants(w, torch, dev, src, dst, a.seed, None)
        del src, dst
    if with_cpu:
        blocks = OracleBlocks(w, w.rows, a.cpu_threads)
        cb = cpu_baseline(w, blocks, counters, a.seed, one_core=False, python_leg=False)
        blocks.close()
        out["cpu_baseline"] = cb
        out["parity_checked"] = cb.get("parity_checked")
    if not a.no_pipeline:
        out["pipeline"] = pipeline_block(lib, local, w.rows, a)
    return out


def shard_shapes(lib, local, torch, a, timed_pair, rows_full, with_cpu, n_gpus=8):
    """One rank's share of config 4 under every 2-D shape R x G of `n_gpus` ranks (R contiguous row shards x G contiguous window groups;
    8x1 is `weak_shard`): rows [0, 1048576 / R) x window group 0 of G, timed like the headline, counters against the oracle on the same
    rows and windows.  The all-reduce of a shape runs inside a row group (R ranks, [n_candidates / G x 3] counters) and is not counted,
    as in projected_strong_scaling."""
    out = {}
    for R in (4, 2, 1):
        G = n_gpus // R
        n = FULL_ROWS // R
        w = Workload(lib, local, torch, 0, n, a, win_part=(0, G), rows=rows_full[:n])
        elapsed, kern_ms, kern_n, samples, sb, _, used, one_ms = timed_pair(w, 1)
        counters = sb.block_of(a.steps - 1).cpu().numpy().copy()
        blk = {"shape": f"{R}x{G}", "rows": n, "windows": w.W, "ms_per_step": elapsed / a.steps * 1e3, "steps_in_flight": used,
               "ms_per_step_one_stream": one_ms, "kernel_ms": kern_ms / max(kern_n, 1),
               "evals_per_step": w.evals, "eval_mode": eval_mode(w.n_rows, w.ctx), "device_bytes": w.ctx.device_bytes(),
               "allreduce_ranks": R, "allreduce_bytes": int(w.n_cand) * 24}
        if with_cpu:
            blocks = OracleBlocks(w, w.rows, a.cpu_threads)
            want, _ = blocks.eval(w.cw, w.codes)
            blocks.close()
            blk["parity_checked"] = bool(np.array_equal(want, counters))
        out[blk["shape"]] = blk
        del sb
        w.ctx.close()
        w.ctx = None
        torch.cuda.empty_cache()
    return out


def k_sweep(lib, local, torch, dev, a, rows_full, with_cpu, ks=(20, 22, 36)):
    """The headline workload at other primer lengths (BASELINE configs[1] names k = 18-22; the reference takes any -l, V20:64-65): the
    same rows, 8 nested candidates per window, every launch timed with HIP events (20 after 3 warm-ups); counters of every 16th
    window against the oracle.  k = 36 runs on eval_chain_x_kernel (64-bit window words)."""
    import argparse
    out = {}
    keep = os.environ.get("MP_EVAL_TIMING_EVERY")
    os.environ["MP_EVAL_TIMING_EVERY"] = "1"
    for kk in ks:
        ak = argparse.Namespace(**{**vars(a), "k": kk})
        w = Workload(lib, local, torch, 0, rows_full.shape[0], ak, rows=rows_full)
        buf = torch.zeros((w.n_cand, 3), dtype=torch.int64, device=dev)
        t = time_launches(w.ctx, torch, buf.data_ptr(), 20, 3)
        blk = {"k": kk, "windows": w.W, "evals_per_step": w.evals, "kernel_ms": t["mean_ms"], "kernel_ms_median": t["median_ms"],
               "evals_per_s": w.evals / (t["mean_ms"] * 1e-3), "eval_mode": "chain_x" if kk > 31 else eval_mode(w.n_rows, w.ctx),
               "compulsory_frac": w.n_rows * w.L * 3 / 8.0 / (t["mean_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if with_cpu:
            blocks = OracleBlocks(w, w.rows, a.cpu_threads)
            sel = np.nonzero(w.cw % 16 == 0)[0]
            want, _ = blocks.eval(np.ascontiguousarray(w.cw[sel]), np.ascontiguousarray(w.codes[sel]))
            blocks.close()
            blk["parity_checked"] = bool(np.array_equal(buf.cpu().numpy()[sel], want))
        out[f"k_{kk}"] = blk
        w.ctx.close()
        w.ctx = None
        del buf
        torch.cuda.empty_cache()
    if keep is None:
        os.environ.pop("MP_EVAL_TIMING_EVERY", None)
    else:
        os.environ["MP_EVAL_TIMING_EVERY"] = keep
    return out


def run_variants(w, torch, dev, scratch_a, scratch_b, seed, blocks, cold=True):
    """The evaluation library on other candidate sets — SURVEY 8d's micro-benchmark: C in {1, 8, 64} per window, nested and not (every
    launch timed with HIP events, 20 launches after 3 warm-ups).  With `blocks` (the oracle's contexts over the same rows) the counters
    of every 16th window's candidates are compared with the oracle's."""
    os.environ["MP_EVAL_TIMING_EVERY"] = "1"
    ctx, k, C = w.ctx, w.k, w.C
    total = int(w.universe.sum())
    out = {}

    def measure(name, cand_w, cand_codes, c_per_window, note):
        ctx.eval_upload(cand_w, cand_codes, w.sF, w.sR)
        buf = torch.zeros((len(cand_w), 3), dtype=torch.int64, device=dev)
        t = time_launches(ctx, torch, buf.data_ptr(), 20, 3)
        evals = total * c_per_window
        out[name] = {"evals_per_s": evals / (t["mean_ms"] * 1e-3), "kernel_ms": t["mean_ms"], "kernel_ms_median": t["median_ms"],
                     "kernel_ms_max": t["max_ms"], "candidates_per_window": c_per_window, "what": note, "eval_mode": eval_mode(w.n_rows, ctx),
                     "plan": ctx.eval_plan_info()}
        if blocks is not None:
            sel = np.nonzero(cand_w % 16 == 0)[0]
            want, _ = blocks.eval(np.ascontiguousarray(cand_w[sel]), np.ascontiguousarray(cand_codes[sel]))
            got = buf.cpu().numpy()[sel]
            out[name]["parity_checked"] = bool(np.array_equal(got, want))
            out[name]["parity_note"] = f"the {len(sel)} candidates of every 16th window, all three counters, GPU == sum of the oracle's row blocks"

    uw, ucodes = make_candidates(w.root_codes, w.p0, w.W, k, C, seed + 1, nested=False)
    measure("unrelated_candidates", uw, ucodes, C, f"{C} candidates per window that are NOT a refinement chain (root + one extra base each): symbol-table kernel")
    os.environ["MP_EVAL_GROUP"] = "plain"
    measure("nested_on_table_kernel", w.cw, w.codes, C, "the headline candidates with chain detection off (MP_EVAL_GROUP=plain): symbol-table kernel")
    del os.environ["MP_EVAL_GROUP"]
    measure("c1", w.cw[::C].copy(), w.codes[::C].copy(), 1, "one candidate per window (the root k-mer)")
    # C = 64: eight refinement chains of eight members per window (the root with random extra degeneracy, seeded per chain)
    chains = [make_candidates(w.root_codes, w.p0, w.W, k, 8, seed + 100 + j)[1].reshape(w.W, 8, k) for j in range(8)]
    codes64 = np.ascontiguousarray(np.concatenate(chains, axis=1).reshape(w.W * 64, k))
    measure("nested_c64", np.repeat(np.arange(w.W, dtype=np.int32), 64), codes64, 64,
            "64 candidates per window: eight nested chains of eight members (SURVEY 8d: C in {1, 8, 64})")
    ctx.eval_upload(w.cw, w.codes, w.sF, w.sR)
    if not cold:
        os.environ["MP_EVAL_TIMING_EVERY"] = "4"
        return out
    # cold: one launch of the headline set with L2 / Infinity Cache flushed by a 2 GiB device copy, no warm-up
    buf = torch.zeros((w.n_cand, 3), dtype=torch.int64, device=dev)
    colds = []
    for _ in range(3):
        scratch_b.copy_(scratch_a)
        torch.cuda.synchronize()
        ctx.eval_timing(reset=True)
        ctx.eval_launch(buf.data_ptr())
        torch.cuda.synchronize()
        ms, _ = ctx.eval_timing(reset=True)
        colds.append(ms)
    out["cold_single_launch"] = {"kernel_ms": colds, "evals_per_s": total * C / (min(colds) * 1e-3),
                                 "what": "headline candidates, ONE launch right after a 2 GiB device copy (planes come from HBM, not from L2 / Infinity Cache), best of 3 listed"}
    os.environ["MP_EVAL_TIMING_EVERY"] = "4"
    return out


PIPELINE_FLAGS = dict(primer_length=18, coverage=0.8, number_of_dege_bases=4, score_of_dege_bases=10, raw_entropy_threshold=3.6, product_len=150,
                      position="2,3,-1", variation=1, distance=4, GC="0.2,0.7", nproc=1)     # tools/make_synth_golden.py: the same


def pipeline_block(lib, local, rows, a, reps=5):
    """pipeline_block_unguarded, or {"error": ...}: a side measurement must not take the headline line with it."""
    try:
        return pipeline_block_unguarded(lib, local, rows, a, reps)
    except (Exception, SystemExit) as e:          # noqa: BLE001 — reported in the line
        return {"rows": int(rows.shape[0]), "cols": int(rows.shape[1]), "error": f"{type(e).__name__}: {e}"}


def pipeline_block_unguarded(lib, local, rows, a, reps=5):
    """NN_degenerate(...).run() — the step the evaluation kernel belongs to — on the workload's own rows: median wall time of `reps`
    runs after one warm-up (each with a fresh context), the phase split of the median run, and the TSV against the checker's (SHA-256
    committed by tools/make_synth_golden.py: the checker takes minutes to hours per size on one core)."""
    from multiprime_amd.core import NN_degenerate
    from multiprime_amd.synth import to_fasta
    n, L = rows.shape
    golden = None
    for e in (load_json(os.path.join("..", "tests", "golden", "synth_pipeline.json")) or {}).get("entries", []):
        if (e["rows"], e["cols"], e["seed"]) == (n, L, a.seed):
            golden = e
    # the FASTA (rows x (cols + ~12) bytes: 1 GB at config 4) goes to memory-backed storage when that has room for it, else to the default
    # temporary directory
    need = int(n) * (int(L) + 16) * 2
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > need else None
    td = tempfile.mkdtemp(prefix="mp_bench_", dir=shm)
    try:
        fa, out = os.path.join(td, "syn.fa"), os.path.join(td, "out.tsv")
        with open(fa, "wb") as f:
            f.write(to_fasta(rows))
        runs = []
        ctx = None                                                    # ONE context for all repetitions, as a --batch worker keeps its own across
        for rep in range(reps + 1):                                   # alignments: its staging area / tables are set up by the first run
            t0 = time.perf_counter()
            app = NN_degenerate(seq_file=fa, outfile=out, library=lib, device=local, write_json=False, keep_bitsets=True, context=ctx, **PIPELINE_FLAGS)
            ctx = app.ctx
            t1 = time.perf_counter()
            app.run()
            t2 = time.perf_counter()
            if rep:                                                   # the first run also warms the process (runtime copy paths, page faults)
                runs.append((t2 - t1, t1 - t0, {key: val for key, val in app.stats.items() if isinstance(val, (int, float))}))
            del app
        ctx.close()
        with open(out, "rb") as f:
            tsv = f.read()
    finally:
        shutil.rmtree(td, ignore_errors=True)
    runs.sort(key=lambda r: r[0])
    run_s, _, stats = runs[len(runs) // 2]
    constructs = sorted(r[1] for r in runs)                       # its own median: a repetition's constructor is not tied to how long its run() took
    construct_s = constructs[len(constructs) // 2]                # (one constructor in ten moves its 1 GB in 0.15 s instead of 0.03 on this pool)
    kern = (load_json("r06_pipeline_kernels.json") or {}).get(f"rows_{n}")
    sha = hashlib.sha256(tsv).hexdigest()
    return {"rows": n, "cols": L, "run_ms": run_s * 1e3, "run_ms_min": runs[0][0] * 1e3, "run_ms_max": runs[-1][0] * 1e3, "construct_ms": construct_s * 1e3,
            "construct_ms_min": constructs[0] * 1e3, "construct_ms_max": constructs[-1] * 1e3, "repetitions": reps, "phases_ms": {key[:-2]: round(val * 1e3, 3) for key, val in stats.items() if key.endswith("_s")},
            "windows": stats.get("n_windows"), "windows_past_the_gates": stats.get("windows_planned"), "candidates": stats.get("n_candidates"),
            "rows_out": stats.get("n_rows"), "tsv_sha256": sha, "oracle_tsv_sha256": golden["tsv_sha256"] if golden else None,
            "tsv_equal_oracle": (sha == golden["tsv_sha256"]) if golden else None,
            "oracle_note": (f"checker: {golden['checker']}, {golden['checker_wall_s']} s" if golden else "no committed checker TSV for this size / seed"),
            "kernels": kern["kernels"] if kern else None,
            "kernels_note": (kern["what"] + " — collected by tools/reproduce.sh pipeline_kernels, not in this run") if kern else "profiles/r06_pipeline_kernels.json absent"}


class OracleBlocks:
    """The plain-C oracle (oracle/mprime_oracle.c) on EVERY host core: one oracle context per thread over a block of the sample's
    rows, built once (untimed, like the GPU's planes); any number of candidate sets are then evaluated on them (the C call releases
    the GIL) and their counters summed over the blocks.  This and python_reference_leg are the only places bench.py touches oracle/."""

    def __init__(self, w, rows, n_threads):
        from multiprime_amd._abi import Library
        so = os.path.join(REPO, "oracle", "_build", "libmprime_oracle.so")
        self.lib = Library(so) if os.path.exists(so) else None
        self.w, self.rows, self.n = w, rows, rows.shape[0]
        self.cores = os.cpu_count() or 1
        T = n_threads or self.cores
        self.T = T = max(1, min(T, self.n // 256))
        self.bounds = [self.n * t // T for t in range(T + 1)]
        self.ctxs, self.universe = [None] * T, [0] * T
        if self.lib is None:
            return
        t0 = time.perf_counter()
        self._threads(self._build)
        self.build_s = time.perf_counter() - t0

    def _threads(self, fn):
        th = [threading.Thread(target=fn, args=(t,)) for t in range(self.T)]
        for x in th:
            x.start()
        for x in th:
            x.join()

    def _build(self, t):
        w = self.w
        blk = self.rows[self.bounds[t]:self.bounds[t + 1]]
        ora = self.lib.context(0)
        ora.load_msa(blk.reshape(-1), np.arange(blk.shape[0] + 1, dtype=np.int64) * w.L)
        n_ex = ora.build_windows(w.p0, w.W, w.k, w.v)
        expand_exceptions(ora, n_ex, w.k, w.v)
        alln = ora.eval_candidates(np.arange(w.W, dtype=np.int32), np.full((w.W, w.k), 15, np.uint8), 0, 0)
        self.universe[t] = alln[:, 0] + alln[:, 1]
        self.ctxs[t] = ora

    def eval(self, cw, codes):
        """(counters summed over the row blocks, wall time of the slowest thread's evaluation call)."""
        res, spans = [None] * self.T, [0.0] * self.T

        def work(t):
            t0 = time.perf_counter()
            res[t] = self.ctxs[t].eval_candidates(cw, codes, self.w.sF, self.w.sR)
            spans[t] = time.perf_counter() - t0

        self._threads(work)
        return np.sum(res, axis=0), max(spans)

    def universe_total(self, windows=None):
        u = np.sum(self.universe, axis=0)
        return int(u.sum() if windows is None else u[windows].sum())

    def close(self):
        for c in self.ctxs:
            if c is not None:
                c.close()
        self.ctxs = []


def cpu_baseline(w, blocks, gpu_counters, seed, one_core=True, python_leg=True):
    """The oracle on the host cores: EVERY core over row blocks of the sample and one core on a bounded sub-sample.  When the sample
    is the whole workload its summed counters are compared with the GPU's, candidate by candidate."""
    if blocks.lib is None:
        return {"value": None, "unit": "evals/s", "cores": 0, "kind": "port", "sample": "oracle library not built", "parity_checked": None}
    W, C, n, T = w.W, w.C, blocks.n, blocks.T
    total, eval_wall = blocks.eval(w.cw, w.codes)
    evals = blocks.universe_total() * C
    parity = None
    if gpu_counters is not None:
        parity = bool(np.array_equal(total, gpu_counters))
    out = {"value": evals / eval_wall, "unit": "evals/s", "cores": T, "kind": "port", "host_cores": blocks.cores,
           "sample": f"all {n} sequences, all {W} windows x {C} candidates = {evals} evals on {T} threads (of {blocks.cores} host cores) in {eval_wall:.2f} s "
                     f"(oracle/mprime_oracle.c; its own tables built beforehand in {blocks.build_s:.1f} s, untimed like the GPU's planes)",
           "parity_checked": parity,
           "parity_note": "per candidate, all three counters, GPU == sum of the oracle's row blocks" if parity is not None else "sample is not the whole workload: no comparison",
           "reference_in_kernel_note": "BASELINE.md section 2: the reference itself (V20, one core - its pool is inert) ran 1.8-2.8e5 evals/s inside mis_primer_check and "
                                       "4.5-7.3e4 evals/s end to end in the authoring container; it cannot run on the GPU box (absent there), `python_reference` restates it"}
    if one_core:            # a bounded sub-sample (first rows)
        L, p0, k, v = w.L, w.p0, w.k, w.v
        n1 = max(256, min(n, 8192))
        ora = blocks.lib.context(0)
        ora.load_msa(blocks.rows[:n1].reshape(-1), np.arange(n1 + 1, dtype=np.int64) * L)
        n_ex = ora.build_windows(p0, W, k, v)
        expand_exceptions(ora, n_ex, k, v)
        alln = ora.eval_candidates(np.arange(W, dtype=np.int32), np.full((W, k), 15, np.uint8), 0, 0)
        ev1 = int((alln[:, 0] + alln[:, 1]).sum()) * C
        t0 = time.perf_counter()
        ora.eval_candidates(w.cw, w.codes, w.sF, w.sR)
        dt1 = time.perf_counter() - t0
        ora.close()
        out["one_core"] = {"value": ev1 / dt1, "cores": 1, "sample": f"first {n1} sequences, {ev1} evals in {dt1:.2f} s"}
    if python_leg:
        out["python_reference"] = python_reference_leg(w, blocks.lib, seed)
    return out


def python_reference_leg(w, oracle_lib, seed, n_rows=2000, budget_s=8.0):
    """The reference's own evaluation (mis_primer_check / Y_distance, V20:1103-1130 / 229-233) restated in Python the way the
    reference computes it, ONE core (its process pool is inert), on gap-free rows of the same generator for as many
    windows as fit `budget_s`; its counts are checked against the plain-C oracle on the same rows."""
    from multiprime_amd import iupac
    from oracle.py_reference_eval import mis_primer_check
    L, p0, k, v, C = w.L, w.p0, w.k, w.v, w.C
    rows = synth_rows(0, n_rows, L, seed, p_gap=0.0, edge_frac=0.0, p_iupac=0.0)
    text = [r.tobytes().decode() for r in rows]
    f_set, r_set = {2, 3, k}, {2, k - 3, k - 2}
    got, wins = [], []
    evals = 0
    t0 = time.perf_counter()
    for win in range(0, w.W, 37):
        cover = {}
        for s in text:
            km = s[p0 + win:p0 + win + k]
            cover[km] = cover.get(km, 0) + 1
        universe = set(cover)
        for c in range(C):
            primer = iupac.strings_of(iupac.SYMBOL_LUT[w.codes[win * C + c][None, :]])[0]
            got.append(mis_primer_check(universe, primer, cover, v, f_set, r_set))
            evals += n_rows
        wins.append(win)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    ora = oracle_lib.context(0)
    ora.load_msa(rows.reshape(-1), np.arange(n_rows + 1, dtype=np.int64) * L)
    ora.build_windows(p0, w.W, k, v)
    sel = np.concatenate([np.arange(win * C, win * C + C) for win in wins])
    want = ora.eval_candidates(w.cw[sel], w.codes[sel], w.sF, w.sR)
    return {"value": evals / dt, "unit": "evals/s", "cores": 1, "kind": "port of the reference's Python (oracle/py_reference_eval.py)",
            "sample": f"{n_rows} gap-free sequences of the same generator, {len(wins)} windows x {C} candidates = {evals} evals in {dt:.1f} s "
                      f"(dict construction included, as in the reference)",
            "equals_oracle": bool(np.array_equal(np.asarray(got, np.int64), want))}


