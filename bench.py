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

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
FULL_ROWS = 1048576            # BASELINE.json configs[3]: 1M sequences
SHARD_ROWS = 131072            # what one GPU holds of it in the 8-GPU job
# measured ceilings (tools/ubench.hip on the same GPU pool, profiles/r02_ubench.json); used when that file is absent
DEFAULT_CEILINGS = {"valu_wave_instr_per_s_per_simd": 8.5e8, "l2_read_GBs": 31559.0, "source": "built-in defaults (profiles/r02_ubench.json missing)"}
COUNTER_FILES = ("r06_counters.json", "r05_counters.json", "r04_counters.json", "r03_counters.json")
# sources of the timed kernel (eval_chain_kernel and what it includes): the key of a counter entry
KERNEL_SOURCES = ("eval.hip", "evalprog.hip", "evalslide.hip", "slidecore.hpp", "slideplan.hpp", "evalslide.hpp", "chainbody.hpp", "bitslice.hpp", "common.hpp",
                  "winwords.hpp", "evalprog.hpp")
CONFIG4_CHECKSUM = [4933256386, 2131385189, 2001280469]      # counter_checksum of the default workload (N = 1, oracle-checked: profiles/r04_bench.json on)
SLIDE_FROM_ROWS = 262145     # evalslide.hip (upload_eval_slide): above 262144 (padded) rows the chains are evaluated by sliding
PROG_FROM_ROWS = 393216      # eval.hip (mp_eval_upload): the program-driven first-pass kernel, when sliding is switched off

KERNELS = {
    "chain": "eval_chain_kernel (bit-sliced one-hot column planes, nested refinement chains; patch rows ride in the same launch)",
    "table": "eval_bits_kernel (bit-sliced one-hot column planes, symbol table per position; patch rows ride in the same launch)",
    "rows": "eval_kernel (row-per-lane, window words derived from the planes)",
    "prog": "eval_prog_kernel (eval_chain_kernel's arithmetic, host-written fetch programs, buffer loads, event planes parked in LDS; patch rows ride in the same launch)",
    "slide": "eval_slide_kernel (5-bit bit-sliced mismatch count of a per-column reference sliding along the windows, event planes fetched once per chain; "
             "patch rows on eval_chain_kernel in the same step)",
}


def eval_mode(n_rows=0, ctx=None):
    """Which evaluation kernel runs the nested chains: what the library says about the staged candidates (mp_eval_plan_info), else
    what its environment switches select (defaults: sliding from SLIDE_FROM_ROWS rows up, the nested-chain kernel below)."""
    if ctx is not None and ctx.eval_plan_info()["sliding_items"]:
        return "slide"
    if os.environ.get("MP_EVAL_MODE") == "rows":
        return "rows"
    if os.environ.get("MP_EVAL_BITS", "0") in ("1", "2") or os.environ.get("MP_EVAL_GROUP") == "plain":
        return "table"
    if os.environ.get("MP_EVAL_SLIDE", "1" if n_rows >= SLIDE_FROM_ROWS else "0") == "1":
        return "slide"
    if os.environ.get("MP_EVAL_PROG", "1" if n_rows >= PROG_FROM_ROWS else "0") == "1":
        return "prog"
    return "chain"


def kernel_source_hash():
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(REPO, "multiprime_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def make_candidates(root_codes, p0, W, k, C, seed, nested=True):
    """C candidates per window.  nested: the root k-mer of the window, then progressively more degenerate versions (one
    more random base at one more random position each), seeded per window — the shape of a refinement chain (SURVEY §8d
    micro-benchmark).  not nested: C unrelated candidates — the root with ONE random extra base at a random position
    each, so that no candidate accepts a subset of another's k-mers."""
    rng = np.random.default_rng(seed)
    cw = np.repeat(np.arange(W, dtype=np.int32), C)
    codes = np.empty((W, C, k), np.uint8)
    root = np.stack([root_codes[p0 + w: p0 + w + k] for w in range(W)]).astype(np.uint8)
    cur = root
    codes[:, 0] = cur
    for c in range(1, C):
        pos = rng.integers(0, k, size=W) if nested else (rng.integers(0, k, size=W) + c) % k
        add = (1 << rng.integers(0, 4, size=W)).astype(np.uint8)
        cur = (cur if nested else root).copy()
        cur[np.arange(W), pos] |= add
        codes[:, c] = cur
    return cw, codes.reshape(W * C, k)


def expand_exceptions(ctx, n_ex, k, v):
    """IUPAC windows of the shard -> concrete extra rows (native expansion, reference order)."""
    from multiprime_amd import host, iupac
    if not n_ex:
        return 0
    ew, er, ec = ctx.get_exceptions(n_ex)
    sel = (ec == 0).sum(axis=1) <= v
    if not sel.any():
        return 0
    exp, src = host.expand_kmers(ec[sel])
    ctx.set_extra_rows(ew[sel][src], iupac.words_of_codes(exp))
    return len(exp)


def synth_rows(row0, n, L, seed, **kw):
    """Rows [row0, row0 + n) of the synthetic alignment; blocks are seeded independently, so they are generated on several threads."""
    from multiprime_amd.synth import synth_block
    step = 32768
    if n <= step:
        return synth_block(row0, n, L, seed, **kw)
    out = np.empty((n, L), np.uint8)

    def part(s):
        m = min(step, n - s)
        out[s:s + m] = synth_block(row0 + s, m, L, seed, **kw)

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        list(ex.map(part, range(0, n, step)))
    return out


def load_json(name):
    try:
        with open(os.path.join(REPO, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def ceilings():
    ub = load_json("r02_ubench.json")
    if not ub:
        return dict(DEFAULT_CEILINGS)
    try:
        valu = max(r["wave_instr_per_s_per_simd"] for r in ub["valu"] if r["op"] in ("v_bitop3_b32", "v_add_u32", "v_xor_b32"))
        l2 = max(r["GBs"] for r in ub["reads"] if r["case"].startswith("l2_"))
        hbm = max(r["GBs"] for r in ub["reads"] if r["case"].startswith("hbm"))
        return {"valu_wave_instr_per_s_per_simd": valu, "l2_read_GBs": l2, "hbm_read_GBs_measured": hbm,
                "source": "profiles/r02_ubench.json (tools/ubench.hip: saturated v_bitop3/v_add issue rate, 8 waves per SIMD; L2-resident dwordx4 reads)"}
    except (KeyError, ValueError):
        return dict(DEFAULT_CEILINGS)


def counters_for(cfg):
    """(entry, note): PMC counters per launch of the timed kernel for exactly this configuration AND this kernel source, or
    (None, why not)."""
    want = kernel_source_hash()
    stale = None
    for name in COUNTER_FILES:
        db = load_json(name)
        for e in (db or {}).get("entries", []):
            if all(e.get(key) == val for key, val in cfg.items()):
                if e.get("source_hash") == want:
                    return e, f"profiles/{name}, source_hash {want}"
                stale = f"profiles/{name} holds counters of kernel source {e.get('source_hash')}, the library is built from {want}: fractions dropped, re-run tools/collect_counters.py"
    return None, stale or "no counters collected for this configuration"


def time_launches(ctx, torch, out_ptr, n, warm):
    """Median / max / mean HIP-event duration (ms) of n launches of the staged candidate set (every launch timed)."""
    for _ in range(warm):
        ctx.eval_launch(out_ptr)
    torch.cuda.synchronize()
    ctx.eval_timing(reset=True)
    for _ in range(n):
        ctx.eval_launch(out_ptr)
    torch.cuda.synchronize()
    ms, cnt = ctx.eval_timing(reset=True)
    s = np.sort(ctx.eval_timing_samples())
    return {"mean_ms": ms / max(cnt, 1), "median_ms": float(s[len(s) // 2]) if len(s) else None,
            "max_ms": float(s[-1]) if len(s) else None, "launches": cnt}


class Workload:
    """The evaluation workload on `n_rows` sequences starting at global row `row0`, resident on the device."""

    def __init__(self, lib, local, torch, row0, n_rows, a, win_part=(0, 1), rows=None):
        """`win_part` = (g, G): the g-th of G contiguous groups of the windows (2-D shards: rows x windows, dist.ShardGrid);
        `rows`: the synthetic rows when the caller holds them already."""
        t0 = time.time()
        self.k, self.v, self.C, self.L, self.n_rows = a.k, a.v, a.cands, a.cols, n_rows
        self.ctx = lib.context(local)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.rows = rows if rows is not None else synth_rows(row0, n_rows, self.L, a.seed)
        self.ctx.load_msa(self.rows.reshape(-1), np.arange(n_rows + 1, dtype=np.int64) * self.L)
        k = self.k
        w_all = self.L - 32 - k                                     # the same windows on every rank of a row group ...
        g, G = win_part                                             # ... cut into G contiguous groups along the window axis
        lo, hi = w_all * g // G, w_all * (g + 1) // G
        self.p0, self.W, self.win_part = 16 + lo, hi - lo, (g, G)
        n_ex = self.ctx.build_windows(self.p0, self.W, k, self.v)
        self.n_extra = expand_exceptions(self.ctx, n_ex, k, self.v)
        from multiprime_amd.synth import synth_root
        self.root_codes = np.array([1, 2, 4, 8], np.uint8)[synth_root(self.L, a.seed)]
        # the candidates of ALL windows (seeded per window set), this group's slice of them: the same candidate for a window whatever the shape
        _, codes_all = make_candidates(self.root_codes, 16, w_all, k, self.C, a.seed)
        self.cw, self.codes = np.repeat(np.arange(self.W, dtype=np.int32), self.C), np.ascontiguousarray(codes_all[lo * self.C:hi * self.C])
        f_set, r_set = {2, 3, k}, {2, k - 3, k - 2}                 # -c 2,3,-1 (multiPrime.yaml), get_Y V20:1091
        self.sF = sum(1 << y for y in f_set if 0 <= y < k)
        self.sR = sum(1 << y for y in r_set if 0 <= y < k)
        self.n_cand = len(self.cw)
        # universe size per window (sequences with <= v gaps, plus expansions): an all-N candidate matches every non-gap
        # symbol, so perfect + F_mis under empty strict masks counts it
        alln = self.ctx.eval_candidates(np.arange(self.W, dtype=np.int32), np.full((self.W, k), 15, np.uint8), 0, 0)
        self.universe = alln[:, 0] + alln[:, 1]
        self.evals = int(self.universe.sum()) * self.C
        self.ctx.eval_upload(self.cw, self.codes, self.sF, self.sR)
        self.setup_s = time.time() - t0

    def describe(self):
        part = "" if self.win_part[1] == 1 else f" (window group {self.win_part[0] + 1} of {self.win_part[1]})"
        return (f"synthetic MSA {self.n_rows} x {self.L}, k={self.k}, v={self.v}, {self.C} candidates/window, {self.W} windows{part}, "
                f"strict -c 2,3,-1")


def roofline_block(w, per_launch_ms, samples, kern_n, every, mode):
    kern_s = per_launch_ms * 1e-3
    alg_bytes = w.evals * 3 * w.k / 8.0            # 3k/8 bytes per evaluation
    ceil = ceilings()
    cfg = {"rows": w.n_rows, "cols": w.L, "k": w.k, "v": w.v, "cands": w.C, "mode": mode}
    pmc, note = counters_for(cfg)
    fr = {"valu": None, "l2": None, "hbm": None}
    traffic = None
    if pmc:
        if pmc.get("valu_insts"):
            fr["valu"] = pmc["valu_insts"] / (kern_s * N_SIMD * ceil["valu_wave_instr_per_s_per_simd"])
        if pmc.get("l2_read_bytes"):
            fr["l2"] = pmc["l2_read_bytes"] / kern_s / 1e9 / ceil["l2_read_GBs"]
        if pmc.get("hbm_read_bytes") is not None:
            traffic = pmc["hbm_read_bytes"] + (pmc.get("hbm_write_bytes") or 0)
            fr["hbm"] = traffic / kern_s / 1e9 / HBM_PEAK_GBS
    known = {key: val for key, val in fr.items() if val is not None}
    bound = max(known, key=known.get) if known else None
    # `frac` = the USEFUL HBM fraction: the compulsory bytes of a pass (the packed planes read once, N L 3/8) / kernel time / 8 TB/s —
    # what an ideal kernel of this formulation would have to move; `hbm_frac` = what the kernel DID move (fabric bytes from the counters) over the
    # same time and peak (over-fetch = traffic / compulsory_bytes); `bound` names the largest of the counter fractions
    compulsory = w.n_rows * w.L * 3 / 8.0
    achieved = compulsory / kern_s / 1e9
    return {"bound": bound, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "compulsory_frac": achieved / HBM_PEAK_GBS, "over_fetch": (traffic / compulsory) if traffic is not None else None,
            "achieved_counter_GBs": traffic / kern_s / 1e9 if traffic is not None else None,
            "valu_frac": fr["valu"], "l2_frac": fr["l2"], "hbm_frac": fr["hbm"],
            "compulsory_bytes": compulsory, "compulsory_note": "SURVEY 8d (ii): packed planes read once, N L 3/8 bytes; `achieved` and `frac` are these bytes over the kernel time",
            "algorithmic_frac": alg_bytes / kern_s / 1e9 / HBM_PEAK_GBS,
            "algorithmic_note": "SURVEY 8d figure: 3k/8 bytes per evaluation over the kernel time vs 8 TB/s; exceeds 1 because 8 nested candidates and "
                                "18 overlapping windows share every loaded plane word — NOT a roofline fraction, kept for comparison with round 1",
            "algorithmic_bytes": alg_bytes, "algorithmic_bytes_per_eval": 3 * w.k / 8.0,
            "bound_note": "valu = vector instructions over the measured issue rate (scalar instructions share the slots: profiles/r03_ubench_salu.json); "
                          "l2 = bytes the vector memory path returns to registers over 31.4 TB/s = 256 CUs x 64 B/clk (what bound the first-pass kernels; "
                          "the sliding kernel returns a quarter of their bytes); hbm = fabric bytes (L2 misses: Infinity Cache or HBM) over 8 TB/s",
            "counters": pmc, "counters_source": note, "counters_stale": pmc is None, "source_hash": kernel_source_hash(), "ceilings": ceil,
            "kernel": KERNELS[mode] + "; timed region = counter memset + every kernel of the step", "eval_mode": mode,
            "kernel_ms": per_launch_ms, "kernel_ms_median": float(samples[len(samples) // 2]) if len(samples) else None,
            "kernel_ms_max": float(samples[-1]) if len(samples) else None,
            "launches_timed": kern_n, "timed_every": every}


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
        if backend == "nccl":
            try:
                raw = ctx.comm_unique_id() if rank == 0 else bytes(128)
                box = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
                dist.broadcast(box, src=0)
                ctx.comm_init(world, rank, bytes(box.cpu().numpy().tobytes()))
                seen = ctx.comm_describe()
                one = ctx.comm_sum(np.ones(1, np.int64))
                comm_info.update({"rccl_ranks_seen": seen[0], "rccl_rank": seen[1], "rccl_library": seen[2],
                                  "sum_of_ones": int(one[0])})
                ctx.comm_destroy()
            except Exception as e:                                      # reported, not fatal: the timed steps above ran on torch's RCCL
                comm_info["library_communicator_error"] = str(e)

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
            blocks = None
            if not a.no_cpu:
                n_cpu = a.cpu_rows or rows_per_gpu
                blocks = OracleBlocks(w, w.rows[:n_cpu], a.cpu_threads)
                res["cpu_baseline"] = cpu_baseline(w, blocks, gpu_counters if n_cpu == rows_per_gpu else None, a.seed)
                res["parity_checked"] = res["cpu_baseline"].get("parity_checked")
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
            if not a.no_pipeline:
                res["pipeline"] = {"what": "the REAL step: the drop-in class NN_degenerate(...).run() (multiprime_amd/core.py; --no-json, coverage bitsets kept on the "
                                           "device) on the same synthetic rows, median of 5 after one warm-up; `run_ms` is run() alone (the context is kept across repetitions, as the "
                                           "--batch workers keep theirs across alignments), `construct_ms` the constructor before it (FASTA parse of a file in /dev/shm + mp_load_msa); TSV compared with the CHECKER's "
                                           "(oracle/core_ref.py over the plain-C oracle, tests/golden/synth_pipeline.json)",
                                   f"rows_{rows_per_gpu}": pipeline_block(lib, local, w.rows, a)}
    # N = 1: the shard one GPU holds in the 8-GPU job, timed the same way (with the variant measurements)
    if rank == 0 and world == 1 and not a.no_shard and rows_per_gpu != SHARD_ROWS:
        del sb
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
    if rank == 0 and world == 1 and not a.no_side:
        try:
            sys.path.insert(0, os.path.join(REPO, "tools"))
            import side_bench
            res["side_steps"] = side_bench.run(lib, local)
            if any(isinstance(b, dict) and b.get("parity_checked") is False for b in res["side_steps"].values()):
                res["parity_checked"] = False
        except Exception as e:                  # noqa: BLE001 — a side measurement must not take the headline line with it
            res["side_steps"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        emit(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
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
           "parity_checked", "ms_per_step_bucketed", "bucket", "steps_in_flight", "ms_per_step_one_stream")
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
    run_s, construct_s, stats = runs[len(runs) // 2]
    kern = (load_json("r06_pipeline_kernels.json") or {}).get(f"rows_{n}")
    sha = hashlib.sha256(tsv).hexdigest()
    return {"rows": n, "cols": L, "run_ms": run_s * 1e3, "run_ms_min": runs[0][0] * 1e3, "run_ms_max": runs[-1][0] * 1e3, "construct_ms": construct_s * 1e3,
            "repetitions": reps, "phases_ms": {key[:-2]: round(val * 1e3, 3) for key, val in stats.items() if key.endswith("_s")},
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


if __name__ == "__main__":
    main()
