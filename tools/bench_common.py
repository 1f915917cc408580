#!/usr/bin/env python3
"""What bench.py and its side measurements share (round 6: split out of bench.py, which keeps the contract — the timed region, the
headline line — and nothing else): the synthetic workload on the device, candidate sets, roofline arithmetic, counter look-up."""
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


