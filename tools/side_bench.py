#!/usr/bin/env python3
"""The steps either side of the core step, measured the way bench.py measures the headline:

  dimer_scan   all-pairs 3'-end dimer scan (mp_dimer_scan mode 0; finDimer_V4.py:191-224) on 2024 and 20 000 random primers
  pcr_scan     exact in-silico PCR (extract_PCR_product_V1.py:189-216) on a synthetic 20 727 x ~1.95 kb database, 64 primer pairs:
               the byte-scanning call (text sent and packed per call) and the resident store (mp_seq_load once, mp_pcr_scan_resident)
  kmm_scan     k-mismatch primer-site scan (primer_coverage_validation_by_BWT_V9.py:264-300) on the same database, one primer pair's
               expansions: bytes per call and resident

Each block: wall time of the call (median of `reps`), the bytes an ideal kernel of the formulation must move (`bytes_compulsory`),
`frac` = those bytes / time / 8 TB/s (these scans are integer-VALU bound: the fraction says how far from a memory bound they are, not
how good they are), rocprofv3 numbers merged from profiles/r06_side_kernels.json when that was collected for this source, the
CHECKER (oracle/mprime_oracle.c, one core) on a bounded sample of the same input, and `parity_checked` (GPU == checker on that sample).

`python tools/side_bench.py` prints the blocks as one JSON line; bench.py imports run()."""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
HBM_PEAK_GBS = 8000.0


def _median_ms(fn, reps):
    fn()                                                        # warm-up: allocations, first launch
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        t.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(t)), out


def _oracle():
    from multiprime_amd._abi import Library
    so = os.path.join(REPO, "oracle", "_build", "libmprime_oracle.so")
    return Library(so) if os.path.exists(so) else None


def random_primers(rng, n):
    out = []
    for _ in range(n):
        L = int(rng.integers(18, 25))
        s = ["ACGT"[int(x)] for x in rng.integers(0, 4, size=L)]
        for p in rng.integers(0, L, size=2):
            if rng.random() < 0.5:
                s[int(p)] = "RYMKSW"[int(rng.integers(0, 6))]
        out.append("".join(s))
    return out


def dimer_block(ctx, ora, sizes=(2024, 20000), reps=3, oracle_primers=300):
    from multiprime_amd import dimer
    rng = np.random.default_rng(7)
    loss, dg, lim = dimer.cached_loss_table(3.96), dimer.dg_params(), dimer.dg_limit()
    res = {"what": "mp_dimer_scan mode 0 (finDimer_V4.py:191-224): every unordered pair of n random 18-24 nt primers, <= 2 IUPAC positions each, Loss >= 3.96"}
    for n in sizes:
        codes, off = dimer.encode_primers(random_primers(rng, n))
        ms, hits = _median_ms(lambda: ctx.dimer_scan(codes, off, 0, 0, loss, dg, lim, cap=1 << 21), reps)
        pairs = n * (n + 1) // 2
        compulsory = n * 80.0 + len(hits) * 24.0                 # the primers' bit planes in, the hit records out
        res[f"primers_{n}"] = {"pairs": pairs, "call_ms": ms, "pairs_per_s": pairs / (ms * 1e-3), "hits": int(len(hits)),
                               "bytes_compulsory": compulsory, "frac": compulsory / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bound": "valu"}
    if ora is not None:
        n = oracle_primers
        codes, off = dimer.encode_primers(random_primers(rng, n))
        o = ora.context(0)
        t0 = time.perf_counter()
        want = o.dimer_scan(codes, off, 0, 0, loss, dg, lim, cap=1 << 20)
        dt = time.perf_counter() - t0
        o.close()
        got = ctx.dimer_scan(codes, off, 0, 0, loss, dg, lim, cap=1 << 20)
        res["cpu_baseline"] = {"value": n * (n + 1) // 2 / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"{n} primers of the same generator, {n * (n + 1) // 2} pairs in {dt:.2f} s"}
        res["parity_checked"] = bool(np.array_equal(got, want))
    return res


def synthetic_database(rows, cols):
    from multiprime_amd.synth import synth_block, synth_root
    blk = [synth_block(r0, min(4096, rows - r0), cols, 20250303, p_iupac=0.0) for r0 in range(0, rows, 4096)]
    seqs = [r[r != ord("-")].tobytes() for b in blk for r in b]
    off = np.zeros(rows + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    root = np.frombuffer(b"ACGT", np.uint8)[synth_root(cols, 20250303)]
    return np.frombuffer(b"".join(seqs), np.uint8), off, root


def primer_pairs(root, n_pairs, seed=3):
    from multiprime_amd import iupac
    rng = np.random.default_rng(seed)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    codes, poff = [], [0]
    for _ in range(n_pairs):
        f0 = int(rng.integers(0, len(root) - 700))
        r0 = f0 + int(rng.integers(150, 600))
        f = iupac.MASK_LUT[root[f0:f0 + 18]].copy()
        r = iupac.MASK_LUT[np.array([comp[c] for c in root[r0:r0 + 18][::-1]], np.uint8)].copy()
        f[int(rng.integers(0, 18))] |= np.uint8(1 << rng.integers(0, 4))       # one degenerate position each
        r[int(rng.integers(0, 18))] |= np.uint8(1 << rng.integers(0, 4))
        codes += [f, r]
        poff += [poff[-1] + 18, poff[-1] + 36]
    return np.concatenate(codes).astype(np.uint8), np.asarray(poff, np.int32)


def scan_blocks(ctx, ora, rows=20727, cols=1951, n_pairs=64, reps=3, oracle_rows=384):
    from multiprime_amd import iupac
    data, off, root = synthetic_database(rows, cols)
    bases = float(off[-1])
    codes, poff = primer_pairs(root, n_pairs)
    out_bytes = n_pairs * rows * 16.0
    pcr = {"what": f"exact in-silico PCR (extract_PCR_product_V1.py:189-216): {rows} sequences x {bases / rows:.0f} bases (synthetic, gap-free rows of the bench "
                   f"generator), {n_pairs} primer pairs of 18 nt with one IUPAC position each", "rows": rows, "bases": bases, "pairs": n_pairs}
    ms_b, out_b = _median_ms(lambda: ctx.pcr_scan(data, off, codes, poff), reps)
    t0 = time.perf_counter()
    ctx.seq_load(data, off)
    load_ms = (time.perf_counter() - t0) * 1e3
    ms_r, out_r = _median_ms(lambda: ctx.pcr_scan_resident(codes, poff), reps)
    for name, ms, comp in (("bytes_per_call", ms_b, bases + out_bytes), ("resident", ms_r, bases / 2 + out_bytes)):
        pcr[name] = {"call_ms": ms, "pair_x_sequence_per_s": rows * n_pairs / (ms * 1e-3), "bases_x_pairs_per_s": bases * n_pairs / (ms * 1e-3),
                     "bytes_compulsory": comp, "frac": comp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bound": "valu"}
    pcr["resident"]["seq_load_ms"] = load_ms
    pcr["resident"]["store_device_bytes"] = ctx.seq_info()[2]
    pcr["speedup_resident"] = ms_b / ms_r
    pcr["resident_equals_bytes"] = bool(np.array_equal(out_b, out_r))
    pcr["amplified_fraction"] = float((out_r.reshape(-1, 4)[:, 0] >= 0).mean())
    # k-mismatch scan: the expansions of the first primer pair, bowtie2's budget for 18 nt (1 mismatch... floor((0.6 + 0.6 L) / 6) = 1), 3'-term 5
    exp = []
    for q in range(2):
        s = iupac.strings_of(iupac.SYMBOL_LUT[codes[poff[q]:poff[q + 1]][None, :]])[0]
        exp += iupac.expand(s)
    pc = iupac.MASK_LUT[np.frombuffer("".join(exp).encode(), np.uint8)]
    po = np.arange(len(exp) + 1, dtype=np.int32) * 18
    kmm = {"what": f"k-mismatch primer-site scan (primer_coverage_validation_by_BWT_V9.py:264-300) on the same database: {len(exp)} reads (the expansions of "
                   f"one primer pair), both strands, <= 1 mismatch, 3'-term 5", "rows": rows, "bases": bases, "reads": len(exp)}
    ms_b, h_b = _median_ms(lambda: ctx.kmm_scan(data, off, pc, po, 1, 5, cap=1 << 21), reps)
    ms_r, h_r = _median_ms(lambda: ctx.kmm_scan_resident(pc, po, 1, 5, cap=1 << 21), reps)
    for name, ms, comp in (("bytes_per_call", ms_b, bases + len(h_b) * 16.0), ("resident", ms_r, bases / 2 + len(h_r) * 16.0)):
        kmm[name] = {"call_ms": ms, "positions_x_reads_x_strands_per_s": bases * len(exp) * 2 / (ms * 1e-3), "hits": int(len(h_r)),
                     "bytes_compulsory": comp, "frac": comp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bound": "valu"}
    kmm["speedup_resident"] = ms_b / ms_r
    kmm["resident_equals_bytes"] = bool(np.array_equal(h_b, h_r))
    if ora is not None:
        n = min(oracle_rows, rows)
        o = ora.context(0)
        t0 = time.perf_counter()
        want = o.pcr_scan(data[: off[n]], off[: n + 1], codes, poff)
        dt = time.perf_counter() - t0
        pcr["cpu_baseline"] = {"value": n * n_pairs / dt, "unit": "pair x sequence/s", "cores": 1, "kind": "port",
                               "sample": f"the first {n} sequences, all {n_pairs} pairs, in {dt:.2f} s"}
        pcr["parity_checked"] = bool(np.array_equal(want, out_r[:, :n]))
        t0 = time.perf_counter()
        want = o.kmm_scan(data[: off[n]], off[: n + 1], pc, po, 1, 5, cap=1 << 20)
        dt = time.perf_counter() - t0
        o.close()
        kmm["cpu_baseline"] = {"value": float(off[n]) * len(exp) * 2 / dt, "unit": "positions x reads x strands/s", "cores": 1, "kind": "port",
                               "sample": f"the first {n} sequences, all {len(exp)} reads, in {dt:.2f} s"}
        kmm["parity_checked"] = bool(np.array_equal(want, h_r[h_r[:, 0] < n]))
    ctx.seq_free()
    return pcr, kmm


def merge_kernel_numbers(blocks):
    """rocprofv3 numbers of the same calls (tools/reproduce.sh side_kernels -> profiles/r06_side_kernels.json), when collected."""
    try:
        with open(os.path.join(REPO, "profiles", "r06_side_kernels.json")) as f:
            k = json.load(f)
    except (OSError, ValueError):
        return
    for name, block in blocks.items():
        if name in k:
            block["kernels"] = k[name]


def run(lib, device=0, reps=3):
    ctx = lib.context(device)
    ora = _oracle()
    try:
        out = {"dimer_scan": dimer_block(ctx, ora, reps=reps)}
        out["pcr_scan"], out["kmm_scan"] = scan_blocks(ctx, ora, reps=reps)
    finally:
        ctx.close()
    merge_kernel_numbers(out)
    return out


if __name__ == "__main__":
    import torch  # noqa: F401
    from multiprime_amd._abi import Library
    print(json.dumps(run(Library())))
