#!/usr/bin/env python3
"""Times mp_pcr_scan (exact in-silico PCR, extract_PCR_product_V1.py:189-216) on a synthetic unaligned database:
`--rows` sequences of ~`--cols` bases (a mutated common root with the gaps removed), `--pairs` primer pairs cut from
the root (one IUPAC symbol each), i.e. BASELINE config 3's shape (20 727 x ~2 kb).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multiprime_amd import iupac  # noqa: E402
from multiprime_amd._abi import Library  # noqa: E402
from multiprime_amd.synth import synth_block, synth_root  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20727)
    ap.add_argument("--cols", type=int, default=1951)
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--oracle", action="store_true", help="time the CPU oracle on the first 512 rows as well")
    a = ap.parse_args()
    rows = synth_block(0, a.rows, a.cols, 20250303, p_iupac=0.0)
    seqs = [r[r != ord("-")].tobytes() for r in rows]
    off = np.zeros(a.rows + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    data = np.frombuffer(b"".join(seqs), np.uint8)
    root = np.frombuffer(b"ACGT", np.uint8)[synth_root(a.cols, 20250303)]
    rng = np.random.default_rng(3)
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    codes, poff = [], [0]
    for _ in range(a.pairs):
        f0 = int(rng.integers(0, a.cols - 700))
        r0 = f0 + int(rng.integers(150, 600))
        f = iupac.MASK_LUT[root[f0:f0 + 18]].copy()
        r = iupac.MASK_LUT[np.array([comp[c] for c in root[r0:r0 + 18][::-1]], np.uint8)].copy()
        f[int(rng.integers(0, 18))] |= np.uint8(1 << rng.integers(0, 4))       # one degenerate position each
        r[int(rng.integers(0, 18))] |= np.uint8(1 << rng.integers(0, 4))
        codes += [f, r]
        poff += [poff[-1] + 18, poff[-1] + 36]
    codes = np.concatenate(codes).astype(np.uint8)
    poff = np.asarray(poff, np.int32)
    ctx = Library().context(0)
    ctx.pcr_scan(data, off, codes, poff)                                         # warm-up (allocations, first launch)
    t0 = time.time()
    out = ctx.pcr_scan(data, off, codes, poff)
    dt = time.time() - t0
    res = {"rows": a.rows, "mean_len": float(np.diff(off).mean()), "pairs": a.pairs, "wall_ms": round(dt * 1e3, 2),
           "pair_x_sequence_per_s": a.rows * a.pairs / dt, "bases_scanned_per_s": float(off[-1]) * a.pairs / dt,
           "amplified_fraction": float((out.reshape(-1, 4)[:, 0] >= 0).mean())}
    if a.oracle:
        ora = Library(os.path.join(REPO, "oracle", "_build", "libmprime_oracle.so")).context(0)
        n = min(512, a.rows)
        t0 = time.time()
        o2 = ora.pcr_scan(data[: off[n]], off[: n + 1], codes, poff)
        dt2 = time.time() - t0
        res["oracle_pair_x_sequence_per_s"] = n * a.pairs / dt2
        res["oracle_agrees_on_sample"] = bool((o2.reshape(a.pairs, n, 4) == out.reshape(a.pairs, a.rows, 4)[:, :n]).all())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
