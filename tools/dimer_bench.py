#!/usr/bin/env python3
"""Times the all-pairs 3'-end dimer scan (mp_dimer_scan mode 0, finDimer_V4.py:191-224) on n random 18-24 nt primers
with a few degenerate positions each (BASELINE config 3: "finDimer all-pairs dG" at database scale).  One JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multiprime_amd import dimer  # noqa: E402
from multiprime_amd._abi import Library  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--primers", type=int, nargs="*", default=[2024, 8000, 20000])
    ap.add_argument("--oracle-primers", type=int, default=300)
    a = ap.parse_args()
    rng = np.random.default_rng(7)
    ctx = Library().context(0)
    loss, dg, lim = dimer.cached_loss_table(3.96), dimer.dg_params(), dimer.dg_limit()

    def primers(n):
        out = []
        for _ in range(n):
            L = int(rng.integers(18, 25))
            s = ["ACGT"[int(x)] for x in rng.integers(0, 4, size=L)]
            for p in rng.integers(0, L, size=2):
                if rng.random() < 0.5:
                    s[int(p)] = "RYMKSW"[int(rng.integers(0, 6))]
            out.append("".join(s))
        return out

    res = []
    for n in a.primers:
        codes, off = dimer.encode_primers(primers(n))
        ctx.dimer_scan(codes[: off[50]], off[:51], 0, 0, loss, dg, lim)            # warm-up
        t0 = time.time()
        hits = ctx.dimer_scan(codes, off, 0, 0, loss, dg, lim, cap=1 << 20)
        dt = time.time() - t0
        pairs = n * (n + 1) // 2
        res.append({"primers": n, "pairs": pairs, "wall_ms": round(dt * 1e3, 1), "pairs_per_s": pairs / dt, "hits": int(len(hits))})
    if a.oracle_primers:
        ora = Library(os.path.join(REPO, "oracle", "_build", "libmprime_oracle.so")).context(0)
        n = a.oracle_primers
        codes, off = dimer.encode_primers(primers(n))
        t0 = time.time()
        h2 = ora.dimer_scan(codes, off, 0, 0, loss, dg, lim, cap=1 << 20)
        dt = time.time() - t0
        h1 = ctx.dimer_scan(codes, off, 0, 0, loss, dg, lim, cap=1 << 20)
        res.append({"oracle_primers": n, "oracle_pairs_per_s": n * (n + 1) // 2 / dt, "agrees": bool(np.array_equal(h1, h2))})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
