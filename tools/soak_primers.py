#!/usr/bin/env python3
"""Randomised soak of the primer-level kernels against the CPU oracle (GPU box): mp_dimer_scan (finDimer and
get_Maxprimerset modes), mp_dimer_pairs, mp_pair_coverage and mp_pcr_scan on random degenerate primers, bitsets and
sequence sets, for `--seconds`.  One JSON line; exit code 1 and the seed of the case on the first difference."""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from multiprime_amd import dimer, iupac  # noqa: E402
from multiprime_amd._abi import Library  # noqa: E402
from test_dimer import random_primers  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    hip = Library().context(0)
    ora = Library(os.path.join(REPO, "oracle", "_build", "libmprime_oracle.so")).context(0)
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    n = {"dimer_scan": 0, "dimer_pairs": 0, "pair_coverage": 0, "pcr_scan": 0, "dimer_cases_thread_per_pair": 0}
    while time.time() < t_end:
        seed = int(rng.integers(1 << 30))
        # every case picks the kernel family of the dimer calls: thread per pair (bit-plane filter + queued search), 16 or 64 lanes
        os.environ["MP_DIMER_LANES"] = str(rng.choice(["1", "16", "64"]))
        try:
            p_deg = float(rng.choice([0.0, 0.03, 0.1, 0.2]))
            # the oracle expands every primer: keep the all-pairs work of a case around a second of CPU
            # one case in four holds primers of up to 64 bases (the 64-bit-plane / 128-bit-string kernels)
            seqs = random_primers(seed, int(rng.integers(2, 260 if p_deg < 0.05 else 40)), p_deg, max_len=64 if rng.random() < 0.25 else 32)
            seqs = [s_ if iupac.degeneracy(s_) <= 256 else "".join(iupac.expand(c)[0] for c in s_) for s_ in seqs]
            codes, off = dimer.encode_primers(seqs)
            for mode, thr in ((0, 3.96), (1, 3.0)):
                args = (codes, off, mode, int(rng.integers(0, len(seqs) + 1)) if mode else 0, dimer.cached_loss_table(thr),
                        dimer.dg_params(), dimer.dg_limit())
                assert np.array_equal(hip.dimer_scan(*args), ora.dimer_scan(*args)), ("dimer_scan", mode)
                n["dimer_scan"] += 1
            n["dimer_cases_thread_per_pair"] += os.environ["MP_DIMER_LANES"] == "1"
            pairs = rng.integers(0, len(seqs), size=(int(rng.integers(1, 400)), 2)).astype(np.int32)
            pargs = (codes, off, pairs, dimer.cached_loss_table(3.6), dimer.dg_params(), dimer.dg_limit())
            assert np.array_equal(hip.dimer_pairs(*pargs), ora.dimer_pairs(*pargs)), "dimer_pairs"
            n["dimer_pairs"] += 1
            ns, nwd = int(rng.integers(1, 60)), int(rng.integers(1, 40))
            sa = rng.integers(0, 1 << 63, size=(ns, nwd), dtype=np.uint64) & rng.integers(0, 1 << 63, size=(ns, nwd), dtype=np.uint64)
            sb = rng.integers(0, 1 << 63, size=(ns, nwd), dtype=np.uint64)
            pr = rng.integers(0, ns, size=(int(rng.integers(1, 300)), 2)).astype(np.int32)
            assert np.array_equal(hip.pair_coverage(sa, sb, pr), ora.pair_coverage(sa, sb, pr)), "pair_coverage"
            n["pair_coverage"] += 1
            # in-silico PCR: sequences built around a template that holds forward sites and reverse-complemented reverse sites
            L = int(rng.integers(60, 400))
            base = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=L)]
            rows = []
            for _ in range(int(rng.integers(1, 120))):
                r = base.copy()
                mut = rng.random(L) < float(rng.choice([0.0, 0.01, 0.05]))
                r[mut] = np.frombuffer(b"ACGTNacgt", np.uint8)[rng.integers(0, 9, size=int(mut.sum()))]
                rows.append(r[: int(rng.integers(L // 2, L + 1))].tobytes())
            roff = np.zeros(len(rows) + 1, np.int64)
            np.cumsum([len(r) for r in rows], out=roff[1:])
            data = np.frombuffer(b"".join(rows), np.uint8)
            comp = {65: 84, 67: 71, 71: 67, 84: 65}
            pc, po = [], [0]
            for _ in range(int(rng.integers(1, 12))):
                lf, lr = int(rng.integers(8, 25)), int(rng.integers(8, 25))
                f0 = int(rng.integers(0, max(1, L - lf - lr - 5)))
                r0 = int(rng.integers(f0, max(f0 + 1, L - lr)))
                f = iupac.MASK_LUT[base[f0:f0 + lf]].copy()
                r = iupac.MASK_LUT[np.array([comp[c] for c in base[r0:r0 + lr][::-1]], np.uint8)].copy()
                if len(f) < lf or len(r) < lr:
                    continue
                for arr in (f, r):
                    if rng.random() < 0.6:
                        arr[int(rng.integers(0, len(arr)))] |= np.uint8(1 << rng.integers(0, 4))
                pc += [f, r]
                po += [po[-1] + lf, po[-1] + lf + lr]
            if pc:
                pcodes, poff = np.concatenate(pc).astype(np.uint8), np.asarray(po, np.int32)
                assert np.array_equal(hip.pcr_scan(data, roff, pcodes, poff), ora.pcr_scan(data, roff, pcodes, poff)), "pcr_scan"
                n["pcr_scan"] += 1
        except AssertionError as e:
            print(json.dumps({"FAILED": str(e.args), "case_seed": seed, "run_seed": a.seed}), flush=True)
            sys.exit(1)
    print(json.dumps({"cases": n, "seconds": a.seconds, "seed": a.seed}))


if __name__ == "__main__":
    main()
