#!/usr/bin/env python3
"""Randomised soak of the HIP library against the CPU oracle through the C ABI (GPU box): random alignment shapes,
k, v, window ranges and candidate lists (chains in either order, unrelated candidates, empty symbols), every
grouping policy / kernel shape override, for `--seconds`.  Compares exceptions, histograms, window statistics,
coverage counters and coverage masks bit for bit; prints one JSON line.  Exit code 1 on the first difference (the case is printed)."""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from multiprime_amd import iupac  # noqa: E402
from multiprime_amd._abi import Library, MprimeError  # noqa: E402
from test_hip_parity import chain_candidates, fuzz_msa  # noqa: E402

ENVS = [{}, {"MP_EVAL_GROUP": "plain"}, {"MP_EVAL_GROUP": "nested"}, {"MP_EVAL_BITS": "1"}, {"MP_EVAL_BITS": "2"},
        {"MP_EVAL_CHAIN": "0"}, {"MP_EVAL_CHAIN": "3"}, {"MP_EVAL_CHAIN": "5"}, {"MP_EVAL_CHAIN": "7"}, {"MP_EVAL_CHAIN": "8"},
        {"MP_EVAL_MODE": "rows"}, {"MP_EVAL_SLIDE": "1"}, {"MP_EVAL_SLIDE": "1", "MP_SLIDE_STRICT": "0"}, {"MP_EVAL_SLIDE": "1", "MP_SLIDE_GW": "1", "MP_SLIDE_BAND": "6"},
        {"MP_EVAL_SLIDE": "1", "MP_SLIDE_GW": "4", "MP_SLIDE_BAND": "40"}, {"MP_EVAL_PROG": "1"},
        {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "7"}, {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "3"},
        {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "11"}, {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "10"}, {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "12"},
        {"MP_HIST_LDS": "4096"}]
# [r6] primers of 32..63 bases / v = 4, 5: eval_chain_x_kernel in its four shapes, and the row-per-lane kernels it replaces
ENVS_X = [{}, {"MP_EVAL_X_SHAPE": "0"}, {"MP_EVAL_X_SHAPE": "1"}, {"MP_EVAL_X_SHAPE": "2"}, {"MP_EVAL_X_SHAPE": "3"}, {"MP_EVAL_NO_X": "1"},
          {"MP_EVAL_NO_X": "1", "MP_EVAL_GENERIC_V": "1"}]
# [r6] histogram paths: hist2_kernel (default), the round-5 row loop, the one-workgroup-per-window kernel
ENVS_HIST = [{}, {"MP_HIST_V1": "1"}, {"MP_HIST_V1": "1", "MP_HIST_FOLDS": "3"}, {"MP_HIST_REP_ROWS": "1"}]
KEYS = sorted({k for e in ENVS + ENVS_X + ENVS_HIST for k in e} | {"MP_EVAL_GENERIC_V"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    hip = Library()
    ora = Library(os.path.join(REPO, "oracle", "_build", "libmprime_oracle.so"))
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    n_cases = n_short = n_cand_total = 0
    while time.time() < t_end:
        n = int(rng.choice([1, 7, 63, 64, 65, 200, 257, 1000, 2049, 5000, 9000, 17000, 33000, 70000]))
        k = int(rng.integers(2, 32)) if rng.random() < 0.7 else int(rng.integers(32, 64))      # 32..63: 64-bit window words, row-per-lane kernels only
        v = int(rng.integers(0, min(6, k)))                        # [r6] v = 4, 5: six counter levels
        L = int(rng.integers(k + 8, 3 * k + 80))
        p0 = int(rng.integers(0, 6))
        ragged = bool(rng.random() < 0.3) and n > 10
        case = {"n": n, "k": k, "v": v, "L": L, "p0": p0, "ragged": ragged, "seed": int(rng.integers(1 << 30))}
        data, off, maxlen = fuzz_msa(case["seed"], n, L, ragged, p_gap=float(rng.choice([0.0, 0.01, 0.05])),
                                     p_iupac=float(rng.choice([0.0, 0.002, 0.01])), edge=float(rng.choice([0.0, 0.3])))
        W = min(maxlen - k - p0, 40)
        if ragged:
            W = min(int(np.sort(np.diff(off))[n // 4]) - k - p0, 40)
        if W <= 0:
            continue
        ctxs = []
        try:
            for lib in (hip, ora):
                c = lib.context(0)
                c.load_msa(data, off)
                ctxs.append(c)
            res = []
            for c in ctxs:
                try:
                    res.append(c.build_windows(p0, W, k, v))
                except MprimeError as e:
                    res.append(("err", e.args[0] if e.args else None))
            if isinstance(res[0], tuple) or isinstance(res[1], tuple):
                assert isinstance(res[0], tuple) and isinstance(res[1], tuple), ("build_windows", res)
                n_short += 1
                continue
            assert res[0] == res[1], ("n_exceptions", res)
            exs = [c.get_exceptions(res[0]) for c in ctxs]
            for x, y in zip(*exs):
                assert np.array_equal(x, y), "exceptions"
            ew, er, ec = exs[0]
            xw, xk = [], []
            for w_, s in zip(ew.tolist(), iupac.strings_of(iupac.SYMBOL_LUT[ec]) if res[0] else []):
                if s.count("-") <= v and iupac.degeneracy(s) <= 64:
                    for e in iupac.expand(s):
                        xw.append(w_)
                        xk.append(e)
            if xw:
                words = iupac.words_of_kmers(np.frombuffer("".join(xk).encode(), np.uint8).reshape(len(xk), k))
                for c in ctxs:
                    c.set_extra_rows(np.asarray(xw, np.int32), words)
            for x, y in zip(ctxs[0].window_stats(), ctxs[1].window_stats()):
                assert np.array_equal(x, y), "window_stats"
            want_u = ctxs[1].window_unique()
            for env in (ENVS_HIST if k <= 31 else [{}]):
                for key in KEYS:
                    os.environ.pop(key, None)
                os.environ.update(env)
                for x, y in zip(ctxs[0].window_unique(), want_u):
                    assert np.array_equal(x, y), ("window_unique", env)
            for key in KEYS:
                os.environ.pop(key, None)
            root = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, size=maxlen + k)]
            cw, codes = chain_candidates(rng, root, W, k, str(rng.choice(["up", "down", "mixed"])))
            if rng.random() < 0.6:          # few strict positions (the reference's -c lists; the sliding kernel's two strict forms: <= 3 per side, <= 6 in all)
                sF = sum(1 << int(j) for j in set(rng.integers(0, k, size=int(rng.integers(0, 5))).tolist()))
                sR = sum(1 << int(j) for j in set(rng.integers(0, k, size=int(rng.integers(0, 5))).tolist()))
            else:                           # any subset of the positions
                sF = int(rng.integers(0, 1 << k))
                sR = int(rng.integers(0, 1 << k))
            want = ctxs[1].eval_candidates(cw, codes, sF, sR)
            for env in (ENVS if k <= 31 and v <= 3 else ENVS_X):
                for key in KEYS:
                    os.environ.pop(key, None)
                os.environ.update(env)
                got = ctxs[0].eval_candidates(cw, codes, sF, sR)
                assert np.array_equal(got, want), ("eval_candidates", env)
            m = min(len(cw), 40)                                     # per-sequence coverage masks of a few candidates
            for x, y in zip(ctxs[0].eval_masks(cw[:m], codes[:m], sF, sR), ctxs[1].eval_masks(cw[:m], codes[:m], sF, sR)):
                assert np.array_equal(x, y), "eval_masks"
            n_cand_total += len(cw)
            n_cases += 1
        except AssertionError as e:
            print(json.dumps({"FAILED": str(e.args), "case": case}), flush=True)
            sys.exit(1)
        finally:
            for key in KEYS:
                os.environ.pop(key, None)
            for c in ctxs:
                c.close()
    print(json.dumps({"cases": n_cases, "short_window_errors_on_both": n_short, "candidates": n_cand_total,
                      "settings_per_case": len(ENVS), "settings_wide_or_v45": len(ENVS_X), "histogram_settings": len(ENVS_HIST), "seconds": a.seconds, "seed": a.seed}))


if __name__ == "__main__":
    main()
