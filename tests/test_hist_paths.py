"""The per-window k-mer histograms (mp_window_unique, V20:689-711) through every key form of csrc/unique.hip against the oracle:
k <= 21 (3k-bit keys), k = 22..31 (2k-bit keys + gap words, round 6), the same wide form forced on small k (MP_WIN_NO_PACK), the
representative-row kernel (MP_HIST_REP_ROWS), hist2_kernel (reference count + dense single-difference counters + dense inserts: the
default) and hist_kernel (MP_HIST_V1, with MP_HIST_FOLDS) — on conserved alignments (the reference and the dense counters do the
work), on random ones (the LDS table overflows into the table in HBM, the tables are rebuilt 8x larger) and with many rows
that carry gaps inside the window (flagged keys).  And the entropy gate's bound against the exact host value on IUPAC-heavy windows."""
import math
from collections import defaultdict

import numpy as np
import pytest
import torch  # noqa: F401  (before the HIP library is loaded)

from multiprime_amd import iupac
from multiprime_amd.synth import synth_block

pytestmark = pytest.mark.gpu


def _tables(lib, data, off, p0, W, k, v, labels=False):
    c = lib.context(0)
    c.load_msa(data, off)
    c.build_windows(p0, W, k, v)
    u = c.window_unique(want_labels=labels)
    lab = [c.get_labels(w).tolist() for w in range(0, W, max(1, W // 7))] if labels else None
    c.close()
    return [x.tolist() for x in u], lab


def _conserved(seed, n, L, p_gap):
    rows = synth_block(0, n, L, seed, p_gap=p_gap, edge_frac=0.2, p_iupac=1e-4, block_rows=4096)
    return rows.reshape(-1), np.arange(n + 1, dtype=np.int64) * L


def _random(seed, n, L):
    rng = np.random.default_rng(seed)
    rows = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=(n, L))]
    rows = np.where(rng.random((n, L)) < 0.01, ord("-"), rows).astype(np.uint8)
    return rows.reshape(-1), np.arange(n + 1, dtype=np.int64) * L


@pytest.mark.parametrize("k", [12, 18, 21, 22, 25, 28, 31])
@pytest.mark.parametrize("env", [{}, {"MP_WIN_NO_PACK": "1"}, {"MP_HIST_V1": "1"}, {"MP_HIST_V1": "1", "MP_HIST_FOLDS": "1"}, {"MP_HIST_REP_ROWS": "1"}])
def test_conserved_alignment_every_key_form(hip_lib, oracle_lib, monkeypatch, k, env):
    if env.get("MP_HIST_REP_ROWS") and k not in (18, 25):
        pytest.skip("the representative-row kernel is one algorithm for every k: two sizes")
    data, off = _conserved(100 + k, 6000, 150, 0.01)
    want, want_lab = _tables(oracle_lib, data, off, 3, 150 - k - 3, k, 1, labels=True)
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    got, got_lab = _tables(hip_lib, data, off, 3, 150 - k - 3, k, 1, labels=True)
    assert got == want
    assert got_lab == want_lab


@pytest.mark.parametrize("v1", ["", "1"])
@pytest.mark.parametrize("k,n", [(18, 9000), (24, 9000), (31, 3000), (11, 40000)])
def test_random_rows_overflow_the_lds_table_and_regrow_the_tables(hip_lib, oracle_lib, monkeypatch, k, n, v1):
    # (k = 11: 4^11 k-mers, 40000 rows -> many repeats AND thousands of distinct keys per slice)
    data, off = _random(k, n, 64)
    want, _ = _tables(oracle_lib, data, off, 0, 64 - k, k, 2)
    monkeypatch.setenv("MP_HIST_V1", v1)
    got, _ = _tables(hip_lib, data, off, 0, 64 - k, k, 2)
    assert got == want


@pytest.mark.parametrize("k", [20, 23, 30])
def test_rows_with_gaps_inside_the_window(hip_lib, oracle_lib, k):
    """p_gap 0.08: most rows carry a gap inside the window (flagged keys with equal base bits and different gap words)."""
    data, off = _conserved(7 + k, 5000, 120, 0.08)
    want, want_lab = _tables(oracle_lib, data, off, 0, 120 - k, k, 3, labels=True)
    got, got_lab = _tables(hip_lib, data, off, 0, 120 - k, k, 3, labels=True)
    assert got == want and got_lab == want_lab


@pytest.mark.parametrize("k", [18, 24])
@pytest.mark.parametrize("threshold", [1.5, 2.6, 3.6])
def test_device_gate_rejects_only_what_the_exact_entropy_rejects(hip_lib, oracle_lib, k, threshold):
    """IUPAC-heavy windows (1 % of the cells: a sixth of the rows of a window carry a code, their expansions make the masses exceed
    the rows): every window the device rejects has an exact tBit (V20:602-614, computed here the reference's way on the oracle's
    tables plus the expanded exception rows) above the threshold."""
    n, L, v = 3000, 140, 1
    rng = np.random.default_rng(k)
    rows = synth_block(0, n, L, 31 + k, p_sub=0.03, p_var=0.3, var_frac=0.15, p_gap=0.004, edge_frac=0.05, p_iupac=0.0, block_rows=4096)
    codes = np.frombuffer(b"RYMKSWHBVDN", np.uint8)
    hit = (rng.random(rows.shape) < 0.01) & (rows != ord("-"))
    rows = np.where(hit, codes[rng.integers(0, len(codes), rows.shape)], rows).astype(np.uint8)
    data, off = rows.reshape(-1), np.arange(n + 1, dtype=np.int64) * L
    W = L - k
    h = hip_lib.context(0)
    h.load_msa(data, off)
    h.build_windows(0, W, k, v)
    h.set_entropy_gate(threshold)
    h.window_unique_device()
    n_rej, rejected = h.entropy_gate_result()
    h.close()
    o = oracle_lib.context(0)
    o.load_msa(data, off)
    n_ex = o.build_windows(0, W, k, v)
    ew, er, ec = o.get_exceptions(n_ex)
    uoff, words, count, first = o.window_unique()
    o.close()
    masses = [defaultdict(int) for _ in range(W)]
    n_rows = np.zeros(W, np.int64)
    for w in range(W):
        a, b = int(uoff[w]), int(uoff[w + 1])
        for i in range(a, b):
            masses[w][(int(words[0][i]), int(words[1][i]), int(words[2][i]))] += int(count[i])
        n_rows[w] = int(count[a:b].sum())
    raw = iupac.strings_of(iupac.SYMBOL_LUT[ec]) if n_ex else []
    for w, s in zip(ew.tolist(), raw):
        n_rows[w] += 1
        if s.count("-") > v:
            masses[w][s] += 1                      # gap_sequence keeps the raw string (V20:689-691)
        else:
            for e in iupac.expand(s):
                masses[w][e] += 1
    exact_reject = np.zeros(W, bool)
    for w in range(W):
        N = float(n_rows[w])
        if N == 0:
            continue
        # the table's words and the expansions' strings may name one k-mer twice (a word triple and a string): entropy of the MERGED
        # masses is what the host sums — merging can only lower it, so the un-merged sum is an upper bound of the exact value and the
        # assertion below (device rejects => exact rejects) is checked against the LOWER, merged one
        merged = defaultdict(int)
        for key, c in masses[w].items():
            if isinstance(key, tuple):
                b0, b1, g = key
                s = "".join("-" if (g >> j) & 1 else "ACGT"[((b0 >> j) & 1) | (((b1 >> j) & 1) << 1)] for j in range(k))
                merged[s] += c
            else:
                merged[key] += c
        t = -sum((c / N) * math.log(c / N, 2) for c in merged.values())
        exact_reject[w] = round(t, 2) > threshold
    assert not (rejected & ~exact_reject).any(), np.nonzero(rejected & ~exact_reject)[0]
    if threshold <= 2.6:
        assert n_rej > 0
