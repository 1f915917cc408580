"""mp_window_stats (state_matrix / trans_matrix, V20:541-577): the oracle against a direct numpy restatement on
the histogram the reference's own frames are built from (CPU), and the HIP kernels against the oracle (GPU)."""
import numpy as np
import pytest

from multiprime_amd import iupac
from multiprime_amd.synth import synth_block


def _msa(seed, n, L, ragged):
    rows = synth_block(0, n, L, seed, p_gap=0.03, edge_frac=0.3, p_iupac=0.004, block_rows=4096)
    rng = np.random.default_rng(seed + 5)
    if n > 6 and not ragged:                       # (a short ragged row of mostly gaps is MP_ERR_SHORT_WINDOW, as in V20)
        rows[3, :] = ord("-")
        rows[4, : L // 2] = ord("-")
        rows[5, L // 3:] = ord("-")
    lens = np.full(n, L, np.int64)
    if ragged:
        lens = rng.integers(L // 2, L + 1, size=n)
        lens[0] = L
    off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    data = np.concatenate([rows[i, : lens[i]] for i in range(n)])
    return data, off


def _prepare(ctx, data, off, p0, W, k, v):
    ctx.load_msa(data, off)
    n_ex = ctx.build_windows(p0, W, k, v)
    xw, xk = [], []
    if n_ex:
        ew, er, ec = ctx.get_exceptions(n_ex)
        for w_, s in zip(ew.tolist(), iupac.strings_of(iupac.SYMBOL_LUT[ec])):
            if s.count("-") <= v:
                for e in iupac.expand(s):
                    xw.append(w_)
                    xk.append(e)
        if xw:
            chars = np.frombuffer("".join(xk).encode(), np.uint8).reshape(len(xk), k)
            ctx.set_extra_rows(np.asarray(xw, np.int32), iupac.words_of_kmers(chars))
    return xw, xk


def _from_histogram(ctx, W, k, v, xw, xk):
    """The reference's definition on its own data structure: `cover` (k-mers with <= v gaps -> count), which holds
    the device histogram plus one entry per expansion of the IUPAC k-mers (degenerate_seq, V20:368-380)."""
    off, words, count, first = ctx.window_unique(want_labels=False)
    chars = iupac.kmers_of_words(words, k)
    lut = np.full(256, 4, np.int64)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    idx = lut[chars]
    gaps = (chars == ord("-")).sum(axis=1)
    freq = np.zeros((W, 4, k), np.int64)
    nn = np.zeros((W, k - 1, 4, 4), np.int64)
    for w in range(W):
        for e in range(int(off[w]), int(off[w + 1])):
            if gaps[e] > v:
                continue
            for j in range(k):
                a = idx[e, j]
                if a < 4:
                    freq[w, a, j] += count[e]
                    if j + 1 < k and idx[e, j + 1] < 4:
                        nn[w, j, a, idx[e, j + 1]] += count[e]
    for w, s in zip(xw, xk):
        for j in range(k):
            a = "ACGT".find(s[j])
            if a >= 0:
                freq[w, a, j] += 1
                if j + 1 < k and s[j + 1] in "ACGT":
                    nn[w, j, a, "ACGT".index(s[j + 1])] += 1
    return freq, nn


CASES = [(11, 120, 90, False, 18, 1, 3), (12, 64, 120, True, 12, 2, 0), (13, 300, 60, False, 8, 0, 5)]


@pytest.mark.parametrize("seed,n,L,ragged,k,v,p0", CASES)
def test_oracle_stats_are_the_histogram_sums(oracle_lib, seed, n, L, ragged, k, v, p0):
    data, off = _msa(seed, n, L, ragged)
    W = (L // 2 if ragged else L) - p0 - k - 2           # ragged rows are at least L // 2 long
    ctx = oracle_lib.context(0)
    xw, xk = _prepare(ctx, data, off, p0, W, k, v)
    freq, nn = ctx.window_stats()
    want_f, want_n = _from_histogram(ctx, W, k, v, xw, xk)
    assert np.array_equal(freq, want_f)
    assert np.array_equal(nn, want_n)
    assert freq.sum() > 0 and nn.sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,L,ragged,k,v,p0", CASES + [(14, 5000, 150, False, 20, 1, 2), (15, 40000, 80, True, 18, 2, 0),
                                                             (16, 70000, 60, False, 28, 1, 1)])
def test_hip_stats_match_oracle(hip_lib, oracle_lib, seed, n, L, ragged, k, v, p0):
    data, off = _msa(seed, n, L, ragged)
    W = (L // 2 if ragged else L) - p0 - k - 2           # ragged rows are at least L // 2 long
    got = []
    for lib in (hip_lib, oracle_lib):
        ctx = lib.context(0)
        _prepare(ctx, data, off, p0, W, k, v)
        got.append(ctx.window_stats())
    assert np.array_equal(got[0][0], got[1][0])
    assert np.array_equal(got[0][1], got[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,L,ragged,k,v,p0", [(21, 33000, 70, False, 18, 1, 2), (22, 40000, 90, True, 18, 2, 0), (23, 70000, 58, False, 28, 1, 1),
                                                    (24, 36000, 64, False, 3, 0, 0), (25, 50000, 100, False, 40, 2, 3)])
def test_grouped_stats_kernel_matches_oracle(hip_lib, oracle_lib, monkeypatch, seed, n, L, ragged, k, v, p0):
    """[r6] From 32768 rows on the plain rows' counts come from window_stats_group_kernel — G consecutive windows per workgroup sharing their
    column loads (G = 4; MP_STATS_GROUP=8 / 0: eight / the per-window kernel): every form against the oracle, window counts that are no multiple
    of G, k from 3 to 40, ragged rows and patch planes beside them."""
    data, off = _msa(seed, n, L, ragged)
    W = (L // 2 if ragged else L) - p0 - k - 2
    ctx = oracle_lib.context(0)
    _prepare(ctx, data, off, p0, W, k, v)
    want = ctx.window_stats()
    ctx.close()
    for group in (None, "8", "0"):
        with monkeypatch.context() as m:
            if group is not None:
                m.setenv("MP_STATS_GROUP", group)
            h = hip_lib.context(0)
            _prepare(h, data, off, p0, W, k, v)
            got = h.window_stats()
            got2 = h.window_stats_begin()
            h.window_stats_end(*got2)
            h.close()
        for a, b, c2 in zip(got, want, got2):
            assert np.array_equal(a, b), f"MP_STATS_GROUP={group}"
            assert np.array_equal(c2, b), f"MP_STATS_GROUP={group} (two halves)"
