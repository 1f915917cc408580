"""GPU parity tests proper: the hand-written HIP kernels against the CPU oracle through the same
C ABI, on seeded inputs — bit-exact (all outputs are integers / bit masks) — plus
size-independent properties at the benchmark's full size."""
import numpy as np
import pytest
import torch  # noqa: F401  (before the HIP library is loaded: torch brings a HIP runtime of its own, and the first one loaded serves the process)

from multiprime_amd import iupac
from multiprime_amd.synth import synth_block

pytestmark = pytest.mark.gpu


def fuzz_msa(seed, n, L, ragged=False, p_gap=0.02, p_iupac=0.004, edge=0.3):
    rows = synth_block(0, n, L, seed, p_gap=p_gap, edge_frac=edge, p_iupac=p_iupac, block_rows=4096)
    rng = np.random.default_rng(seed + 1000)
    junk = np.frombuffer(b"RYMKSWHBVDNnacgtx*?.", dtype=np.uint8)
    hit = rng.random(rows.shape) < p_iupac
    rows = np.where(hit, junk[rng.integers(0, len(junk), rows.shape)], rows).astype(np.uint8)
    if n > 6:
        rows[3, :] = ord("-")                       # an all-gap row
        rows[4, : L // 2] = ord("-")                # long leading run
        rows[5, L // 3:] = ord("-")                 # long trailing run
    lens = np.full(n, L, np.int64)
    if ragged:
        lens = rng.integers(L // 2, L + 1, size=n)
        lens[0] = L
        if n > 8:
            lens[7] = 0                             # an empty record
    off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    data = np.concatenate([rows[i, : lens[i]] for i in range(n)]) if n else np.zeros(0, np.uint8)
    return data, off, int(lens.max())


def both(hip_lib, oracle_lib, data, off):
    ctxs = []
    for lib in (hip_lib, oracle_lib):
        c = lib.context(0)
        c.load_msa(data, off)
        ctxs.append(c)
    return ctxs


def random_candidates(rng, W, k, per_window):
    cw = np.repeat(np.arange(W, dtype=np.int32), per_window)
    codes = rng.integers(1, 16, size=(len(cw), k)).astype(np.uint8)
    one_hot = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, size=codes.shape)]
    concrete = rng.random(codes.shape) < 0.8
    return cw, np.where(concrete, one_hot, codes).astype(np.uint8)


CASES = [
    # seed, n, L, ragged, k, v, p0
    (1, 300, 200, False, 18, 1, 5),
    (2, 257, 150, False, 22, 2, 0),
    (3, 64, 400, True, 16, 1, 0),
    (4, 1000, 120, False, 20, 0, 3),
    (5, 33, 300, True, 28, 3, 10),
    (6, 700, 90, False, 8, 1, 0),
    # primers of 32..63 bases: 64-bit window words
    (7, 300, 260, False, 36, 1, 5),
    (8, 130, 500, True, 45, 2, 0),
    (9, 500, 200, False, 63, 3, 2),
    (10, 257, 150, False, 32, 1, 0),
    (11, 4200, 130, False, 40, 2, 1),
]


@pytest.mark.parametrize("seed,n,L,ragged,k,v,p0", CASES)
def test_every_abi_output_matches_oracle(hip_lib, oracle_lib, seed, n, L, ragged, k, v, p0):
    data, off, maxlen = fuzz_msa(seed, n, L, ragged)
    h, o = both(hip_lib, oracle_lib, data, off)
    for a, b in zip(h.row_attributes(), o.row_attributes()):
        assert a.tolist() == b.tolist()
    W = maxlen - k - p0
    if ragged:
        W = int(np.sort(np.diff(off))[n // 4]) - k - p0       # stay where most rows still have residues
    try:
        ne_h = h.build_windows(p0, W, k, v)
    except Exception as e:                                    # a short-window error must be raised by both
        with pytest.raises(type(e)):
            o.build_windows(p0, W, k, v)
        return
    ne_o = o.build_windows(p0, W, k, v)
    assert ne_h == ne_o
    eh, eo = h.get_exceptions(ne_h), o.get_exceptions(ne_o)
    for a, b in zip(eh, eo):
        assert a.tolist() == b.tolist()
    for w in range(0, W, max(1, W // 40)):
        assert h.get_window_words(w).tolist() == o.get_window_words(w).tolist(), f"window words {w}"
    # host-side expansion of the exceptions, exactly as the product does it
    ew, er, ec = eh
    raw = iupac.strings_of(iupac.SYMBOL_LUT[ec]) if ne_h else []
    xw, xk = [], []
    for w_, s in zip(ew.tolist(), raw):
        if s.count("-") <= v and iupac.degeneracy(s) <= 64:
            for e in iupac.expand(s):
                xw.append(w_)
                xk.append(e)
    if xw:
        words = iupac.words_of_kmers(np.frombuffer("".join(xk).encode(), np.uint8).reshape(len(xk), k))
        h.set_extra_rows(np.asarray(xw, np.int32), words)
        o.set_extra_rows(np.asarray(xw, np.int32), words)
    for a, b in zip(h.window_stats(), o.window_stats()):       # state_matrix / trans_matrix sums
        assert np.array_equal(a, b)
    uh, uo = h.window_unique(want_labels=True), o.window_unique(want_labels=True)
    for a, b in zip(uh, uo):
        assert a.tolist() == b.tolist()
    for w in range(0, W, max(1, W // 25)):
        assert h.get_labels(w).tolist() == o.get_labels(w).tolist(), f"labels {w}"
    rng = np.random.default_rng(seed)
    cw, codes = random_candidates(rng, W, k, 3)
    # make a share of the candidates near-matches of real k-mers so that all three counters move
    sF = int(rng.integers(0, 1 << k)) & 0b1110
    sR = (0b111 << (k - 3)) & ((1 << k) - 1)
    rh = h.eval_candidates(cw, codes, sF, sR)
    ro = o.eval_candidates(cw, codes, sF, sR)
    assert rh.tolist() == ro.tolist()
    words0 = o.get_window_words(W // 2)
    ok = (words0[2] >> (8 * words0.dtype.itemsize - 1)) == 0            # MP_WIN_SKIP: the top bit of g
    if ok.any():
        kmers = iupac.kmers_of_words(words0[:, ok][:, :40], k)
        cand = iupac.MASK_LUT[kmers]
        cand[cand == 0] = 15
        cw2 = np.full(len(cand), W // 2, np.int32)
        assert h.eval_candidates(cw2, cand, sF, sR).tolist() == o.eval_candidates(cw2, cand, sF, sR).tolist()


def test_many_distinct_kmers_take_the_global_table_path(hip_lib, oracle_lib):
    # random (non-homologous) rows: every window holds ~n distinct k-mers, far more than the LDS table
    rng = np.random.default_rng(9)
    n, L, k = 6000, 64, 18
    rows = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=(n, L))]
    data, off = rows.reshape(-1), np.arange(n + 1, dtype=np.int64) * L
    h, o = both(hip_lib, oracle_lib, data, off)
    h.build_windows(0, 8, k, 1)
    o.build_windows(0, 8, k, 1)
    uh, uo = h.window_unique(want_labels=True), o.window_unique(want_labels=True)
    assert np.diff(uh[0]).min() > 4096
    for a, b in zip(uh, uo):
        assert a.tolist() == b.tolist()
    assert h.get_labels(3).tolist() == o.get_labels(3).tolist()


def test_full_size_properties(hip_lib):
    """At the benchmark's size (131072 x 1000, k=18) the oracle is too slow; check properties
    that do not depend on size: an all-N candidate covers exactly the universe; the histogram
    counts add up to the same universe; evaluation is additive over row shards; a k-mer that
    occurs c times is perfectly covered c times by itself."""
    n, L, k, v = 131072, 1000, 18, 1
    rows = synth_block(0, n, L, 20250303)
    data, off = rows.reshape(-1), np.arange(n + 1, dtype=np.int64) * L
    ctx = hip_lib.context(0)
    ctx.load_msa(data, off)
    W = 900
    n_ex = ctx.build_windows(20, W, k, v)
    assert 0 < n_ex < n
    uoff, words, count, first = ctx.window_unique()
    gaps = np.array([bin(int(x) & ((1 << k) - 1)).count("1") for x in words[2]])
    win_of = np.repeat(np.arange(W), np.diff(uoff))
    universe = np.bincount(win_of, weights=np.where(gaps <= v, count, 0), minlength=W).astype(np.int64)
    cw = np.arange(W, dtype=np.int32)
    alln = np.full((W, k), 15, np.uint8)
    ev = ctx.eval_candidates(cw, alln, 0, 0)
    # gap symbols mismatch every candidate symbol, so "perfect" under all-N = rows without gaps and
    # perfect + F_mis = the whole universe (strict masks empty)
    assert (ev[:, 0] + ev[:, 1] == universe).all()
    assert (ev[:, 1] == ev[:, 2]).all()
    # the most frequent k-mer of each window covers itself `count` times
    top = np.array([uoff[w] + np.argmax(np.where(gaps[uoff[w]:uoff[w + 1]] == 0, count[uoff[w]:uoff[w + 1]], -1))
                    for w in range(W)])
    cand = iupac.MASK_LUT[iupac.kmers_of_words(words[:, top], k)]
    ev2 = ctx.eval_candidates(cw, cand, 0, 0)
    assert (ev2[:, 0] == count[top]).all()
    # additivity over row shards (what the multi-GPU all-reduce relies on)
    half = n // 2
    parts = []
    for a, b in ((0, half), (half, n)):
        c2 = hip_lib.context(0)
        c2.load_msa(rows[a:b].reshape(-1), np.arange(b - a + 1, dtype=np.int64) * L)
        c2.build_windows(20, W, k, v)
        parts.append(c2.eval_candidates(cw, cand, 0b1100, 0b11 << 14))
        c2.close()
    whole = ctx.eval_candidates(cw, cand, 0b1100, 0b11 << 14)
    assert (parts[0] + parts[1] == whole).all()
    ctx.close()


def chain_candidates(rng, root, W, k, kind):
    """Structured candidate lists per window: refinement chains (each member adds bases to the previous one),
    in either order, several chains back to back, duplicates, unrelated candidates, empty symbols."""
    cw, codes = [], []
    for w in range(W):
        seed = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, size=k)] if rng.random() < 0.3 else root[w: w + k].copy()
        members = []
        n_chains = 1 if kind in ("up", "down") else int(rng.integers(1, 4))
        for _ in range(n_chains):
            cur = seed.copy() if rng.random() < 0.7 else np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, size=k)]
            chain = [cur.copy()]
            for _ in range(int(rng.integers(0, 12))):
                if rng.random() < 0.15:
                    chain.append(cur.copy())                        # a repeated member
                    continue
                for _ in range(int(rng.integers(1, 3))):            # one or two positions widen in one step
                    cur[rng.integers(0, k)] |= np.uint8(1 << rng.integers(0, 4))
                chain.append(cur.copy())
            if kind == "down" or (kind == "mixed" and rng.random() < 0.5):
                chain.reverse()
            members += chain
        if kind == "mixed":
            for _ in range(int(rng.integers(0, 4))):                # unrelated candidates in between
                members.insert(int(rng.integers(0, len(members) + 1)), rng.integers(1, 16, size=k).astype(np.uint8))
            if rng.random() < 0.2:
                members[int(rng.integers(0, len(members)))][rng.integers(0, k)] = 0        # a symbol that matches nothing
        cw += [w] * len(members)
        codes += members
    return np.asarray(cw, np.int32), np.asarray(codes, np.uint8)


@pytest.mark.parametrize("n,v,kind", [(300, 1, "up"), (300, 2, "down"), (700, 0, "mixed"), (9000, 1, "mixed"),
                                      (40000, 2, "mixed"), (40000, 1, "up"), (5000, 3, "mixed")])
def test_candidate_grouping_paths_match_oracle(hip_lib, oracle_lib, monkeypatch, n, v, kind):
    """mp_eval_candidates groups a window's candidates into nested runs (chain kernel) or plain groups of 8
    (symbol-table kernel): every policy and kernel shape must give the oracle's counters."""
    L, k, p0 = 120, 18, 4
    data, off, _ = fuzz_msa(77 + n + v, n, L, ragged=False, p_gap=0.03, p_iupac=0.002)
    W = L - p0 - k - 3
    rng = np.random.default_rng(n * 7 + v)
    root = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, size=L)]
    cw, codes = chain_candidates(rng, root, W, k, kind)
    sF = sum(1 << y for y in (2, 3) if y < k)
    sR = sum(1 << y for y in (2, k - 3, k - 2))
    hip, ora = both(hip_lib, oracle_lib, data, off)
    for c in (hip, ora):
        n_ex = c.build_windows(p0, W, k, v)
        if n_ex:
            ew, er, ec = c.get_exceptions(n_ex)
            raw = iupac.strings_of(iupac.SYMBOL_LUT[ec])
            xw, xk = [], []
            for w_, s in zip(ew.tolist(), raw):
                if s.count("-") <= v:
                    for e in iupac.expand(s):
                        xw.append(w_)
                        xk.append(e)
            if xw:
                chars = np.frombuffer("".join(xk).encode(), np.uint8).reshape(len(xk), k)
                c.set_extra_rows(np.asarray(xw, np.int32), iupac.words_of_kmers(chars))
    want = ora.eval_candidates(cw, codes, sF, sR)
    settings = [{}, {"MP_EVAL_GROUP": "plain"}, {"MP_EVAL_GROUP": "nested"}, {"MP_EVAL_BITS": "1"}, {"MP_EVAL_BITS": "2"},
                {"MP_EVAL_CHAIN": "0"}, {"MP_EVAL_CHAIN": "3"}, {"MP_EVAL_CHAIN": "5"}, {"MP_EVAL_CHAIN": "7"},
                {"MP_EVAL_MODE": "rows"}, {"MP_EVAL_SLIDE": "1"}, {"MP_EVAL_SLIDE": "1", "MP_SLIDE_GW": "1", "MP_SLIDE_BAND": "5"},
                {"MP_EVAL_SLIDE": "1", "MP_SLIDE_GW": "4", "MP_SLIDE_BAND": "64"}, {"MP_EVAL_SLIDE": "1", "MP_SLIDE_STRICT": "0"}, {"MP_EVAL_PROG": "1"},
                {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "3"}, {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "7"},
                {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "10"}, {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "11"}, {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "12"}]
    for env in settings:
        with monkeypatch.context() as m:
            for key, val in env.items():
                m.setenv(key, val)
            got = hip.eval_candidates(cw, codes, sF, sR)
        assert np.array_equal(got, want), f"counters differ with {env or 'defaults'}"


@pytest.mark.parametrize("n,v,ragged,k", [(77, 0, False, 18), (300, 1, False, 18), (2100, 2, False, 18), (9000, 3, False, 18), (40000, 1, False, 18), (513, 4, False, 18),
                                          (6, 1, True, 18), (5, 2, True, 18), (300, 1, False, 33), (2100, 2, False, 47), (130, 3, True, 40), (700, 5, False, 63)])
def test_coverage_masks_match_oracle(hip_lib, oracle_lib, monkeypatch, n, v, ragged, k):
    """mp_eval_masks: the bit-sliced form (the evaluation pass storing its final words; v <= 3) and the row-per-thread form
    (MP_MASK_MODE=rows; any v) against the oracle, bit for bit — alignments with edge gaps, ragged ends, IUPAC codes, rows with
    more than v gaps, candidate groups of every kind, row counts off the word boundaries."""
    L, p0 = 92 + k, 3
    data, off, _ = fuzz_msa(1234 + n + v, n, L, ragged=ragged, p_gap=0.05, p_iupac=0.004)
    W = L - p0 - k - 2
    rng = np.random.default_rng(n * 3 + v)
    root = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, size=L)]
    cw, codes = chain_candidates(rng, root, W, k, "mixed")
    keep = rng.permutation(len(cw))[:120]
    keep.sort()
    cw, codes = cw[keep], codes[keep]
    sF = sum(1 << y for y in (2, 3))
    sR = sum(1 << y for y in (2, k - 3, k - 2))
    hip, ora = both(hip_lib, oracle_lib, data, off)
    if ragged:
        W = int(np.sort(np.diff(off))[n // 4]) - k - p0       # stay where most rows still have residues
        sel = cw < W
        cw, codes = cw[sel], codes[sel]
    try:
        hip.build_windows(p0, W, k, v)
    except Exception as e:                                    # a short-window error must be raised by both
        with pytest.raises(type(e)):
            ora.build_windows(p0, W, k, v)
        return
    ora.build_windows(p0, W, k, v)
    want = ora.eval_masks(cw, codes, sF, sR)
    for env in ({}, {"MP_MASK_MODE": "rows"}):
        with monkeypatch.context() as m:
            for key, val in env.items():
                m.setenv(key, val)
            got = hip.eval_masks(cw, codes, sF, sR)
        for x, y, which in zip(got, want, ("not_f", "not_r")):
            assert np.array_equal(x, y), f"{which} differs with {env or 'defaults'}"


@pytest.mark.parametrize("n,k,v", [(1, 5, 0), (63, 2, 1), (64, 16, 2), (65, 17, 1), (257, 27, 2), (2049, 28, 1),
                                   (33000, 28, 2), (300, 29, 1), (4100, 30, 3), (33000, 31, 2), (520, 31, 0), (8200, 3, 0), (16500, 21, 1), (700, 20, 3), (40000, 18, 3), (300, 12, 4),
                                   (300, 32, 1), (2049, 33, 2), (33000, 48, 2), (65, 62, 3), (520, 63, 0), (4100, 63, 4)])
def test_kernel_shapes_and_extreme_k(hip_lib, oracle_lib, monkeypatch, n, k, v):
    """Row counts around the word / block boundaries and the smallest and largest k, chains of every kind."""
    v = min(v, k - 1)
    L, p0 = 4 * k + 40, 1                              # the half-gap fuzz rows keep more than k residues
    data, off, _ = fuzz_msa(1000 + n + k, n, L, ragged=False, p_gap=0.02, p_iupac=0.003, edge=0.2)
    W = min(L - p0 - k - 1, 60)
    rng = np.random.default_rng(n + 31 * k)
    root = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, size=L)]
    hip, ora = both(hip_lib, oracle_lib, data, off)
    n_ex = hip.build_windows(p0, W, k, v)
    assert ora.build_windows(p0, W, k, v) == n_ex
    for a, b in zip(hip.window_stats(), ora.window_stats()):
        assert np.array_equal(a, b)
    sF = sum(1 << y for y in (0, 2) if y < k)
    sR = sum(1 << y for y in (k - 1, k - 3) if 0 <= y < k)
    for kind in ("up", "down", "mixed"):
        cw, codes = chain_candidates(rng, root, W, k, kind)
        want = ora.eval_candidates(cw, codes, sF, sR)
        for env in ({}, {"MP_EVAL_GROUP": "plain"}, {"MP_EVAL_CHAIN": "0"}, {"MP_EVAL_CHAIN": "6"}, {"MP_EVAL_SLIDE": "1"}, {"MP_EVAL_SLIDE": "1", "MP_SLIDE_GW": "1", "MP_SLIDE_BAND": "3"},
                    {"MP_EVAL_PROG": "1"}, {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "6"}, {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "10"},
                    {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "11"}, {"MP_EVAL_PROG": "1", "MP_EVAL_CHAIN": "12"}):
            with monkeypatch.context() as m:
                for key, val in env.items():
                    m.setenv(key, val)
                got = hip.eval_candidates(cw, codes, sF, sR)
            assert np.array_equal(got, want), f"{kind} counters differ with {env or 'defaults'}"


@pytest.mark.parametrize("strict_f,strict_r", [((), ()), ((0,), (17,)), ((1, 2, 17), (0, 15, 16)), ((2, 3, 5), (2, 3, 5)), ((16, 17), (0, 1, 2)),
                                               ((1, 2, 3, 4), (2,)), ((3,), (0, 4, 9, 16)), ((0, 1, 2), ())])
@pytest.mark.parametrize("n,v", [(2100, 1), (9000, 2), (700, 0), (5000, 3)])
def test_sliding_kernel_strict_position_forms_match_oracle(hip_lib, oracle_lib, monkeypatch, n, v, strict_f, strict_r):
    """[r6] The sliding kernel keeps a launch's strict positions as a two-bit count per side when a side has at most three of them (the reference's
    default `-c 1,2,-1`, V20:85; slidecore.hpp FAST), position by position otherwise (or with MP_SLIDE_STRICT=0): no, one, two, three positions per
    side, the same positions on both sides, four on a side (the per-position form), chains that widen AT the strict positions (every event there is
    a take-back of the count) — every form and band shape against the oracle."""
    L, k, p0 = 110, 18, 2
    data, off, _ = fuzz_msa(4242 + n + v, n, L, ragged=False, p_gap=0.03, p_iupac=0.002)
    W = L - p0 - k - 3
    rng = np.random.default_rng(n * 13 + v + 101 * len(strict_f) + 7 * len(strict_r))
    root = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, size=L)]
    strict = sorted(set(strict_f) | set(strict_r))
    cw, codes = [], []
    for w in range(W):
        for _ in range(int(rng.integers(1, 3))):
            cur = root[w: w + k].copy()
            chain = [cur.copy()]
            for _ in range(int(rng.integers(1, 8))):
                at_strict = strict and rng.random() < 0.6                   # most steps widen a strict position
                cur[int(rng.choice(strict)) if at_strict else int(rng.integers(0, k))] |= np.uint8(1 << rng.integers(0, 4))
                chain.append(cur.copy())
            chain.reverse()                                                 # most degenerate member first, as refinement chains come
            cw += [w] * len(chain)
            codes += chain
    cw, codes = np.asarray(cw, np.int32), np.asarray(codes, np.uint8)
    sF, sR = sum(1 << y for y in strict_f), sum(1 << y for y in strict_r)
    hip, ora = both(hip_lib, oracle_lib, data, off)
    for c in (hip, ora):
        c.build_windows(p0, W, k, v)
    want = ora.eval_candidates(cw, codes, sF, sR)
    for env in ({"MP_EVAL_SLIDE": "1"}, {"MP_EVAL_SLIDE": "1", "MP_SLIDE_STRICT": "0"}, {"MP_EVAL_SLIDE": "1", "MP_SLIDE_GW": "1", "MP_SLIDE_BAND": "7"},
                {"MP_EVAL_SLIDE": "1", "MP_SLIDE_GW": "4", "MP_SLIDE_BAND": "40"}, {}):
        with monkeypatch.context() as m:
            for key, val in env.items():
                m.setenv(key, val)
            got = hip.eval_candidates(cw, codes, sF, sR)
        assert np.array_equal(got, want), f"counters differ with {env or 'defaults'}"


@pytest.mark.parametrize("n,v,kind", [(300, 1, "up"), (9000, 1, "mixed"), (40000, 2, "mixed")])
def test_rotating_launches_match_oracle(hip_lib, oracle_lib, monkeypatch, n, v, kind):
    """mp_eval_launch_rotating: the launch adds to a block the caller vouches is zero and clears the NEXT launch's block inside
    its own grid (no fill dispatch between evaluations).  Three blocks in rotation, every kernel that can carry the side job
    (nested-chain, symbol-table, sliding) and the forms that cannot (row-per-lane, program-driven: a fill launch of their own):
    every launch's counters equal the oracle's, the cleared block is all zeros, and a block the launch must not touch is intact."""
    L, k, p0 = 120, 18, 4
    data, off, _ = fuzz_msa(501 + n + v, n, L, ragged=False, p_gap=0.03, p_iupac=0.002)
    W = L - p0 - k - 3
    rng = np.random.default_rng(n * 11 + v)
    root = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, size=L)]
    cw, codes = chain_candidates(rng, root, W, k, kind)
    sF = sum(1 << y for y in (2, 3) if y < k)
    sR = sum(1 << y for y in (2, k - 3, k - 2))
    hip, ora = both(hip_lib, oracle_lib, data, off)
    for c in (hip, ora):
        c.build_windows(p0, W, k, v)
    want = ora.eval_candidates(cw, codes, sF, sR)
    # the checker's own rotating form: adds to the block, clears the other
    blk, other = want.copy(), np.full_like(want, 5)
    ora.eval_upload(cw, codes, sF, sR)
    ora.eval_launch_rotating(blk.ctypes.data, other.ctypes.data)
    assert np.array_equal(blk, 2 * want) and not other.any()
    dev = torch.device("cuda", 0)
    hip.set_stream(torch.cuda.current_stream().cuda_stream)
    for env in ({}, {"MP_EVAL_GROUP": "plain"}, {"MP_EVAL_BITS": "1"}, {"MP_EVAL_SLIDE": "1"}, {"MP_EVAL_SLIDE": "1", "MP_SLIDE_GW": "1", "MP_SLIDE_BAND": "5"},
                {"MP_EVAL_MODE": "rows"}, {"MP_EVAL_PROG": "1"}):
        with monkeypatch.context() as m:
            for key, val in env.items():
                m.setenv(key, val)
            hip.eval_upload(cw, codes, sF, sR)
            ring = torch.zeros((3, len(cw), 3), dtype=torch.int64, device=dev)
            ring[1:] = 7                                         # only block 0 starts zeroed; 1 is cleared by launch 0, 2 by launch 1
            for step in range(5):
                a, b, c3 = step % 3, (step + 1) % 3, (step + 2) % 3
                before = ring[c3].clone()
                hip.eval_launch_rotating(ring[a].data_ptr(), ring[b].data_ptr())
                torch.cuda.synchronize()
                assert np.array_equal(ring[a].cpu().numpy(), want), f"step {step} with {env or 'defaults'}"
                assert not ring[b].any().item(), f"step {step}: next block not cleared with {env or 'defaults'}"
                assert torch.equal(ring[c3], before)
            # no block to clear: the launch only adds (twice into the same block = twice the counts)
            ring[0].zero_()
            hip.eval_launch_rotating(ring[0].data_ptr())
            hip.eval_launch_rotating(ring[0].data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(ring[0].cpu().numpy(), 2 * want)
    with pytest.raises(Exception):
        hip.eval_launch_rotating(ring[0].data_ptr(), ring[0].data_ptr())
