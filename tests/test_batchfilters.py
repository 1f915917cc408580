"""multiprime_amd/batchfilters.py (numpy over all output primers) against the scalar restatements of the reference's functions
(filters.py / thermo.py, pinned to V20's known answers in test_kat.py): Tm, GC fraction, repeats, hairpin, Information —
value for value on random degenerate primers of several lengths."""
import numpy as np
import pytest

from multiprime_amd import batchfilters, iupac, thermo
from oracle import filters_ref as filters


def random_primers(seed, n, k, p_deg):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        s = list(rng.choice(list("ACGT"), size=k))
        if rng.random() < 0.3:                                   # plant low-complexity stretches and palindromes
            a = int(rng.integers(0, k - 8))
            unit = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 4))))
            s[a:a + 9] = list((unit * 9)[:9])[: len(s[a:a + 9])]
        if rng.random() < 0.3 and k >= 16:
            a = int(rng.integers(0, k - 15))
            stem = s[a:a + 5]
            s[a + 10:a + 15] = list(iupac.revcomp("".join(stem)))
        deg = 1
        for j in range(k):
            if rng.random() < p_deg and deg < 12:
                sym = "RYMKSWHBVDN"[int(rng.integers(0, 11))]
                deg *= iupac.SET_SIZE[sym]
                s[j] = sym
        out.append("".join(s))
    if k % 2 == 0:
        half = "".join(rng.choice(list("ACGT"), size=k // 2))
        out.append(half + iupac.revcomp(half))                   # a self-complementary primer: the symmetry branch of Tm
        out.append(half + iupac.revcomp(half[::-1])[::-1])
    return out


@pytest.mark.parametrize("k,seed", [(14, 1), (18, 2), (18, 3), (22, 4), (27, 5), (36, 6), (45, 7), (63, 8)])
def test_batch_equals_scalar(k, seed):
    prim = random_primers(seed, 600 if k <= 27 else 150, k, 0.06)
    codes = iupac.MASK_LUT[np.frombuffer("".join(prim).encode(), np.uint8)].reshape(len(prim), k)
    want_tm = []
    for p in prim:
        tms = [thermo.tm(e) for e in iupac.expand(p)]
        want_tm.append(round(iupac.exact_mean(tms), 2))
    assert batchfilters.tm_of_primers(codes) == want_tm                       # the native host stage (mp_primer_tm)
    assert batchfilters.tm_of_primers_numpy(codes) == want_tm                 # the same on numpy arrays
    assert batchfilters.gc_of_primers(codes) == [filters.gc_fraction(p) for p in prim]
    want_rep = [filters.has_repeat(p) for p in prim]
    assert batchfilters.repeat_of_primers(codes).tolist() == want_rep and any(want_rep) and not all(want_rep)
    for d in (3, 4):
        want_hp = [filters.has_hairpin(p, d) for p in prim]
        assert batchfilters.hairpin_of_primers(codes, d).tolist() == want_hp
    assert any(filters.has_hairpin(p, 4) for p in prim)
    want_info = [str(filters.pre_filter(p, ["0.2", "0.7"], 4)) for p in prim]
    for native in (True, False):                                              # mp_primer_filters / the numpy forms checked above
        got = batchfilters.information_of_primers(codes, ["0.2", "0.7"], 4, native=native)
        assert [str(x) for x in got] == want_info
        got3 = batchfilters.information_of_primers(codes, ["0.35", "0.55"], 3, native=native)
        assert [str(x) for x in got3] == [str(filters.pre_filter(p, ["0.35", "0.55"], 3)) for p in prim]


def test_native_mean_and_rounding_on_tie_heavy_values():
    """mp_primer_tm's two exactness devices on their own: primers whose expansions' Tm values average to something close to a
    two-decimal tie must round like Python (exact mean, then round-half-even on the exact binary value)."""
    rng = np.random.default_rng(11)
    k = 16
    prim = []
    for _ in range(3000):
        s = list(rng.choice(list("ACGT"), size=k))
        for j in rng.choice(k, size=int(rng.integers(1, 4)), replace=False):
            s[j] = "RYMKSW"[int(rng.integers(0, 6))]                          # 2, 4 or 8 expansions: means land on x.xx5 often
        prim.append("".join(s))
    codes = iupac.MASK_LUT[np.frombuffer("".join(prim).encode(), np.uint8)].reshape(len(prim), k)
    want = [round(iupac.exact_mean([thermo.tm(e) for e in iupac.expand(p)]), 2) for p in prim]
    assert batchfilters.tm_of_primers(codes) == want


def test_segmented_exact_means_equal_statistics_mean():
    """_exact_means (two int64 halves per segment, one int / int division) against statistics.mean on rounded Tm-like values,
    GC-like values, signed values and values with wide exponent spread (which take the rational fallback)."""
    import random
    from statistics import mean

    import numpy as np

    from multiprime_amd.batchfilters import _exact_means
    rng = random.Random(1)
    for trial in range(120):
        counts = [rng.choice([1, 2, 3, 4, 7, 16, 64, 1000]) for _ in range(rng.randrange(1, 30))]
        seg = np.concatenate([[0], np.cumsum(counts)])
        kind = trial % 4
        if kind == 0:
            vals = [round(rng.uniform(20, 90), 2) for _ in range(seg[-1])]
        elif kind == 1:
            vals = [round(rng.randrange(0, 19) / 18, 3) for _ in range(seg[-1])]
        elif kind == 2:
            vals = [rng.uniform(-5, 5) for _ in range(seg[-1])]
        else:
            vals = [rng.choice([0.0, 1e-9, 3.3, -7.25, 1e6]) for _ in range(seg[-1])]
        assert _exact_means(vals, seg) == [mean(vals[a:b]) for a, b in zip(seg[:-1], seg[1:])]


def test_first_stem_hairpins_equal_the_pairing_step_scalar_check():
    """batchfilters.hairpin_first_stem_of_primers against Primers_filter.hairpin_check (get_multiPrime's generator quirk: only the
    first expansion of a stem is tried) on random degenerate primers of several lengths and distances."""
    import random

    import numpy as np

    from multiprime_amd import batchfilters, iupac
    from multiprime_amd.pairing import Primers_filter

    class Stub:
        pass
    rng = random.Random(4)
    for L, dist in ((18, 4), (47, 4), (24, 3), (16, 0), (33, 6)):
        prim = []
        for _ in range(300):
            s = [rng.choice("ACGT") for _ in range(L)]
            for p in range(L):
                if rng.random() < 0.12:
                    s[p] = rng.choice("RYMKSWHBVDN")
            prim.append("".join(s))
        stub = Stub()
        stub.distance = dist
        want = [Primers_filter.hairpin_check(stub, p) for p in prim]
        codes = iupac.MASK_LUT[np.frombuffer("".join(prim).encode(), np.uint8)].reshape(len(prim), L)
        assert batchfilters.hairpin_first_stem_of_primers(codes, dist).tolist() == want and any(want)


def test_primers_the_native_form_declines_take_the_numpy_form(monkeypatch):
    """mp_primer_tm / mp_primer_filters decline a primer beyond 2^22 expansions (a very high -d) or a mean outside their 128-bit sum
    (MP_ERR_CAPACITY / MP_ERR_ARG).  The run must not abort there: such a primer goes through the numpy form, the others stay native.
    A stand-in library declines every primer with three or more degenerate positions; the results must not change."""
    prim = random_primers(11, 200, 18, 0.1)
    codes = iupac.MASK_LUT[np.frombuffer("".join(prim).encode(), np.uint8)].reshape(len(prim), 18)
    want_tm = batchfilters.tm_of_primers(codes)
    want_info = batchfilters.information_of_primers(codes, ["0.2", "0.7"], 4)
    real = batchfilters.host.dll()
    declined = {"tm": 0, "filters": 0}

    def n_degenerate(ptr, n, k):
        a = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint8 * (n * k)).from_address(ptr.value)).reshape(n, k)
        return (np.bitwise_count(a) > 1).sum(axis=1)

    class Picky:
        def mp_primer_tm(self, k, n, codes_p, params, out):
            if (n_degenerate(codes_p, n, k) >= 3).any():
                declined["tm"] += 1
                return -4
            return real.mp_primer_tm(k, n, codes_p, params, out)

        def mp_primer_filters(self, k, n, codes_p, r3, distance, gc, rep, hp):
            if (n_degenerate(codes_p, n, k) >= 3).any():
                declined["filters"] += 1
                return -1
            return real.mp_primer_filters(k, n, codes_p, r3, distance, gc, rep, hp)

        def __getattr__(self, name):
            return getattr(real, name)

    monkeypatch.setattr(batchfilters.host, "dll", lambda: Picky())
    assert batchfilters.tm_of_primers(codes) == want_tm
    assert [str(x) for x in batchfilters.information_of_primers(codes, ["0.2", "0.7"], 4)] == [str(x) for x in want_info]
    assert declined["tm"] > 1 and declined["filters"] > 1
    # a symbol no expansion knows still fails, with the expansion's message
    bad = codes[:2].copy()
    bad[1, 3] = 0
    with pytest.raises(batchfilters.host.MprimeError):
        batchfilters.tm_of_primers(bad)


def test_native_exception_verdicts_equal_the_numpy_statement():
    """mp_exception_verdicts (csrc/primerstats.cpp, threads from 16384 rows up) against the vectorised statement it replaced in
    core._resident_bitsets: gap-type rows, rows with more than v positions that can mismatch, strict positions that can."""
    from multiprime_amd import host
    rng = np.random.default_rng(5)
    for k, v, n in ((18, 1, 50000), (18, 0, 700), (27, 3, 20000), (45, 2, 3000)):
        one_hot = np.array([1, 2, 4, 8], np.uint8)
        primers = np.where(rng.random((37, k)) < 0.8, one_hot[rng.integers(0, 4, size=(37, k))], rng.integers(1, 16, size=(37, k))).astype(np.uint8)
        of = rng.integers(0, len(primers), size=n)
        xc = (primers[of] & (~primers[of] + 1)).astype(np.uint8)                # a row that matches its primer: the lowest member everywhere
        which = rng.random((n, k))
        hit = which < 0.03                                                      # ... then a few other symbols, IUPAC codes and gaps
        xc[hit] = rng.integers(1, 16, size=int(hit.sum())).astype(np.uint8)
        xc[which > 0.99] = 0
        sF = sum(1 << int(y) for y in rng.choice(k, 3, replace=False))
        sR = sum(1 << int(y) for y in rng.choice(k, 3, replace=False))
        got = host.exception_verdicts(xc, of, primers, v, sF, sR)
        gap_type = (xc == 0).sum(axis=1) > v
        can_miss = (xc == 0) | ((xc & ~primers[of]) != 0)
        many = can_miss.sum(axis=1) > v
        pos = np.arange(k)
        want = np.empty((n, 2), bool)
        want[:, 0] = gap_type | many | (can_miss & ((sF >> pos) & 1).astype(bool)).any(axis=1)
        want[:, 1] = gap_type | many | (can_miss & ((sR >> pos) & 1).astype(bool)).any(axis=1)
        assert np.array_equal(got, want) and want.any() and not want.all()
    with pytest.raises(host.MprimeError):
        host.exception_verdicts(xc, of + 1000, primers, v, sF, sR)


def test_native_exception_assignments_equal_selection_plus_verdicts():
    """mp_exception_assignments (selection of output windows and of the shard's rows + verdicts + the layout mp_masks_set_bits takes, on
    several threads from 16384 rows up) against the numpy selection around mp_exception_verdicts that core._resident_bitsets used."""
    from multiprime_amd import host
    rng = np.random.default_rng(11)
    for k, v, n, n_win, row0, n_rows in ((18, 1, 60000, 400, 0, 5000), (18, 0, 900, 50, 1000, 700), (18, 1, 0, 10, 0, 10), (31, 2, 20000, 120, 250, 100000)):
        wins = np.sort(rng.choice(n_win, size=max(1, n_win // 3), replace=False)).astype(np.int32)
        n_out = len(wins)
        primers = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, size=(n_out, k))]
        primers[rng.random((n_out, k)) < 0.1] = 15
        x_window = np.sort(rng.integers(0, n_win, size=n)).astype(np.int32)
        x_row = rng.integers(0, row0 + n_rows + 500, size=n).astype(np.int64)
        xc = rng.integers(0, 16, size=(n, k)).astype(np.uint8)
        xc[rng.random((n, k)) < 0.9] = 1
        sF, sR = 0b101, 0b11 << (k - 2)
        slot_of = np.full(n_win, -1, np.int32)
        slot_of[wins] = np.arange(n_out, dtype=np.int32)
        cand, row, which, value = host.exception_assignments(x_window, x_row, xc, slot_of, row0, n_rows, primers, v, sF, sR)
        mask_i = slot_of[x_window].astype(np.int64)
        r_loc = x_row - row0
        sel = (mask_i >= 0) & (r_loc >= 0) & (r_loc < n_rows)
        m = int(sel.sum())
        assert len(cand) == 2 * m
        if m == 0:
            continue
        bad = host.exception_verdicts(xc[sel], mask_i[sel], primers, v, sF, sR)
        assert np.array_equal(cand, np.repeat(mask_i[sel], 2)) and np.array_equal(row, np.repeat(r_loc[sel], 2))
        assert np.array_equal(which, np.tile(np.array([0, 1], np.uint8), m)) and np.array_equal(value, bad.reshape(-1).astype(np.uint8))
        assert 0 < m < n
    with pytest.raises(host.MprimeError):
        host.exception_assignments(x_window + 1000, x_row, xc, slot_of, row0, n_rows, primers, v, sF, sR)
    with pytest.raises(host.MprimeError):
        host.exception_assignments(np.zeros(3, np.int32), np.zeros(3, np.int64), np.ones((3, 18), np.uint8), np.array([5], np.int32), 0, 10,
                                   np.ones((2, 18), np.uint8), 1, 0, 0)
