"""The per-primer string filters and Tm of the core step (oracle/filters_ref.py, thermo.py — V20:282-336, 387-416, 507-521) for ALL output
primers of an alignment at once, on symbol-code matrices with numpy.

Same results as the scalar functions, value for value (tests/test_batchfilters.py runs both on random degenerate primers;
the 15 golden TSVs hold the "Tm" and "Information" columns):
  * Tm: the nearest-neighbour sums are accumulated position by position (one vector add per position: the same left-to-right
    double additions per expansion as the reference's loop), the closing formula is elementwise IEEE arithmetic, and the final
    round(x, 2) is Python's, applied per value;
  * GC fraction: round(g / n, 3) is looked up per GC count, the mean over a primer's expansions is the exact one;
  * repeats / hairpin: an expansion picks one base per position independently, so "some expansion contains the pattern" is a
    statement about intersections of the positions' base sets — no expansion is enumerated.
"""
from __future__ import annotations

import numpy as np

from . import host, iupac, thermo

_COMP_MASK = np.zeros(16, np.uint8)          # complement of a base set: A<->T, C<->G
for _m in range(16):
    _COMP_MASK[_m] = ((_m & 1) << 3) | ((_m & 2) << 1) | ((_m & 4) >> 1) | ((_m & 8) >> 3)
_POP = np.array([bin(m).count("1") for m in range(16)], np.uint8)
_BASE_OF_CODE = np.zeros(16, np.int64)       # concrete code 1,2,4,8 -> 0..3
_BASE_OF_CODE[[1, 2, 4, 8]] = [0, 1, 2, 3]
_DH = np.asarray(thermo._DH, np.float64)
_DS = np.asarray(thermo._DS, np.float64)
_DH_END = np.asarray([thermo._DH_END[b] for b in "ACGT"], np.float64)
_DS_END = np.asarray([thermo._DS_END[b] for b in "ACGT"], np.float64)


def _segments(src, n):
    """start offsets of every primer's run inside the expansion list (src ascending)."""
    return np.searchsorted(src, np.arange(n + 1))


def _exact_means(vals, seg):
    """statistics.mean (exact rational sum, one rounding) of vals[seg[i]:seg[i+1]] for every i, without rationals: the doubles
    are integers times one common power of two, summed per segment in two int64 halves; Python's int / int is correctly
    rounded, as Fraction -> float is."""
    vals = np.asarray(vals, np.float64)
    seg = np.asarray(seg, np.int64)
    n = len(seg) - 1
    if n == 0:
        return []
    m, e = np.frexp(vals)
    nz = m != 0
    emin = int(e[nz].min()) - 53 if nz.any() else 0
    shift = np.where(nz, e - 53 - emin, 0).astype(np.int64)
    counts = np.diff(seg)
    if int(shift.max(initial=0)) > 15 or int(counts.max()) >= 1 << 20 or not np.isfinite(vals).all() or int(counts.min()) < 1:
        v = vals.tolist()
        return [iupac.exact_mean(v[a:b]) for a, b in zip(seg[:-1].tolist(), seg[1:].tolist())]
    mant = np.ldexp(m, 53).astype(np.int64)                     # exact: |mant| < 2^53
    hi = mant >> 26                                             # floor split, also right for negative values
    lo = mant - (hi << 26)
    hs = np.add.reduceat(hi << shift, seg[:-1]).tolist()
    ls = np.add.reduceat(lo << shift, seg[:-1]).tolist()
    out = []
    for h, l, c in zip(hs, ls, counts.tolist()):
        total = (h << 26) + l
        out.append(total / (c << -emin) if emin < 0 else (total << emin) / c)
    return out


_ERR_ARG = -1          # MP_ERR_ARG (include/mprime.h)


def _no_gap_or_unknown(codes, rc, what):
    """The one refusal of the native forms that no other form may paper over: a primer with a gap or a symbol outside IUPAC."""
    if ((codes == 0) | (codes > 15)).any():
        raise host.MprimeError(rc, what + ": a primer holds a gap / unknown symbol")


def _round2(x: np.ndarray) -> np.ndarray:
    """Python's round(v, 2) of every element: the double nearest to the correctly rounded two-decimal value of v (ties to even on
    the EXACT binary value).  rint(v * 100) / 100 is that same double whenever v * 100 is not within rounding error of a tie (the
    quotient of an integer by 100 is correctly rounded, like the decimal string's conversion); the few elements near a tie — and
    anything not finite or too large for the product to be an integer comparison — go through Python's own round."""
    x = np.asarray(x, np.float64)
    with np.errstate(invalid="ignore"):
        y = x * 100.0
        r = np.rint(y) / 100.0
        frac = np.abs(y - np.floor(y))
    near_tie = ~(np.abs(frac - 0.5) > 1e-6) | ~(np.abs(y) < 4e9)      # (below 2^32 the product is off by < 1e-6)
    if near_tie.any():
        idx = np.nonzero(near_tie)[0]
        r[idx] = [round(v, 2) for v in x[idx].tolist()]
    return r


_TM_PARAMS = np.concatenate([np.asarray(thermo._DH, np.float64).reshape(-1), np.asarray(thermo._DS, np.float64).reshape(-1),
                             np.asarray([thermo._DH_END[b] for b in "ACGT"], np.float64), np.asarray([thermo._DS_END[b] for b in "ACGT"], np.float64),
                             np.asarray([thermo._DS_SYMMETRY, thermo._LN_CONC_A, thermo._LN_CONC_B, thermo.SALT_CORRECTION, thermo.KELVIN], np.float64)])


def tm_of_primers(codes: np.ndarray):
    """[round(mean(Calc_Tm_v2 over the expansions), 2)] per primer (V20:849-852, 282-336): one call of the native host stage
    (mp_primer_tm, csrc/primerstats.cpp).  tm_of_primers_numpy is the same computation on numpy arrays; tests compare the two and both
    with the scalar thermo.tm."""
    codes = np.ascontiguousarray(codes, np.uint8)
    n, k = codes.shape
    if n == 0:
        return []
    out = np.empty(n, np.float64)
    rc = host.dll().mp_primer_tm(k, n, host._ptr(codes), host._ptr(_TM_PARAMS), host._ptr(out))
    if rc == 0:
        return out.tolist()
    if rc not in (_ERR_ARG, host.MP_ERR_CAPACITY):
        raise host.MprimeError(rc, "mp_primer_tm")
    # The native form declines a primer of more than 2^22 expansions (a very high -d) or a mean beyond its 128-bit sum: such a primer
    # goes through the numpy rows and Python's rationals, as rounds 2-3 did for every primer (the reference enumerates them all too);
    # the others stay native.  A gap / unknown symbol fails there as well, with the expansion's own message.
    _no_gap_or_unknown(codes, rc, "mp_primer_tm")
    if n == 1:
        return tm_of_primers_numpy(codes)
    return [tm_of_primers(codes[i:i + 1])[0] for i in range(n)]


def tm_of_primers_numpy(codes: np.ndarray):
    """tm_of_primers on numpy arrays (rounds 2-3; kept as the cross-check of the native form)."""
    n, k = codes.shape
    if n == 0:
        return []
    exp, src = host.expand_kmers(codes)
    idx = _BASE_OF_CODE[exp]                                   # [m][k] base indices
    dh = np.zeros(len(idx), np.float64)
    ds = np.zeros(len(idx), np.float64)
    for t in range(1, k):                                      # dh += Htable[cur][prev], left to right (V20:253-256)
        dh += _DH[idx[:, t], idx[:, t - 1]]
        ds += _DS[idx[:, t], idx[:, t - 1]]
    dh += _DH_END[idx[:, 0]] + _DH_END[idx[:, k - 1]]
    ds += _DS_END[idx[:, 0]] + _DS_END[idx[:, k - 1]]
    sym = np.zeros(len(idx), bool)
    if k % 2 == 0:                                             # symmetry (V20:237-246): first half == RC(reversed second half)
        h = k // 2
        sym = (idx[:, :h] == 3 - idx[:, h:]).all(axis=1)
        ds = np.where(sym, ds + thermo._DS_SYMMETRY, ds)
    dh = dh * 1000
    ln_c = np.where(sym, thermo._LN_CONC_A, thermo._LN_CONC_B)
    t_raw = 1 / ((1 / (dh / (ds + ln_c))) + thermo.SALT_CORRECTION) - thermo.KELVIN
    vals = _round2(t_raw)                                      # thermo.tm rounds every expansion's Tm
    return [round(x, 2) for x in _exact_means(vals, _segments(src, n))]


def gc_of_primers(codes: np.ndarray):
    """filters.gc_fraction per primer (V20:401-407)."""
    n, k = codes.shape
    if n == 0:
        return []
    exp, src = host.expand_kmers(codes)
    r3 = np.asarray([round(g / k, 3) for g in range(k + 1)], np.float64)
    vals = r3[((exp == 2) | (exp == 4)).sum(axis=1)]
    return [round(x, 2) for x in _exact_means(vals, _segments(src, n))]


def repeat_of_primers(codes: np.ndarray) -> np.ndarray:
    """filters.has_repeat per primer (di_nucleotide, V20:410-416): some expansion holds XXXX, (XY)x4 with X != Y or (XYZ)x3 with
    X != Y and Y != Z."""
    n, k = codes.shape
    M = codes
    hit = np.zeros(n, bool)
    for o in range(0, k - 3):
        hit |= (M[:, o] & M[:, o + 1] & M[:, o + 2] & M[:, o + 3]) != 0
    for o in range(0, k - 7):
        a = M[:, o] & M[:, o + 2] & M[:, o + 4] & M[:, o + 6]
        b = M[:, o + 1] & M[:, o + 3] & M[:, o + 5] & M[:, o + 7]
        hit |= (a != 0) & (b != 0) & ~((a == b) & (_POP[a] == 1))
    for o in range(0, k - 8):
        a = M[:, o] & M[:, o + 3] & M[:, o + 6]
        b = M[:, o + 1] & M[:, o + 4] & M[:, o + 7]
        c = M[:, o + 2] & M[:, o + 5] & M[:, o + 8]
        for y in (1, 2, 4, 8):
            hit |= ((b & y) != 0) & ((a & ~np.uint8(y) & 15) != 0) & ((c & ~np.uint8(y) & 15) != 0)
    return hit


def hairpin_of_primers(codes: np.ndarray, distance: int) -> np.ndarray:
    """filters.has_hairpin per primer (V20:387-398): a 5-mer (any of its expansions) whose reverse complement occurs in some
    expansion of the primer at least `distance` bases downstream."""
    n, k = codes.shape
    M = codes
    CM = _COMP_MASK[M]
    hit = np.zeros(n, bool)
    for s in range(0, k - 5 - 5 - distance + 1):
        for o in range(s + 5 + distance, k - 4):
            ok = np.ones(n, bool)
            for j in range(5):                                  # RC(stem)[j] = comp(stem[4 - j]) must lie in the set at o + j
                ok &= (CM[:, s + 4 - j] & M[:, o + j]) != 0
            hit |= ok
    return hit


def information_of_primers(codes: np.ndarray, gc_range, distance: int, native: bool = True):
    """filters.pre_filter per primer (primer_pre_filter, V20:507-521): the TSV's "Information" column.  The three inputs — GC fraction,
    repeat, hairpin — come from one call of the native host stage (mp_primer_filters); native=False takes them from the numpy forms
    above (the cross-check in tests/test_batchfilters.py)."""
    lo, hi = float(gc_range[0]), float(gc_range[1])
    codes = np.ascontiguousarray(codes, np.uint8)
    n, k = codes.shape
    if native and n:
        r3 = np.asarray([round(g / k, 3) for g in range(k + 1)], np.float64)
        gc_a, rep_a, hp_a = np.empty(n, np.float64), np.empty(n, np.uint8), np.empty(n, np.uint8)
        rc = host.dll().mp_primer_filters(k, n, host._ptr(codes), host._ptr(r3), int(distance), host._ptr(gc_a), host._ptr(rep_a), host._ptr(hp_a))
        if rc in (_ERR_ARG, host.MP_ERR_CAPACITY):           # as in tm_of_primers: the primers the native form declines take the numpy forms
            _no_gap_or_unknown(codes, rc, "mp_primer_filters")
            if n == 1:
                return information_of_primers(codes, gc_range, distance, native=False)
            return [information_of_primers(codes[i:i + 1], gc_range, distance)[0] for i in range(n)]
        if rc != 0:
            raise host.MprimeError(rc, "mp_primer_filters")
        gcs, rep, hp = gc_a.tolist(), rep_a.astype(bool).tolist(), hp_a.astype(bool).tolist()
    else:
        gcs = gc_of_primers(codes)
        rep = repeat_of_primers(codes).tolist()
        hp = hairpin_of_primers(codes, distance).tolist()
    out = []
    for gc, r, h in zip(gcs, rep, hp):
        info = []
        if not lo <= gc <= hi:
            info.append("GC_out_of_range (" + str(gc) + ")")
        if r:
            info.append("di_nucleotide")
        if h:
            info.append("hairpin")
        out.append(gc if not info else "|".join(info))
    return out


_FIRST_MASK = np.zeros(16, np.uint8)         # one-hot mask of the first member of a symbol (the reference's member order)
for _sym, _mem in iupac.MEMBERS.items():
    if _sym != "-":
        _FIRST_MASK[iupac.MASK[_sym]] = iupac.MASK[_mem[0]]


def hairpin_first_stem_of_primers(codes: np.ndarray, distance: int) -> np.ndarray:
    """Primers_filter.hairpin_check of get_multiPrime (GM:373-384) per primer (all of one length): only the FIRST expansion of every
    5-mer stem is tried (the reference's generator quirk), against any expansion of the tail `distance` bases further on."""
    n, L = codes.shape
    M = codes
    stem = _COMP_MASK[_FIRST_MASK[M]]                      # complement of the first member, position by position
    hit = np.zeros(n, bool)
    for s in range(0, L - 5 - 5 - distance + 1):
        for o in range(s + 5 + distance, L - 4):
            ok = np.ones(n, bool)
            for j in range(5):                              # RC(stem)[j] = comp(stem[4 - j]) must lie in the set at o + j
                ok &= (stem[:, s + 4 - j] & M[:, o + j]) != 0
            hit |= ok
    return hit
