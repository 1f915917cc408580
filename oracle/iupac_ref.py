"""TEST INFRASTRUCTURE — the checker's own copy of multiprime_amd/iupac.py (round 4: the Python oracle no longer imports the
product's host modules, so a slip in one of them cannot hide on both sides of a comparison).  Pinned like the rest of oracle/:
tests/test_oracle_*.py hold it against the fixtures recorded from the unmodified reference (tests/golden/).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Original header:
IUPAC symbol tables shared by the host logic and the kernels' encodings.

A symbol is a 4-bit base-set mask (A=1, C=2, G=4, T=8; '-' = 0).  The reference encodes
the same sets as floating-point "scores" whose sums identify unions (V20:109-110,
`score_table` / `trans_score_table`); SURVEY §0-6 / Appendix A-9 show the two are
equivalent, so the host works on masks and only the *order* of an expansion follows the
reference's `degenerate_base` lists (V20:105-107).
"""
from __future__ import annotations

import re
from itertools import product

import numpy as np

BASES = "ACGT"
MASK = {"-": 0, "A": 1, "C": 2, "G": 4, "T": 8, "R": 5, "Y": 10, "M": 3, "K": 12, "S": 6, "W": 9,
        "H": 11, "B": 14, "V": 7, "D": 13, "N": 15}
SYMBOL = {m: s for s, m in MASK.items()}
# order in which the reference enumerates the members of a degenerate symbol (V20:105-107)
MEMBERS = {"-": "-", "A": "A", "G": "G", "C": "C", "T": "T", "R": "AG", "Y": "CT", "M": "AC", "K": "GT",
           "S": "GC", "W": "AT", "H": "ATC", "B": "GTC", "V": "GAC", "D": "GAT", "N": "ATGC"}
SET_SIZE = {s: max(1, bin(m).count("1")) for s, m in MASK.items()}     # floor(score) in V20:211,215
_COMP = str.maketrans("ATGCRYMKSWHBVDN", "TACGYRKMSWDVBHN")             # V20:218

MASK_LUT = np.zeros(256, np.uint8)
for _s, _m in MASK.items():
    MASK_LUT[ord(_s)] = _m
SYMBOL_LUT = np.frombuffer("".join(SYMBOL[m] for m in range(16)).encode(), dtype=np.uint8)


def expand(seq: str) -> list[str]:
    """All concrete members of a degenerate string, in the reference's order
    (itertools.product over per-position member lists, last position fastest; V20:368-380)."""
    return ["".join(t) for t in product(*(MEMBERS[c] for c in seq))]


_CLASS_OF_BASE = {b: "[" + "".join(sym for sym, m in MASK.items() if sym != "-" and m & MASK[b]) + "]" for b in BASES}
_OCCURS = {}


def occurs_in_some_expansion(concrete: str, degenerate: str) -> bool:
    """any(concrete in e for e in expand(degenerate)) without enumerating the expansions: positions expand independently,
    so the concrete string occurs in SOME expansion iff at some offset every base lies in the symbol's set — one regex
    search with a character class per base (compiled once per concrete string)."""
    pat = _OCCURS.get(concrete)
    if pat is None:
        import re
        pat = _OCCURS[concrete] = re.compile("".join(_CLASS_OF_BASE[b] for b in concrete))
    return pat.search(degenerate) is not None


def exact_mean(vals):
    """statistics.mean of floats (the exact rational mean, rounded once — what the reference uses) without Fractions: every
    double is an integer over a power of two, the sum is kept as one integer over the largest such power, and Python's
    int / int is correctly rounded like Fraction -> float."""
    n = len(vals)
    if n == 1:
        return vals[0]
    if n == 2:
        return (vals[0] + vals[1]) / 2          # fl(a + b) / 2 == fl((a + b) / 2): halving is exact
    try:
        if type(vals[0]) is not float:
            raise ValueError
        total, shift = 0, 0                      # sum = total / 2**shift
        for v in vals:
            a, b = v.as_integer_ratio()
            k = b.bit_length() - 1
            if k > shift:
                total <<= k - shift
                shift = k
            total += a << (shift - k)
        return total / (n << shift)
    except (AttributeError, OverflowError, ValueError):      # ints, inf / nan: the library's own path
        from statistics import mean
        return mean(vals)


def degeneracy(seq) -> int:
    """score_trans (V20:210-211): product of set sizes."""
    d = 1
    for c in seq:
        d *= SET_SIZE[c]
    return d


def n_degenerate(seq) -> int:
    """dege_number (V20:214-215): number of positions holding more than one base."""
    return sum(SET_SIZE[c] > 1 for c in seq)


def revcomp(seq: str) -> str:
    return seq.translate(_COMP)[::-1]


def codes_of(seq: str) -> np.ndarray:
    return MASK_LUT[np.frombuffer(seq.encode(), dtype=np.uint8)]


def words_of_kmers(chars: np.ndarray) -> np.ndarray:
    """(n,k) ASCII matrix of concrete k-mers over ACGT- -> (n,3) window words (mprime.h: uint32 for k <= 31, uint64 above)."""
    n, k = chars.shape
    wt = np.uint32 if k <= 31 else np.uint64
    idx = np.zeros((n, k), wt)
    idx[chars == ord("C")] = 1
    idx[chars == ord("G")] = 2
    idx[chars == ord("T")] = 3
    gap = (chars == ord("-")).astype(wt)
    sh = np.arange(k, dtype=wt)[None, :]
    out = np.empty((n, 3), wt)
    out[:, 0] = np.bitwise_or.reduce((idx & wt(1)) << sh, axis=1)
    out[:, 1] = np.bitwise_or.reduce((idx >> wt(1)) << sh, axis=1)
    out[:, 2] = np.bitwise_or.reduce(gap << sh, axis=1)
    return out


def words_of_codes(codes: np.ndarray) -> np.ndarray:
    """(n,k) concrete symbol codes (A=1 C=2 G=4 T=8, '-'=0) -> (n,3) window words (mprime.h: uint32 for k <= 31, uint64 above)."""
    codes = np.asarray(codes, np.uint8)
    n, k = codes.shape
    wt = np.uint32 if k <= 31 else np.uint64
    sh = np.arange(k, dtype=wt)[None, :]
    out = np.empty((n, 3), wt)
    out[:, 0] = np.bitwise_or.reduce((((codes & 10) != 0).astype(wt)) << sh, axis=1) if n else 0      # C or T: low index bit
    out[:, 1] = np.bitwise_or.reduce((((codes & 12) != 0).astype(wt)) << sh, axis=1) if n else 0      # G or T: high index bit
    out[:, 2] = np.bitwise_or.reduce(((codes == 0).astype(wt)) << sh, axis=1) if n else 0
    return out


_WORD_LUT = np.frombuffer(b"ACGT----", dtype=np.uint8)      # index = b0 | b1 << 1 | gap << 2


def kmers_of_words(words: np.ndarray, k: int) -> np.ndarray:
    """(3,n) window words (uint32, or uint64 for k > 31) -> (n,k) ASCII matrix over ACGT-."""
    n = words.shape[1]
    if n == 0:
        return np.zeros((0, k), np.uint8)
    wb = 4 if k <= 31 else 8
    bits = [np.unpackbits(np.ascontiguousarray(words[i]).astype("<u%d" % wb, copy=False).view(np.uint8).reshape(n, wb),
                          axis=1, bitorder="little")[:, :k] for i in range(3)]
    idx = bits[1] << 1
    idx |= bits[0]
    idx |= bits[2] << 2
    return _WORD_LUT[idx]


def strings_of(chars: np.ndarray) -> list[str]:
    n, k = chars.shape
    if n == 0:
        return []
    if not k:
        return [""] * n
    buf = np.ascontiguousarray(chars).tobytes().decode("ascii")      # one decode, then n slices
    return [buf[i:i + k] for i in range(0, n * k, k)]


def _repeat_patterns():
    """The ACGT members of the reference's `di_nucleotides` set (V20:196-207): XXXX, (XY)x4 with X != Y, (XYZ)x3 with X != Y and
    Y != Z (the reference's `i != j != k` is a chained comparison, so X == Z is allowed).  Members containing '#' can never match
    a primer."""
    pats = set()
    for a in "ACGT":
        pats.add(a * 4)
        for b in "ACGT":
            if a != b:
                pats.add((a + b) * 4)
            for c in "ACGT":
                if a != b and b != c:
                    pats.add((a + b + c) * 3)
    return pats


REPEATS = re.compile("|".join(sorted(_repeat_patterns())))
