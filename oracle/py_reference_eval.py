"""TEST INFRASTRUCTURE — a pure-Python restatement of the reference's coverage evaluation the way the reference computes it
(multiPrime-core_V20.py: Y_distance V20:229-233, mis_primer_check V20:1103-1130, degenerate_seq V20:369-380): a dict of
distinct k-mers with their sequence counts, the candidate expanded into its concrete primers, one numpy score-table difference
per (candidate, uncovered k-mer).  bench.py's `cpu_baseline` leg times it on ONE core (the reference's process pool is inert,
BASELINE.md section 2) on a bounded sample, next to the plain-C oracle, and checks its counts against the oracle's; nothing under
multiprime_amd/ imports it.  Counting convention = BASELINE.md: evals of one call = sum(cover.values())."""
from itertools import product

import numpy as np

# V20:105-110
DEGENERATE_BASE = {"-": ["-"], "A": ["A"], "G": ["G"], "C": ["C"], "T": ["T"], "R": ["A", "G"], "Y": ["C", "T"], "M": ["A", "C"],
                   "K": ["G", "T"], "S": ["G", "C"], "W": ["A", "T"], "H": ["A", "T", "C"], "B": ["G", "T", "C"], "V": ["G", "A", "C"],
                   "D": ["G", "A", "T"], "N": ["A", "T", "G", "C"]}
SCORE_TABLE = {"-": 100, "#": 0.00, "A": 1, "G": 1.11, "C": 1.21, "T": 1.40, "R": 2.11, "Y": 2.61, "M": 2.21, "K": 2.51, "S": 2.32,
               "W": 2.40, "H": 3.61, "B": 3.72, "V": 3.32, "D": 3.51, "N": 4.72}
_SCORES = set(SCORE_TABLE.values())


def y_distance(primer, kmer):
    """Positions where the concrete `kmer` is not covered by the degenerate `primer` (V20:229-233): the score difference of a
    covered position is itself a table value, rounded to two places."""
    diff = np.array([SCORE_TABLE[x] for x in primer]) - np.array([SCORE_TABLE[x] for x in kmer])
    return [i for i in range(len(diff)) if round(diff[i], 2) not in _SCORES]


def expansions(primer):
    """Concrete primers of a degenerate one, in the reference's order (V20:369-380)."""
    return ["".join(p) for p in product(*[DEGENERATE_BASE[s] for s in primer])]


def mis_primer_check(universe, primer, cover, variation, strict_f, strict_r):
    """(perfect, F_mis, R_mis) sequence counts of one candidate over the k-mers of one window (V20:1103-1130; the id lists the
    reference also collects are left out).  `cover`: k-mer -> number of sequences carrying it."""
    own = set(expansions(primer))
    perfect = sum(cover[p] for p in own if p in cover)
    f_mis = r_mis = 0
    for kmer in universe - own:
        dist = y_distance(primer, kmer)
        if len(dist) > variation:
            continue
        hit = set(dist)
        if not hit & strict_f:
            f_mis += cover[kmer]
        if not hit & strict_r:
            r_mis += cover[kmer]
    return perfect, f_mis, r_mis
