"""TEST INFRASTRUCTURE — scalar restatements of the reference's per-primer string filters (GC content, low-complexity repeats,
hairpin, the 3'-end self-dimer test; V20:387-416, 457-521), one primer at a time, pinned to V20's known answers in
tests/test_kat.py.  The product computes the same values for all output primers at once (multiprime_amd/batchfilters.py, the dimer
kernels); these functions are its checker (tests/test_batchfilters.py) and serve oracle/core_ref.py.  Nothing under multiprime_amd/
imports this module.
"""
from __future__ import annotations

import re
from statistics import mean

from oracle.iupac_ref import REPEATS as _REPEATS, exact_mean, expand, occurs_in_some_expansion, revcomp
from oracle.thermo_ref import delta_g, penalty_points


def gc_fraction(primer: str) -> float:
    """GC_fraction (V20:401-407): mean over expansions of the 3-decimal GC fraction, 2 decimals.
    statistics.mean is exact (rational arithmetic), as in the reference."""
    n = len(primer)
    vals = [round((s.count("G") + s.count("C")) / n, 3) for s in expand(primer)]
    return round(exact_mean(vals), 2)


def has_repeat(primer: str) -> bool:
    """di_nucleotide (V20:410-416)."""
    return any(_REPEATS.search(s) for s in expand(primer))


def has_hairpin(primer: str, distance: int) -> bool:
    """hairpin_check (V20:387-398): a 5-mer whose reverse complement occurs at least `distance`
    bases downstream."""
    for n in range(0, len(primer) - 5 - 5 - distance + 1):
        tail = primer[n + 5 + distance:]
        for s in expand(primer[n:n + 5]):
            if occurs_in_some_expansion(revcomp(s), tail):      # == any(stem in t for t in expand(tail))
                return True
    return False


def three_prime_ends(primer: str, num: int = 5, length: int = 14) -> list[str]:
    """current_end (V20:457-464): expansions of the 3' suffixes of length num .. num+length-1
    (a suffix longer than the primer is the whole primer again, as Python slicing gives)."""
    ends = []
    for i in range(num, num + length):
        s = primer[-i:]
        if s:
            ends.extend(expand(s))
    return ends


def self_dimer(primer: str) -> bool:
    """dimer_check (V20:487-503): does any 3' end (longest first) find its reverse complement
    inside an expansion of the primer with Loss >= 3, or dG < -5 with a flush 3' end?"""
    ends = sorted(three_prime_ends(primer), key=len, reverse=True)
    members = expand(primer)
    seen = {}
    for end in ends:
        rc = revcomp(end)
        for p in members:
            idx = p.find(rc)
            if idx < 0:
                continue
            d2 = len(p) - len(end) - idx
            loss = penalty_points(len(end), end.count("G") + end.count("C"), 0, d2)
            if end not in seen:
                seen[end] = delta_g(end)
            if loss >= 3 or (seen[end] < -5 and d2 == 0):
                return True
    return False


def pre_filter(primer: str, gc_range, distance: int):
    """primer_pre_filter (V20:507-521): the TSV's "Information" column — the GC fraction when
    the primer is clean, else the '|'-joined reasons."""
    lo, hi = gc_range
    info = []
    gc = gc_fraction(primer)
    if not float(lo) <= gc <= float(hi):
        info.append("GC_out_of_range (" + str(gc) + ")")
    if has_repeat(primer):
        info.append("di_nucleotide")
    if has_hairpin(primer, distance):
        info.append("hairpin")
    return gc if not info else "|".join(info)
