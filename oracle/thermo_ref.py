"""TEST INFRASTRUCTURE — the checker's own copy of multiprime_amd/thermo.py (round 4: the Python oracle no longer imports the
product's host modules, so a slip in one of them cannot hide on both sides of a comparison).  Pinned like the rest of oracle/:
tests/test_oracle_*.py hold it against the fixtures recorded from the unmodified reference (tests/golden/).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Original header:
Nearest-neighbour thermodynamics of the core step (host side, FP64).

Restates Calc_Tm_v2 / Calc_deltaH_deltaS / symmetry (V20:237-336), deltaG (V20:466-485) and
Penalty_points (V20:192-193).  The arithmetic is kept in the reference's operation order
because the results are rounded to two decimals and printed: a different association could
flip a last digit.  north_star allows 1e-4 on Tm/dG; tests demand identical values.
"""
from __future__ import annotations

import math
from math import log10

from .iupac_ref import expand, revcomp

_IDX = {"A": 0, "C": 1, "G": 2, "T": 3}

# dH (kcal/mol) and dS (cal/mol/K) of the ten nearest-neighbour stacks, indexed
# [next base][this base] like the reference does (V20:154-163, :253)
_DH = ((-7.9, -8.5, -8.2, -7.2), (-8.4, -8, -9.8, -8.2), (-7.8, -10.6, -8, -8.5), (-7.2, -7.8, -8.4, -7.9))
_DS = ((-22.2, -22.7, -22.2, -21.3), (-22.4, -19.9, -24.4, -22.2), (-21, -27.2, -19.9, -22.7),
       (-20.4, -21, -22.4, -22.2))
_DH_END = {"A": 2.3, "T": 2.3, "C": 0.1, "G": 0.1}          # V20:169
_DS_END = {"A": 4.1, "T": 4.1, "C": -2.8, "G": -2.8}        # V20:170
_DS_SYMMETRY = -1.4                                          # V20:173

# 37 C hydrogen-bond model used for the 3'-end dimer dG (V20:129-147)
_FREE = ((-0.7, -0.81, -0.65, -0.65), (-0.67, -0.72, -0.8, -0.65), (-0.69, -0.87, -0.72, -0.81),
         (-0.61, -0.69, -0.67, -0.7))
_PEN = ((0.4, 0.575, 0.33, 0.73), (0.23, 0.32, 0.17, 0.33), (0.41, 0.45, 0.32, 0.575), (0.33, 0.41, 0.23, 0.4))
_HB = ((2, 2.5, 2.5, 2), (2.5, 3, 3, 2.5), (2.5, 3, 3, 2.5), (2, 2.5, 2.5, 2))
_DG_END = {"A": 0.98, "T": 0.98, "C": 1.03, "G": 1.03}      # V20:143
_DG_TA = 0.4
_DG_SYMMETRY = 0.4
# per-stack dG contribution, same expression as V20:473-474 so the products round identically
_DG = tuple(tuple(_FREE[i][j] * _HB[i][j] + _PEN[i][j] for j in range(4)) for i in range(4))
_NA_TERM = 0.175 * math.log(50 / 1000, math.e) + 0.20       # V20:481

KELVIN = 273.15


def _salt_correction() -> float:
    """The constant V20:293-326 actually evaluates to.  Mo=50 mM, Mg=1.5 mM, dNTP=0.25 mM give
    R = sqrt(Mg_free)/Mo = 0.707 >= 0.22, so the "Eq 16" branch runs; its continuation lines
    (V20:324-326) are separate expression statements, so only `a + b*ln(Mg_free)` is kept
    (SURVEY §0-7).  Parity requires this value, not the paper's formula."""
    mono = 50 / 1000.0
    free_mg = (1.5 - 0.25) / 1000.0
    a = 3.92 * pow(10, -5)
    b = -9.11 * pow(10, -6)
    if math.sqrt(free_mg) / (50 / 1000) < 6.0:
        a = 3.92 * pow(10, -5) * (0.843 - (0.352 * math.sqrt(mono) * math.log(mono, math.e)))
    return a + (b * math.log(free_mg, math.e))


SALT_CORRECTION = _salt_correction()
_LN_CONC_B = 1.9872 * math.log(100 / (4 * pow(10, 9)), math.e)      # V20:334
_LN_CONC_A = 1.9872 * math.log(100 / (1 * pow(10, 9)), math.e)      # V20:330 (self-complementary)


def is_symmetric(seq: str) -> bool:
    """symmetry (V20:237-246): even length and first half == complement of reversed... i.e.
    the first half equals RC(reverse(second half))."""
    n = len(seq)
    if n % 2:
        return False
    h = n // 2
    return seq[:h] == revcomp(seq[h:][::-1])


def delta_h_s(seq: str):
    """Calc_deltaH_deltaS (V20:249-261) for a concrete sequence: (dH in cal/mol, dS)."""
    dh = 0
    ds = 0
    prev = _IDX[seq[0]]
    for ch in seq[1:]:
        cur = _IDX[ch]
        dh += _DH[cur][prev]
        ds += _DS[cur][prev]
        prev = cur
    dh += _DH_END[seq[0]] + _DH_END[seq[-1]]
    ds += _DS_END[seq[0]] + _DS_END[seq[-1]]
    if is_symmetric(seq):
        ds += _DS_SYMMETRY
    return dh * 1000, ds


def tm(seq: str) -> float:
    """Calc_Tm_v2 (V20:282-336)."""
    dh, ds = delta_h_s(seq)
    ln_c = _LN_CONC_A if is_symmetric(seq) else _LN_CONC_B
    return round(1 / ((1 / (dh / (ds + ln_c))) + SALT_CORRECTION) - KELVIN, 2)


def delta_g(seq: str) -> float:
    """deltaG (V20:466-485): max over the expansions of `seq`, rounded to 2 decimals."""
    ends_ta = seq[-2:] == "TA"
    best = None
    for s in expand(seq):
        g = 0
        prev = _IDX[s[0]]
        for ch in s[1:]:
            cur = _IDX[ch]
            g += _DG[cur][prev]
            prev = cur
        if ends_ta:
            g += _DG_END[s[0]] + _DG_END[s[-1]] + _DG_TA
        else:
            g += _DG_END[s[0]] + _DG_END[s[-1]]
        g -= _NA_TERM * len(s)
        if is_symmetric(s):
            g += _DG_SYMMETRY
        if best is None or g > best:
            best = g
    return round(best, 2)


def penalty_points(length: int, gc: int, d1: int, d2: int) -> float:
    """Penalty_points (V20:192-193)."""
    return log10((2 ** length * 2 ** gc) / ((2 ** d1 - 0.9) * (2 ** d2 - 0.9)))
