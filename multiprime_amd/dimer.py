"""All-pairs 3'-end dimer scan — drop-in for scripts/finDimer.py (finDimer_V4.py, "FD") — and the
dimer test of the greedy primer-set cover (get_Maxprimerset_V1.3.py `dimer_examination`).

The search (which 3' suffix of which primer finds its reverse complement where in which other
primer, first passing combination in the reference's loop order) runs on the GPU through
`mp_dimer_scan`; the floating-point columns of the few hits are then computed on the host with
the reference's own expressions (thermo.delta_g, thermo.penalty_points), so Delta G and Loss are
the identical doubles and print identically.  The two decisions that involve floating point
inside the search — Loss >= threshold and round(deltaG, 2) < -5 — are handed to the library as
an exact table / exact constants computed here with the same libm calls as the reference.
"""
from __future__ import annotations

import math
import os
import time
from collections import defaultdict

import numpy as np

from . import iupac, thermo
from ._abi import Library

MAX_LEN = 64               # MP_DIMER_MAX_LEN: primers of the dimer scans (adaptor-tailed primers included)
PATTERN_MAX_LEN = 64       # MP_PATTERN_MAX_LEN: primers of the sequence scans (in-silico PCR, validation)
HEADERS = ["Primer_ID", "Primer seq", "Primer end", "Delta G", "Primer end length", "End (distance 1)", "End (GC)",
           "Dimer-primer_ID", "Dimer-primer seq", "End (distance 2)", "Loss"]


def loss_table(threshold: float) -> np.ndarray:
    """loss_hit[l][GC][d2] = Penalty_points(l, GC, 0, d2) >= threshold  (FD:90-92, 207)."""
    # 2**l * 2**GC is the integer 2**(l+GC): the value depends on (l + GC, d2) only, and it does not fall when l + GC grows — per d2
    # the first sum that passes is found by bisection with the reference's own expression (512 calls instead of 8256: this table is
    # built once per process and was a fifth of a 40 ms core step; tests/test_dimer.py compares it with the full evaluation)
    n_sum = 2 * MAX_LEN + 1
    first = np.empty(64, np.int64)
    for d2 in range(64):
        lo, hi = 0, n_sum                      # first sum in [lo, hi] whose points reach the threshold (hi = n_sum: none)
        while lo < hi:
            mid = (lo + hi) // 2
            if thermo.penalty_points(mid, 0, 0, d2) >= threshold:
                hi = mid
            else:
                lo = mid + 1
        first[d2] = lo
    by_sum = (np.arange(n_sum)[:, None] >= first[None, :]).astype(np.uint8)
    l, gc = np.arange(MAX_LEN + 1)[:, None], np.arange(MAX_LEN + 1)[None, :]
    t = by_sum[l + gc]                                                     # [l][gc][d2]
    t[(gc > l) | (l == 0)] = 0                                             # an end of l bases holds at most l G/C; there is no end of length 0
    return np.ascontiguousarray(t)


def loss_table_by_evaluation(threshold: float) -> np.ndarray:
    """The same table, every entry from the reference's expression (the checker of loss_table's bisection)."""
    by_sum = np.array([[thermo.penalty_points(sm, 0, 0, d2) >= threshold for d2 in range(64)] for sm in range(2 * MAX_LEN + 1)],
                      np.uint8)
    t = np.zeros((MAX_LEN + 1, MAX_LEN + 1, 64), np.uint8)
    for l in range(1, MAX_LEN + 1):
        for gc in range(0, l + 1):
            t[l, gc] = by_sum[l + gc]
    return t


_LOSS_CACHE: dict[float, np.ndarray] = {}


def cached_loss_table(threshold: float) -> np.ndarray:
    if threshold not in _LOSS_CACHE:
        _LOSS_CACHE[threshold] = loss_table(threshold)
    return _LOSS_CACHE[threshold]


def dg_params() -> np.ndarray:
    """The constants deltaG (FD:171-189) adds up, in the layout include/mprime.h documents."""
    p = np.zeros(16 + 32 + MAX_LEN + 1 + 1, np.float64)
    for i in range(4):
        for j in range(4):
            p[i * 4 + j] = thermo._DG[i][j]
    for a, ca in enumerate("ACGT"):
        for b, cb in enumerate("ACGT"):
            p[16 + (a * 4 + b) * 2 + 0] = thermo._DG_END[ca] + thermo._DG_END[cb]
            p[16 + (a * 4 + b) * 2 + 1] = thermo._DG_END[ca] + thermo._DG_END[cb] + thermo._DG_TA
    for n in range(MAX_LEN + 1):
        p[48 + n] = thermo._NA_TERM * n
    p[48 + MAX_LEN + 1] = thermo._DG_SYMMETRY
    return p


def dg_limit() -> float:
    """Smallest double x with round(x, 2) >= -5, i.e. `round(dG, 2) < -5`  <=>  `dG < dg_limit()`."""
    lo, hi = -5.01, -5.0            # round(lo,2) < -5 <= round(hi,2)
    while True:
        mid = (lo + hi) / 2
        if mid == lo or mid == hi:
            break
        if round(mid, 2) < -5:
            lo = mid
        else:
            hi = mid
    while round(math.nextafter(hi, -math.inf), 2) >= -5:
        hi = math.nextafter(hi, -math.inf)
    return hi


def encode_primers(seqs, max_len: int = MAX_LEN):
    codes = np.concatenate([iupac.codes_of(s) for s in seqs]) if seqs else np.zeros(0, np.uint8)
    off = np.zeros(len(seqs) + 1, np.int32)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    for s in seqs:
        if not 1 <= len(s) <= max_len:
            # the reference has no length limit; this build packs a primer into 128 (dimer scans) / 64 bits ("Limits")
            raise ValueError(f"primer length {len(s)} outside 1..{max_len} (limit of this build, see INTEGRATION.md): {s}")
    if (codes == 0).any():
        raise ValueError("primers may only hold IUPAC nucleotide codes")
    return codes, off


class Dimer(object):
    """Drop-in for the reference class of the same name (FD:127-280)."""

    def __init__(self, primer_file="", outfile="", threshold=3.96, nproc=10, *, library: Library | None = None,
                 device: int = 0):
        self.nproc = nproc                      # accepted for compatibility; the scan runs on the GPU
        self.primers_file = primer_file
        self.threshold = threshold
        self.outfile = os.path.abspath(outfile)
        self.primers = self.parse_primers()
        self.primers_list = list(self.primers.keys())
        self.lib = library if library is not None else Library()
        self.ctx = self.lib.context(device)
        self.stats = {}

    def parse_primers(self):
        """FD:138-146: keyed by sequence — duplicate sequences collapse, the last name wins."""
        primer_dict = {}
        name = None
        with open(self.primers_file, "r") as f:
            for line in f:
                if line.startswith(">"):
                    name = line.strip()
                else:
                    primer_dict[line.strip()] = name
        return primer_dict

    def scan(self):
        """Hit rows in (i, j) order, formatted exactly as FD:208-212."""
        seqs = self.primers_list
        codes, off = encode_primers(seqs)
        t0 = time.time()
        hits = self.ctx.dimer_scan(codes, off, 0, 0, cached_loss_table(self.threshold), dg_params(), dg_limit())
        self.stats["scan_s"] = time.time() - t0
        rows = []
        for i, j, l, ei, pi, idx in hits.tolist():
            pi_seq, pj_seq = seqs[i], seqs[j]
            end = iupac.expand(pi_seq[-l:])[ei]
            d2 = len(pj_seq) - l - idx
            gc = end.count("G") + end.count("C")
            rows.append((self.primers[pi_seq], pi_seq, end, thermo.delta_g(end), l, 0, gc, self.primers[pj_seq], pj_seq,
                         d2, thermo.penalty_points(l, gc, 0, d2)))
        return rows

    def run(self):
        rows = self.scan()
        primer_id_sum = defaultdict(int)
        dimer_primer_id_sum = defaultdict(int)
        with open(self.outfile, "w") as fo:                                   # FD:250-266
            fo.write("\t".join(HEADERS) + "\n")
            for res in rows:
                primer_id_sum[res[0]] += 1
                dimer_primer_id_sum[res[7]] += 1
                fo.write("\t".join(map(str, res)) + "\n")
        with open(self.outfile + ".dimer_num", "w") as fo:                    # FD:274-280
            fo.write("SeqName\tPrimer_ID\tDimer-primer_ID\tRowSum\n")
            for k in primer_id_sum.keys():
                p_id = primer_id_sum[k]
                d_id = dimer_primer_id_sum[k]
                fo.write("\t".join(map(str, [k, p_id, d_id, p_id + d_id])) + "\n")


class DimerExaminer:
    """dimer_examination (get_Maxprimerset_V1.3.py:193-215) against a growing set of concrete
    primers: True iff some 3' suffix (5 .. len-1 nt) of a member of new ∪ selected finds its reverse
    complement in a member with Loss >= 3 or (deltaG < -5 and flush 3' end).  Pairs inside the
    already-selected set were examined when they were added, so only pairs touching a new primer are
    searched (mode 1 of mp_dimer_scan)."""

    def __init__(self, ctx, threshold: float = 3.0):
        self.ctx = ctx
        self.loss = cached_loss_table(threshold)
        self.dg = dg_params()
        self.limit = dg_limit()

    def any_dimer(self, new: list[str], selected: list[str]) -> bool:
        sel = set(selected)
        new = [s for s in dict.fromkeys(new) if s not in sel]
        if not new:
            return False
        codes, off = encode_primers(new + list(selected))
        return self.ctx.dimer_any(codes, off, 1, len(new), self.loss, self.dg, self.limit)
