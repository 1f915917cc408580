"""TEST INFRASTRUCTURE — the checker's own copy of multiprime_amd/msa.py (round 4: the Python oracle no longer imports the
product's host modules, so a slip in one of them cannot hide on both sides of a comparison).  Pinned like the rest of oracle/:
tests/test_oracle_*.py hold it against the fixtures recorded from the unmodified reference (tests/golden/).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

(The file readers of the original are left out: the checker parses records itself, oracle/core_ref.py::parse_records.)

Original header:
FASTA / alignment input for the core step (host side).

`read_records` keeps parse_seq's record semantics (V20:441-455): lines starting with '#' are
skipped, a '>' line sets the current id to its first space-delimited token (leading '>'
included), every other line is stripped and appended to the current id's sequence — so a
repeated id concatenates, as the reference's defaultdict(str) does.  The per-character
mapping of V20:453 (upper-case, keep ACGTRYMKSWHBVD, else '-') is NOT done here: it is
O(bytes) work and runs on the device (mp_load_msa).  The record logic itself is native
(csrc/fasta.cpp behind include/mprime_host.h); its pure-Python restatement is oracle/core_ref.py::parse_records.
"""
from __future__ import annotations

import numpy as np


def region(lead_gap: np.ndarray, rstrip_len: np.ndarray, coverage: float):
    """seq_attribute (V20:617-640): [start, stop) = (higher / lower) quantile at `coverage` of
    the per-row leading-gap length / right-stripped length."""
    start = np.quantile(np.asarray(lead_gap, dtype=np.int64), coverage, method="higher")
    stop = np.quantile(np.asarray(rstrip_len, dtype=np.int64), coverage, method="lower")
    return start, stop


def region_from_histograms(lead_hist: np.ndarray, rstrip_hist: np.ndarray, coverage: float):
    """The same two order statistics from histograms of the two per-row quantities (mp_row_histograms): np.quantile's
    "higher" / "lower" take sorted[ceil((n - 1) q)] / sorted[floor((n - 1) q)], and sorted[i] is the first value whose
    cumulative count exceeds i."""
    n = int(lead_hist.sum())
    q = np.float64(coverage)
    hi = int(np.ceil((n - 1) * q))
    lo = int(np.floor((n - 1) * q))
    start = np.int64(np.searchsorted(np.cumsum(lead_hist), hi, side="right"))
    stop = np.int64(np.searchsorted(np.cumsum(rstrip_hist), lo, side="right"))
    return start, stop


def strict_sets(position: str, k: int):
    """get_Y (V20:1091-1101)."""
    f, r = set(), set()
    for tok in position.split(","):
        y = int(tok.strip())
        if y > 0:
            f.add(y)
            r.add(k - y)
        else:
            f.add(k + y + 1)
            r.add(-y + 1)
    return f, r


def strict_mask(s, k: int) -> int:
    """Bit mask over mismatch indices 0..k-1; members outside that range can never equal an
    index Y_distance returns, so they drop out."""
    m = 0
    for y in s:
        if 0 <= y < k:
            m |= 1 << y
    return m
