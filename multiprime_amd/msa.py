"""FASTA / alignment input for the core step (host side).

`read_records` keeps parse_seq's record semantics (V20:441-455): lines starting with '#' are
skipped, a '>' line sets the current id to its first space-delimited token (leading '>'
included), every other line is stripped and appended to the current id's sequence — so a
repeated id concatenates, as the reference's defaultdict(str) does.  The per-character
mapping of V20:453 (upper-case, keep ACGTRYMKSWHBVD, else '-') is NOT done here: it is
O(bytes) work and runs on the device (mp_load_msa).
"""
from __future__ import annotations

import numpy as np


def read_records(path: str):
    """Returns (ids, data, row_off): ids in first-appearance order, `data` the concatenated raw
    residue bytes of all records, row r = data[row_off[r]:row_off[r+1]]."""
    with open(path, "rb") as f:
        raw = f.read()
    return parse_records(raw)


def parse_records(raw: bytes):
    pieces: dict[bytes, list[bytes]] = {}
    cur = None
    for line in raw.splitlines():
        if line.startswith(b"#"):
            continue
        if line.startswith(b">"):
            cur = line.strip().split(b" ")[0]
        else:
            if cur is None:
                raise ValueError("sequence data before the first '>' header")
            pieces.setdefault(cur, []).append(line.strip())
    ids = [k.decode("latin-1") for k in pieces]
    rows = [b"".join(v) for v in pieces.values()]
    lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
    row_off = np.zeros(len(rows) + 1, np.int64)
    np.cumsum(lens, out=row_off[1:])
    data = np.frombuffer(b"".join(rows), dtype=np.uint8)
    return ids, data, row_off


def region(lead_gap: np.ndarray, rstrip_len: np.ndarray, coverage: float):
    """seq_attribute (V20:617-640): [start, stop) = (higher / lower) quantile at `coverage` of
    the per-row leading-gap length / right-stripped length."""
    start = np.quantile(np.asarray(lead_gap, dtype=np.int64), coverage, method="higher")
    stop = np.quantile(np.asarray(rstrip_len, dtype=np.int64), coverage, method="lower")
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
