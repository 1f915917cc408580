"""Deterministic synthetic MSA generator (SURVEY.md §8d input 4).

The reference ships no generator; this one follows the recipe the survey states:
a uniform-random ACGT root of length L, per-base substitution p_sub (uniform over
the other three bases), a fraction of "variable" columns with a higher rate,
internal gaps, leading/trailing gap runs ~ Geometric(mean 8) on a fraction of
rows, and rare IUPAC R/Y codes.  Rows are produced in blocks of `block_rows`
rows, each block seeded by (seed, block index), so a rank that owns rows
[r0, r1) of the global MSA can generate exactly its shard without touching
the others (bench.py --gpus N).

Output is a uint8 matrix of ASCII codes, one row per sequence (no newlines);
`to_fasta` renders it as single-line FASTA with ids ``>s{i:07d}``.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def synth_root(L: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng([seed, 0x7FFF_FFFF])
    return rng.integers(0, 4, size=L, dtype=np.uint8)


def synth_variable_columns(L: int, seed: int, frac: float = 0.05) -> np.ndarray:
    rng = np.random.default_rng([seed, 0x7FFF_FFFE])
    return rng.random(L) < frac


def synth_block(
    row0: int,
    n_rows: int,
    L: int,
    seed: int,
    *,
    p_sub: float = 0.015,
    p_var: float = 0.25,
    var_frac: float = 0.05,
    p_gap: float = 0.002,
    edge_frac: float = 0.10,
    edge_mean: float = 8.0,
    p_iupac: float = 1e-5,
    block_rows: int = 4096,
) -> np.ndarray:
    """ASCII rows [row0, row0+n_rows) of the global synthetic MSA, shape (n_rows, L)."""
    root = synth_root(L, seed)
    var_cols = synth_variable_columns(L, seed, var_frac)
    p_col = np.where(var_cols, p_var, p_sub).astype(np.float64)
    out = np.empty((n_rows, L), dtype=np.uint8)
    b0 = row0 // block_rows
    b1 = (row0 + n_rows - 1) // block_rows if n_rows else b0 - 1
    for b in range(b0, b1 + 1):
        blk = _one_block(b, block_rows, L, seed, root, p_col, p_gap, edge_frac, edge_mean, p_iupac)
        g0 = max(row0, b * block_rows)
        g1 = min(row0 + n_rows, (b + 1) * block_rows)
        out[g0 - row0:g1 - row0] = blk[g0 - b * block_rows:g1 - b * block_rows]
    return out


def _one_block(b, rows, L, seed, root, p_col, p_gap, edge_frac, edge_mean, p_iupac):
    rng = np.random.default_rng([seed, b])
    base = np.broadcast_to(root, (rows, L)).copy()
    sub = rng.random((rows, L)) < p_col
    shift = rng.integers(1, 4, size=(rows, L), dtype=np.uint8)
    base = np.where(sub, (base + shift) & 3, base).astype(np.uint8)
    txt = _ACGT[base]
    gap = rng.random((rows, L)) < p_gap
    txt[gap] = ord("-")
    iu = rng.random((rows, L)) < p_iupac
    ry = np.where(rng.random((rows, L)) < 0.5, ord("R"), ord("Y")).astype(np.uint8)
    txt = np.where(iu, ry, txt)
    col = np.arange(L)[None, :]
    lead_on = rng.random(rows) < edge_frac
    lead_len = np.where(lead_on, rng.geometric(1.0 / edge_mean, size=rows), 0)
    trail_on = rng.random(rows) < edge_frac
    trail_len = np.where(trail_on, rng.geometric(1.0 / edge_mean, size=rows), 0)
    txt = np.where(col < lead_len[:, None], ord("-"), txt)
    txt = np.where(col >= (L - trail_len)[:, None], ord("-"), txt)
    return txt.astype(np.uint8)


def to_fasta(rows: np.ndarray, row0: int = 0) -> bytes:
    parts = []
    for i in range(rows.shape[0]):
        parts.append(b">s%07d\n" % (row0 + i))
        parts.append(rows[i].tobytes())
        parts.append(b"\n")
    return b"".join(parts)
