"""the resident sequence store (mp_seq_load) — the PCR search and the k-mismatch scan on the packed words give, bit for
bit, what the byte-scanning entry points give on the same text, on the GPU and through the checker; mixed case, IUPAC / junk
characters, empty and ragged sequences, sequences longer than a segment, primers of 33..64 bases, overflowing occurrence lists."""
import numpy as np
import pytest

from multiprime_amd import iupac


def _database(seed, n, max_len, lower=0.05, junk=0.01, low_complexity=False):
    rng = np.random.default_rng(seed)
    seqs = []
    for i in range(n):
        L = int(rng.integers(0, max_len + 1)) if i % 7 else (0 if i == 7 else max_len)
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 2 if low_complexity else 4, size=L)].copy()
        m = rng.random(L)
        s[m < lower] += 32                                             # lower case
        j = (m >= lower) & (m < lower + junk)
        s[j] = np.frombuffer(b"NRYn-*x", np.uint8)[rng.integers(0, 7, size=int(j.sum()))]
        seqs.append(s.tobytes())
    off = np.zeros(n + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    return np.frombuffer(b"".join(seqs), np.uint8), off, seqs


def _pairs(seqs, rng, n_pairs, length, dege=1):
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    codes, poff = [], [0]
    long_ones = [s for s in seqs if len(s) > 4 * length + 200]
    for _ in range(n_pairs):
        s = np.frombuffer(long_ones[int(rng.integers(0, len(long_ones)))].upper(), np.uint8)
        f0 = int(rng.integers(0, len(s) - 2 * length - 150))
        r0 = f0 + length + int(rng.integers(20, 120))
        f = np.array([c if c in comp else 65 for c in s[f0:f0 + length]], np.uint8)
        r = np.array([comp.get(c, 65) for c in s[r0:r0 + length][::-1]], np.uint8)
        fm, rm = iupac.MASK_LUT[f].copy(), iupac.MASK_LUT[r].copy()
        for _ in range(dege):
            fm[int(rng.integers(0, length))] |= np.uint8(1 << rng.integers(0, 4))
            rm[int(rng.integers(0, length))] |= np.uint8(1 << rng.integers(0, 4))
        codes += [fm, rm]
        poff += [poff[-1] + length, poff[-1] + 2 * length]
    return np.concatenate(codes).astype(np.uint8), np.asarray(poff, np.int32)


def _check(lib, data, off, seqs, seed, length, low_complexity=False):
    rng = np.random.default_rng(seed)
    ctx = lib.context(0)
    try:
        ctx.seq_load(data, off)
        n, bases, _ = ctx.seq_info()
        assert (n, bases) == (len(off) - 1, int(off[-1]))
        codes, poff = _pairs(seqs, rng, 6, length, dege=2 if not low_complexity else 1)
        assert np.array_equal(ctx.pcr_scan_resident(codes, poff), ctx.pcr_scan(data, off, codes, poff))
        pat = [np.frombuffer(s.upper(), np.uint8) for s in seqs if len(s) >= 200][:3]
        pc = np.concatenate([iupac.MASK_LUT[np.where(np.isin(p[40:40 + length], [65, 67, 71, 84]), p[40:40 + length], 65)] for p in pat]).astype(np.uint8)
        po = np.arange(len(pat) + 1, dtype=np.int32) * length
        for mm, term in ((0, 0), (2, 3), (3, 0)):
            assert np.array_equal(ctx.kmm_scan_resident(pc, po, mm, term), ctx.kmm_scan(data, off, pc, po, mm, term))
        ctx.seq_free()
        assert ctx.seq_info()[0] == 0
        assert ctx.pcr_scan_resident(codes, poff).shape[1] == 0
    finally:
        ctx.close()


def test_checker_store_equals_its_byte_scans(oracle_lib):
    data, off, seqs = _database(1, 40, 700)
    _check(oracle_lib, data, off, seqs, 2, 18)


@pytest.mark.gpu
@pytest.mark.parametrize("length,max_len,low", [(18, 1500, False), (24, 9500, False), (40, 3000, False), (64, 1200, False), (12, 6000, True)])
def test_resident_scans_equal_the_byte_scans_and_the_oracle(hip_lib, oracle_lib, length, max_len, low):
    import torch  # noqa: F401
    data, off, seqs = _database(10 + length, 120, max_len, low_complexity=low)
    _check(hip_lib, data, off, seqs, length, length, low_complexity=low)
    # and against the checker's byte scan
    rng = np.random.default_rng(length)
    codes, poff = _pairs(seqs, rng, 6, length, dege=1 if low else 2)
    h, o = hip_lib.context(0), oracle_lib.context(0)
    try:
        h.seq_load(data, off)
        assert np.array_equal(h.pcr_scan_resident(codes, poff), o.pcr_scan(data, off, codes, poff))
        # a second load replaces the store
        h.seq_load(data[: off[5]], off[:6])
        assert np.array_equal(h.pcr_scan_resident(codes, poff), o.pcr_scan(data[: off[5]], off[:6], codes, poff))
    finally:
        h.close()
        o.close()


@pytest.mark.gpu
def test_prefix_prefilter_changes_nothing(hip_lib, oracle_lib, monkeypatch):
    """The 8-base prefix filter of pcr_block_kernel (every pattern >= 8 bases) against the unfiltered pattern loop and the checker:
    18-mers and 8-mers (the shortest it admits), and 6-mers (filter off by itself)."""
    import torch  # noqa: F401
    data, off, seqs = _database(77, 150, 2500)
    for length in (18, 8, 6):
        codes, poff = _pairs(seqs, np.random.default_rng(length), 8, length, dege=1)
        o = oracle_lib.context(0)
        want = o.pcr_scan(data, off, codes, poff)
        o.close()
        for env in ("", "1"):
            monkeypatch.setenv("MP_PCR_NO_PREFILTER", env) if env else monkeypatch.delenv("MP_PCR_NO_PREFILTER", raising=False)
            h = hip_lib.context(0)
            h.seq_load(data, off)
            assert np.array_equal(h.pcr_scan_resident(codes, poff), want), (length, env)
            assert np.array_equal(h.pcr_scan(data, off, codes, poff), want), (length, env)
            h.close()
