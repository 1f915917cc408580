"""The primer region (seq_attribute, V20:617-640) from device histograms (mp_row_histograms) instead of per-row arrays: the same
order statistics as np.quantile(method="higher" / "lower") on the rows' values."""
import numpy as np
import pytest

from multiprime_amd import msa


def random_rows(seed, n, width, ragged=False):
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n):
        L = width if not ragged else int(rng.integers(1, width + 1))
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=L)].copy()
        a = int(rng.integers(0, L + 1)) if rng.random() < 0.3 else 0
        b = int(rng.integers(0, L + 1)) if rng.random() < 0.3 else 0
        s[:a] = ord("-")
        if b:
            s[L - b:] = ord("-")
        rows.append(s.tobytes())
    off = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    return np.frombuffer(b"".join(rows), np.uint8), off


def test_order_statistics_from_histograms_equal_numpy_quantiles():
    rng = np.random.default_rng(0)
    for trial in range(300):
        n = int(rng.integers(1, 400))
        width = int(rng.integers(1, 60))
        lead = rng.integers(0, width + 1, size=n)
        rstrip = rng.integers(0, width + 1, size=n)
        for q in (0.0, 0.1, 0.5, 0.6, 0.8, 0.9, 0.95, 1.0, float(rng.random())):
            want = msa.region(lead, rstrip, q)
            got = msa.region_from_histograms(np.bincount(lead, minlength=width + 1), np.bincount(rstrip, minlength=width + 1), q)
            assert (int(got[0]), int(got[1])) == (int(want[0]), int(want[1])), (trial, q)


def check(lib, seed, n, width, ragged):
    data, off = random_rows(seed, n, width, ragged)
    ctx = lib.context(0)
    ctx.load_msa(data, off)
    lead, rstrip, _ = ctx.row_attributes()
    lh, rh = ctx.row_histograms(width + 1)
    assert np.array_equal(lh, np.bincount(lead, minlength=width + 1))
    assert np.array_equal(rh, np.bincount(rstrip, minlength=width + 1))
    with pytest.raises(Exception):
        ctx.row_histograms(max(1, int(max(lead.max(), rstrip.max()))))      # too few bins for the largest value
    ctx.close()


@pytest.mark.parametrize("seed,n,width,ragged", [(1, 50, 30, False), (2, 700, 120, True), (3, 1, 5, False)])
def test_row_histograms_oracle(oracle_lib, seed, n, width, ragged):
    check(oracle_lib, seed, n, width, ragged)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,width,ragged", [(1, 50, 30, False), (2, 700, 120, True), (3, 1, 5, False), (4, 70000, 300, False)])
def test_row_histograms_hip(hip_lib, seed, n, width, ragged):
    check(hip_lib, seed, n, width, ragged)
