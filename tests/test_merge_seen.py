"""core._merge_seen against a row-by-row replay of V20:689-711: the dictionary the reference builds by walking
the sequences in order (plain rows add their k-mer, IUPAC rows add their expansions) must be what the merge of
the device histogram (k-mer -> count, first row) with the exception k-mers gives — keys, order and counts."""
import numpy as np
import pytest

from multiprime_amd.core import _merge_seen


def _replay(rows):
    d = {}
    for _, keys in sorted(rows.items()):
        for key in keys:
            d[key] = d.get(key, 0) + 1
    return d


@pytest.mark.parametrize("seed", range(40))
def test_merge_equals_row_by_row_replay(seed):
    rng = np.random.default_rng(seed)
    n_rows = int(rng.integers(1, 400))
    alphabet = [f"K{i}" if rng.random() < 0.8 else f"K{i}-" for i in range(int(rng.integers(1, 30)))]
    exc_rows = set(rng.choice(n_rows, size=int(rng.integers(0, max(1, n_rows // 3))), replace=False).tolist())
    rows = {}
    for r in range(n_rows):
        if r in exc_rows:                                   # an IUPAC row: 1..4 expansions, known or new keys
            keys = []
            for _ in range(int(rng.integers(1, 5))):
                keys.append(alphabet[int(rng.integers(0, len(alphabet)))] if rng.random() < 0.6 else f"X{int(rng.integers(0, 12))}")
            rows[r] = list(dict.fromkeys(keys))             # the expansions of one k-mer are distinct
        else:
            rows[r] = [alphabet[int(rng.integers(0, len(alphabet)))]]
    want = _replay(rows)
    # what the device reports: the plain rows only, as (k-mer, count, first row) in first-row order
    dev, first = {}, {}
    for r in sorted(rows):
        if r not in exc_rows:
            key = rows[r][0]
            dev[key] = dev.get(key, 0) + 1
            first.setdefault(key, r)
    order = sorted(dev, key=lambda key: first[key])
    dev = {key: dev[key] for key in order}
    first_arr = np.asarray([first[key] for key in order], np.int64)
    gapfree = np.asarray(["-" not in key for key in order], bool)
    items = [(r, j, key) for r in sorted(exc_rows) for j, key in enumerate(rows[r])]
    got, cnt, flags = _merge_seen(dict(dev), first_arr, gapfree, items)
    assert list(got.items()) == list(want.items())
    assert cnt.tolist() == list(want.values())
    assert flags.tolist() == ["-" not in key for key in want]
    got2, _, none = _merge_seen(dict(dev), first_arr, None, items)      # the gap dictionary carries no flags
    assert list(got2.items()) == list(want.items()) and none is None
