"""SURVEY §8a rows D (finDimer.py) and M (get_Maxprimerset.py): host logic + mp_dimer_scan against
outputs recorded from the unmodified reference scripts (tests/golden/make_golden_dimer.py).
finDimer's row order is the arrival order of a process pool, so hit lines compare as sorted lists;
every field (Delta G, Loss as printed doubles) must be identical."""
import contextlib
import gzip
import io
import json
import os
import types

import numpy as np
import pytest

from conftest import GOLDEN
from multiprime_amd import dimer, iupac, maxset


@pytest.fixture(scope="module")
def gold():
    g = json.loads(gzip.open(os.path.join(GOLDEN, "dimer_maxset.json.gz")).read())
    g.update(json.loads(gzip.open(os.path.join(GOLDEN, "dimer_long.json.gz")).read()))     # primers of 33..64 bases
    return g


FD = [("findimer_cluster0", "cluster0_candidates.fa", 3.96), ("findimer_cluster0_t3", "cluster0_candidates.fa", 3.0),
      ("findimer_syn_a", "findimer_syn_a.fa", 3.96), ("findimer_syn_a_t3", "findimer_syn_a.fa", 3.0),
      ("findimer_syn_b", "findimer_syn_b.fa", 3.96), ("findimer_syn_b_t3", "findimer_syn_b.fa", 3.0)]
# adaptor-tailed primers (12..59 nt, an NN-tailed adaptor among them; tests/golden/make_golden_dimer_long.py)
FD += [("findimer_long_a", "findimer_long_a.fa", 3.96), ("findimer_long_a_t3", "findimer_long_a.fa", 3.0),
       ("findimer_long_b", "findimer_long_b.fa", 3.96), ("findimer_long_b_t3", "findimer_long_b.fa", 3.0)]
MS = ["maxset_shipped", "maxset_fake1", "maxset_fake2", "maxset_fake3", "maxset_long1", "maxset_long2"]
# multi-cluster inputs with planted cross-cluster dimers (tests/golden/make_golden_maxset2.py): pairs are skipped, clusters go
# to .next.xls (-m T), the maximum-set search back-tracks or dies as the reference does (-m F)
MS_MULTI = [f"maxset_multi{i}" for i in range(1, 9)]


def test_dg_limit_is_the_rounding_boundary():
    import math
    lim = dimer.dg_limit()
    assert round(lim, 2) >= -5 and round(math.nextafter(lim, -math.inf), 2) < -5


def check_findimer(lib, gold, name, inp, thr, tmp_path):
    fa = tmp_path / inp
    fa.write_bytes(gzip.open(os.path.join(GOLDEN, "inputs", inp + ".gz")).read())
    out = tmp_path / "dimer.tsv"
    dimer.Dimer(primer_file=str(fa), outfile=str(out), threshold=thr, library=lib).run()
    lines = out.read_text().splitlines()
    num = open(str(out) + ".dimer_num").read().splitlines()
    want = gold[name]
    assert lines[0] == want["header"] and sorted(lines[1:]) == want["hits"]
    assert num[0] == want["dimer_num_header"] and sorted(num[1:]) == want["dimer_num"]


def check_maxset(lib, gold, name, method, tmp_path):
    if name.startswith("maxset_multi"):
        gold = json.loads(gzip.open(os.path.join(GOLDEN, "maxset_multi.json.gz")).read())
    want = gold[f"{name}_{method}"]
    if name == "maxset_shipped":
        text = gzip.open(os.path.join(GOLDEN, "inputs", "candidate_primers_sets.txt.gz")).read().decode()
    else:
        text = "".join("\t".join(r) + "\n" for r in gold[name + "_rows"])
    inp = tmp_path / "cand.txt"
    inp.write_text(text)
    out = tmp_path / "final.xls"
    opts = types.SimpleNamespace(input=str(inp), step=5, method=method, out=str(out), device=0)
    buf, rc = io.StringIO(), 0
    with contextlib.redirect_stdout(buf):
        try:
            maxset.run(opts, library=lib)
        except SystemExit as e:
            rc = e.code
    nxt = str(out).rstrip(".xls") + ".next.xls"          # the reference strips characters, not the suffix
    assert rc == want["returncode"]
    assert buf.getvalue().splitlines() == want["stdout"]
    assert (out.read_text() if out.exists() else None) == want["out"]
    assert (open(nxt).read() if os.path.exists(nxt) else None) == want["next"]
    assert (tmp_path / "sort.cand.txt").read_text() == want["sort"]


@pytest.mark.parametrize("name,inp,thr", FD)
def test_findimer_matches_reference(name, inp, thr, oracle_lib, gold, tmp_path):
    check_findimer(oracle_lib, gold, name, inp, thr, tmp_path)


@pytest.mark.parametrize("name", MS + MS_MULTI)
@pytest.mark.parametrize("method", ["T", "F"])
def test_maxprimerset_matches_reference(name, method, oracle_lib, gold, tmp_path):
    check_maxset(oracle_lib, gold, name, method, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name,inp,thr", FD)
def test_findimer_hip_matches_reference(name, inp, thr, hip_lib, gold, tmp_path):
    check_findimer(hip_lib, gold, name, inp, thr, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", MS + MS_MULTI)
@pytest.mark.parametrize("method", ["T", "F"])
def test_maxprimerset_hip_matches_reference(name, method, hip_lib, gold, tmp_path):
    check_maxset(hip_lib, gold, name, method, tmp_path)


def random_primers(seed, n, p_deg, max_len=32):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        L = int(rng.integers(4, max_len + 1))
        s = [("ACGT"[int(rng.integers(0, 4))]) for _ in range(L)]
        for p in range(L):
            if rng.random() < p_deg:
                s[p] = "RYMKSWHBVDN"[int(rng.integers(0, 11))]
        out.append("".join(s))
    for _ in range(n // 2):                               # plant complementary 3' ends
        i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
        e = iupac.expand(out[i][-int(rng.integers(5, 10)):])[0]
        t = iupac.revcomp(e)
        if len(out[j]) >= len(t):
            pos = len(out[j]) - len(t) - int(rng.integers(0, 3))
            if pos >= 0:
                out[j] = out[j][:pos] + t + out[j][pos + len(t):]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,p_deg,mode", [(1, 400, 0.03, 0), (2, 150, 0.12, 0), (3, 300, 0.0, 1), (4, 64, 0.2, 0),
                                               (5, 500, 0.0, 0)])
def test_dimer_scan_hip_equals_oracle(hip_lib, oracle_lib, seed, n, p_deg, mode):
    seqs = random_primers(seed, n, p_deg)
    codes, off = dimer.encode_primers(seqs)
    args = (codes, off, mode, n // 5, dimer.cached_loss_table(3.0 if mode else 3.96), dimer.dg_params(), dimer.dg_limit())
    h = hip_lib.context(0).dimer_scan(*args)
    o = oracle_lib.context(0).dimer_scan(*args)
    assert len(o) > 0 and h.tolist() == o.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", ["1", "16", "64"])
def test_dimer_kernels_agree_for_every_lane_width(hip_lib, oracle_lib, lanes, monkeypatch):
    """thread-per-pair (dimer_rows_kernel / dimer_pairs_kernel: bit-plane run filter first) and sub-wave-per-pair (dimer_group_kernel<16|64>, tables in LDS) against the oracle on the
    same degenerate primers, both scan modes and an explicit pair list — incl. primers shorter than 5 and 32-mers, which take
    the un-staged table."""
    monkeypatch.setenv("MP_DIMER_LANES", lanes)
    seqs = random_primers(11, 220, 0.08)
    codes, off = dimer.encode_primers(seqs)
    hc, oc = hip_lib.context(0), oracle_lib.context(0)
    for mode, thr in ((0, 3.96), (1, 3.0)):
        args = (codes, off, mode, 40, dimer.cached_loss_table(thr), dimer.dg_params(), dimer.dg_limit())
        assert hc.dimer_scan(*args).tolist() == oc.dimer_scan(*args).tolist()
    rng = np.random.default_rng(5)
    pairs = rng.integers(0, len(seqs), size=(3000, 2)).astype(np.int32)
    args = (codes, off, pairs, dimer.cached_loss_table(3.6), dimer.dg_params(), dimer.dg_limit())
    got, want = hc.dimer_pairs(*args), oc.dimer_pairs(*args)
    assert want.any() and got.tolist() == want.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_dimer_rows_with_more_hits_than_the_lds_buffer(hip_lib, oracle_lib, mode, monkeypatch):
    """Rows holding thousands of hits (copies of primers that pair with each other) flush the workgroup's LDS hit list several
    times; a capacity below the number of hits reports the full count and fills what fits."""
    monkeypatch.setenv("MP_DIMER_LANES", "1")
    a, b = "ACGTTGCAAGGCTTAACCGGAT", "TTGACCATGGATCCGGTTAAGC"
    rc_a = a[::-1].translate(str.maketrans("ACGT", "TGCA"))
    seqs = [a, b, rc_a] * 500 + random_primers(9, 100, 0.1)
    codes, off = dimer.encode_primers(seqs)
    args = (codes, off, mode, 700, dimer.cached_loss_table(3.0 if mode else 3.96), dimer.dg_params(), dimer.dg_limit())
    o = oracle_lib.context(0).dimer_scan(*args, cap=1 << 21)
    h = hip_lib.context(0).dimer_scan(*args)                     # default capacity 65536: overflows, reports the count, is re-run
    assert len(o) > 70000 and h.tolist() == o.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,p_deg", [(21, 300, 0.0), (22, 200, 0.02), (23, 120, 0.05), (24, 40, 0.0)])
def test_primers_of_up_to_64_bases_hip_equals_oracle(hip_lib, oracle_lib, seed, n, p_deg):
    """Primers of 4..64 bases (the 64-bit-plane / 128-bit-string kernels take every list that holds one of more than 32), both
    scan modes (end lengths up to 63 in the Maxprimerset mode) and an explicit pair list, planted 3' complementarity, 64-mers."""
    seqs = random_primers(seed, n, p_deg, max_len=64)
    seqs = [s if iupac.degeneracy(s) <= 256 else "".join(iupac.expand(c)[0] for c in s) for s in seqs]
    rng = np.random.default_rng(seed)
    x64 = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=64))
    # listed first (the "new" primers of the Maxprimerset mode): a 64-mer, a 64-mer whose 3' end pairs with 40 bases of its end, and
    # a 48-mer whose 3' end pairs with 18 of them
    seqs = [x64, "ACGTACGTACGTACGTACGTACGT" + iupac.revcomp(x64[-40:]), "TTGACCATGGATCCGGTTAAGCTTGACCAT" + iupac.revcomp(x64[-18:])] + seqs
    codes, off = dimer.encode_primers(seqs)
    assert int(np.diff(off).max()) == 64
    hc, oc = hip_lib.context(0), oracle_lib.context(0)
    for mode, thr in ((0, 3.96), (1, 3.0)):
        args = (codes, off, mode, len(seqs) // 3, dimer.cached_loss_table(thr), dimer.dg_params(), dimer.dg_limit())
        h, o = hc.dimer_scan(*args), oc.dimer_scan(*args)
        assert len(o) > 0 and h.tolist() == o.tolist(), mode
        assert int(o[:, 2].max()) == (18 if mode == 0 else 40)                               # the longest end found
    pairs = rng.integers(0, len(seqs), size=(4000, 2)).astype(np.int32)
    args = (codes, off, pairs, dimer.cached_loss_table(3.6), dimer.dg_params(), dimer.dg_limit())
    got, want = hc.dimer_pairs(*args), oc.dimer_pairs(*args)
    assert want.any() and got.tolist() == want.tolist()


@pytest.mark.parametrize("threshold", [3.0, 2.0, 4.5, 0.0, -3.0, 10.0, 60.0, 3.0000000001, 17.76])
def test_loss_table_bisection_equals_entry_by_entry_evaluation(threshold):
    """dimer.loss_table finds, per d2, the first l + GC that passes by bisection (the points do not fall with l + GC); the table is the
    one every entry of which comes from the reference's Penalty_points expression (FD:90-92)."""
    from multiprime_amd import dimer
    assert np.array_equal(dimer.loss_table(threshold), dimer.loss_table_by_evaluation(threshold))
