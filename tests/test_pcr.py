"""exact in-silico PCR (extract_PCR_product.py) against files recorded from the unmodified
reference script (tests/golden/make_golden_pcr.py): every output file by name and sha256, the coverage
table (pair lines as a sorted list: the reference's order is pool arrival order)."""
import gzip
import hashlib
import json
import os

import pytest

from conftest import GOLDEN
from multiprime_amd.pcr import Product

CASES = {   # golden key -> (reference fasta, primer source, format)
    "shipped_xls": ("Cluster_0_20727.tfa", "final_maxprimers_set.xls", "xls"),
    "seq_format": ("Cluster_0_20727.tfa", "GGTAYGGYYTCAGRCATC,CRACRTATTTCTCDAGGT", "seq"),
    "fa_1000": ("1000.fasta", "pcr_primers.fa", "fa"),
    "variant1_fa": ("pcr_variant1.fa", "pcr_primers.fa", "fa"),
    "variant2_fa": ("pcr_variant2.fa", "pcr_primers.fa", "fa"),
}


# primers of 41-52 nt (adaptor-tailed): recorded by tests/golden/make_golden_pcr_long.py
LONG_CASES = {"long_fa": ("pcr_long_ref.fa", "pcr_long_primers.fa", "fa"), "long_seq": ("pcr_long_ref.fa", None, "seq")}


def unzip(name, tmp_path):
    path = tmp_path / name
    path.write_bytes(gzip.open(os.path.join(GOLDEN, "inputs", name + ".gz")).read())
    return str(path)


def check(case, lib, tmp_path):
    if case in LONG_CASES:
        book = json.loads(gzip.open(os.path.join(GOLDEN, "pcr_long.json.gz")).read())
        gold = book[case]
        ref_name, primers, fmt = LONG_CASES[case]
        primers = primers or book["long_seq_primers"]
    else:
        gold = json.loads(gzip.open(os.path.join(GOLDEN, "pcr.json.gz")).read())[case]
        ref_name, primers, fmt = CASES[case]
    ref = unzip(ref_name, tmp_path)
    primer_arg = primers if fmt == "seq" else unzip(primers, tmp_path)
    od, cov = tmp_path / "out", tmp_path / "cov.xls"
    Product(primer_file=primer_arg, output_file=str(od), ref_file=ref, file_format=fmt, coverage=str(cov), library=lib).run()
    got = {fn: hashlib.sha256(open(od / fn, "rb").read()).hexdigest() for fn in sorted(os.listdir(od))}
    assert got == {fn: v["sha256"] for fn, v in gold["files"].items()}
    lines = cov.read_text().splitlines()
    assert sorted(l for l in lines if l.startswith("Number of")) == gold["coverage_pairs"]
    assert [l for l in lines if not l.startswith("Number of")] == gold["coverage_totals"]


@pytest.mark.parametrize("case", list(CASES) + list(LONG_CASES))
def test_pcr_matches_reference(case, oracle_lib, tmp_path):
    check(case, oracle_lib, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES) + list(LONG_CASES))
def test_pcr_hip_matches_reference(case, hip_lib, tmp_path):
    check(case, hip_lib, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("rolling,long_primers", [(False, False), (True, False), (False, True), (True, True)])
def test_pcr_kernels_equal_oracle_on_random_database(rolling, long_primers, hip_lib, oracle_lib, monkeypatch):
    """Both in-silico PCR kernels — block-per-sequence (LDS-packed segments, occurrence list) and the rolling scan it falls
    back to (also forced for a sequence whose occurrence list overflows) — against the oracle: degenerate primers planted into
    random sequences incl. lower case, N, repeated forward sites, sequences longer than one 4096-position segment and a
    low-complexity sequence with thousands of occurrences.  long_primers: 28-64 nt, the two-word forms of both kernels."""
    import numpy as np
    from multiprime_amd import iupac
    from multiprime_amd.dimer import encode_primers
    if rolling:
        monkeypatch.setenv("MP_PCR_ROLLING", "1")
    rng = np.random.default_rng(3)
    comp = str.maketrans("ACGT", "TGCA")
    primers = []
    for _ in range(12):
        lo, hi = (16, 24) if not long_primers else (28, 64)          # long: two-word patterns, character search in the fall-backs
        f = "".join(rng.choice(list("ACGT"), size=int(rng.integers(lo, hi + 1))))
        r = "".join(rng.choice(list("ACGT"), size=int(rng.integers(lo, hi + 1))))
        f = f[:5] + "R" + f[6:] if rng.random() < 0.5 else f
        r = r[:7] + "Y" + r[8:12] + "N" + r[13:] if rng.random() < 0.5 else r
        primers += [f, r]
    seqs = []
    for i in range(300):
        n = int(rng.integers(50, 9000))
        s = list(rng.choice(list("ACGT"), size=n))
        for _ in range(int(rng.integers(0, 4))):
            p = int(rng.integers(0, 12))
            fe = iupac.expand(primers[2 * p])[int(rng.integers(0, 2)) % len(iupac.expand(primers[2 * p]))]
            re_ = iupac.expand(primers[2 * p + 1])[0].translate(comp)[::-1]
            a = int(rng.integers(0, max(1, n - 700)))
            b = a + int(rng.integers(60, 600))
            if b + len(re_) < n:
                s[a:a + len(fe)] = fe
                s[b:b + len(re_)] = re_
                if rng.random() < 0.3 and b + 100 + len(fe) < n:
                    s[b + 60:b + 60 + len(fe)] = fe          # a second forward site behind the product
        if rng.random() < 0.2:
            j = int(rng.integers(0, n))
            s[j] = "N" if rng.random() < 0.5 else s[j].lower()
        seqs.append("".join(s))
    poly = iupac.expand(primers[0])[0]
    seqs.append((poly + "A") * 400)                        # thousands of forward sites: the occurrence list overflows
    data = np.frombuffer("".join(seqs).encode(), np.uint8)
    off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    codes, poff = encode_primers(primers)
    got = hip_lib.context(0).pcr_scan(data, off, codes, poff)
    want = oracle_lib.context(0).pcr_scan(data, off, codes, poff)
    assert (want[:, :, 0] >= 0).sum() > 50
    assert got.tolist() == want.tolist()
