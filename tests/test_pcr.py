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


def unzip(name, tmp_path):
    path = tmp_path / name
    path.write_bytes(gzip.open(os.path.join(GOLDEN, "inputs", name + ".gz")).read())
    return str(path)


def check(case, lib, tmp_path):
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


@pytest.mark.parametrize("case", list(CASES))
def test_pcr_matches_reference(case, oracle_lib, tmp_path):
    check(case, oracle_lib, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_pcr_hip_matches_reference(case, hip_lib, tmp_path):
    check(case, hip_lib, tmp_path)
