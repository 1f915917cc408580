"""SURVEY §8f-3, the mapper itself: the drop-in's k-mismatch scan against the reference author's OWN bowtie2 + samtools run.

tests/golden/bwt_cluster0.json.gz (made by tests/golden/make_golden_bwt.py from the files the reference ships under
test_data/results/Core_primers_set/BWT_coverage/) holds what rule BWT_validation (multiPrime.py:441-457: `-l 18 -t 1 -s 50,2000`,
primers GGTAYGGYYTCAGRCATC / CRACRTATTTCTCDAGGT) decided for 1158 sequences whose text is committed here:

  * the 500 records of Cluster_0_20727.tfa — 485 with one product row each (start, stop, 175), 15 without;
  * the 673 records of `.out.unmatched.fa` (15 of them the cluster's) — no product.

`bowtie2 -a` reports every alignment, so the run's decisions for these sequences do not depend on the other 19 569 of its database.
The drop-in (scripts/primer_coverage_validation_by_BWT.py -> multiprime_amd/validate.py -> mp_kmm_scan) must decide the same, row
for row.  The fixture is sharp: a budget of 0 or 2 mismatches, or a 3'-term threshold of 0 or 2, each give a different answer
(`test_fixture_discriminates`), so budget = floor((0.6 + 0.6 L) / 6) = 1 at L = 18 and the MD:Z trailing-match rule in REFERENCE
orientation are what bowtie2 + V9 did.  No delta against the real mapper on any of the 1158 sequences."""
import os
import pickle
import subprocess
import sys

import pytest

from conftest import REPO, golden_input, load_gz_json
from multiprime_amd.validate import off_targets


@pytest.fixture(scope="module")
def bwt():
    return load_gz_json("bwt_cluster0.json.gz")


def _inputs(bwt, tmp_path):
    primers = tmp_path / "core_final_maxprimers_set.fa"
    primers.write_text(bwt["primers_fa"])
    cluster = tmp_path / "Cluster_0_20727.tfa"
    cluster.write_bytes(golden_input("Cluster_0_20727.tfa"))
    unmatched = tmp_path / "unmatched.fa"
    unmatched.write_bytes(golden_input("bwt_unmatched.fa"))
    return primers, cluster, unmatched


def _rows(path):
    lines = open(path).read().splitlines()
    assert lines[0].split("\t") == ["Chrom (or Genes)", "Start", "Stop", "Primer_F", "Primer_R", "Product length"]
    out = {}
    for line in lines[1:]:
        c = line.split("\t")
        out.setdefault(c[0], []).append([int(c[1]), int(c[2]), c[3], c[4], int(c[5])])
    return out


def _run(lib, primers, ref, out, flags, **kw):
    off_targets(primer_file=str(primers), term_length=flags["l"], reference_file=str(ref), PCR_product_size=flags["s"],
                mismatch_num=flags["m"], outfile=str(out), term_threshold=kw.pop("term", flags["t"]), library=lib, **kw).run()
    return _rows(out)


def _check_against_bowtie2(lib, bwt, tmp_path):
    primers, cluster, unmatched = _inputs(bwt, tmp_path)
    fl = bwt["flags"]
    # the cluster: 485 sequences with exactly the run's row, the other 15 in <out>.unmatched.fa
    ids = [l[1:].split()[0] for l in open(cluster) if l.startswith(">")]
    records = {}
    for line in open(cluster):
        if line.startswith(">"):
            name = line[1:].split()[0]
            records[name] = line
        else:
            records[name] += line
    targets = tmp_path / "targets.pkl"
    with open(targets, "wb") as f:
        pickle.dump(records, f)
    got = _run(lib, primers, cluster, tmp_path / "cluster.out", fl, targets=str(targets))
    assert got == bwt["rows"]
    assert len(got) == 485 and sum(len(v) for v in got.values()) == 485
    left = [l[1:].split()[0] for l in open(str(tmp_path / "cluster.out") + ".unmatched.fa") if l.startswith(">")]
    assert sorted(left) == bwt["unmatched_in_cluster"] and len(left) == 15 and set(left) | set(got) == set(ids)
    assert (tmp_path / "cluster.out.pair.num").read_text() == \
        "Primer_F\tPrimer_R\tPair_num\ttarget accession number\nCluster_0_20727_F\tCluster_0_20727_R\t485\t485\n"
    assert (tmp_path / "cluster.out.total.acc.num").read_text() == \
        "total coverage of primer set (PS) is: 485\ntotal target number is: 500\n"
    assert (tmp_path / "core_final_maxprimers_set.term.fa").read_text().count(">") == 16 + 12        # Y.YY..R -> 16, R..R....D -> 12
    # all 673 records the run left without a product: none here either
    assert _run(lib, primers, unmatched, tmp_path / "un.out", fl) == {}
    return primers, cluster, unmatched


def test_scan_decides_like_the_reference_bowtie2_run(bwt, oracle_lib, tmp_path, capsys):
    _check_against_bowtie2(oracle_lib, bwt, tmp_path)
    capsys.readouterr()


def test_fixture_discriminates(bwt, oracle_lib, tmp_path, capsys):
    """What the 1158 recorded decisions rule out: other mismatch budgets and other readings of the 3'-term rule."""
    primers, cluster, unmatched = _inputs(bwt, tmp_path)
    fl = bwt["flags"]
    n = {}
    for mm, term in ((0, 1), (2, 1), (1, 0), (1, 2)):
        n[mm, term] = (len(_run(oracle_lib, primers, cluster, tmp_path / "c.out", fl, max_mismatch=mm, term=term)),
                       len(_run(oracle_lib, primers, unmatched, tmp_path / "u.out", fl, max_mismatch=mm, term=term)))
    assert n[0, 1][0] < 485 and n[0, 1][1] == 0            # 45 of the run's products need the one mismatch
    assert n[2, 1][0] > 485 and n[2, 1][1] > 400           # a second mismatch would amplify most of unmatched.fa
    assert n[1, 0][0] == 485 and n[1, 0][1] > 0            # without the trailing-match rule 26 unmatched records get a product
    assert n[1, 2][0] < 485                                # ... and with -t 2 the run would have lost products it reports
    capsys.readouterr()


@pytest.mark.gpu
def test_hip_scan_decides_like_the_reference_bowtie2_run(bwt, hip_lib, tmp_path, capsys):
    assert hip_lib.backend == "hip"
    primers, cluster, _ = _check_against_bowtie2(hip_lib, bwt, tmp_path)
    capsys.readouterr()
    # and through the drop-in command, as the Snakemake rule spells it (multiPrime.py:455-457)
    out = tmp_path / "cli.out"
    cmd = [sys.executable, os.path.join(REPO, "scripts", "primer_coverage_validation_by_BWT.py"), "-i", str(primers), "-r", str(cluster),
           "-l", "18", "-t", "1", "-s", "50,2000", "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    assert _rows(out) == bwt["rows"]
