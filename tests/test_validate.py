"""the k-mismatch primer-site scan (mp_kmm_scan) and the drop-in of primer_coverage_validation_by_BWT.py.

PARITY UNPINNED against the reference: bowtie2 / samtools are not installed, no reference output could be recorded.
What is tested: the oracle's scan against a brute-force Python statement of the acceptance rule in mprime.h, the HIP kernel
against the oracle (`-m gpu`; segment borders, both strands, N / lower case, patterns of several lengths), and the script's
own logic around the mapper (get_term, PCR_product, writers) on a constructed case with known amplicons."""
import os

import numpy as np
import pytest

from multiprime_amd import iupac
from multiprime_amd.validate import bowtie2_mismatch_budget, degenerate_seq, off_targets

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def brute(seqs, pats, max_mm, term):
    hits = []
    for r, s in enumerate(seqs):
        s = s.upper()
        for i, p in enumerate(pats):
            m = len(p)
            for strand in (0, 1):
                q = p if strand == 0 else "".join(COMP[c] for c in reversed(p))
                for pos in range(0, len(s) - m + 1):
                    mis = [j for j in range(m) if s[pos + j] != q[j]]
                    run = m - 1 - mis[-1] if mis else m
                    if len(mis) <= max_mm and run >= term:
                        hits.append((r, pos, i, strand))
    return sorted(hits)


def make_case(rng, n_rows, max_len, n_pat):
    seqs = []
    for _ in range(n_rows):
        n = int(rng.integers(0, max_len))
        s = rng.choice(list("ACGT"), size=n)
        if n:
            s[rng.random(n) < 0.01] = "N"
            low = rng.random(n) < 0.05
            s = np.where(low, np.char.lower(s), s)
        seqs.append("".join(s))
    pats = []
    for _ in range(n_pat):
        m = int(rng.integers(6, 25))
        src = seqs[int(rng.integers(0, n_rows))].upper().replace("N", "A")
        if len(src) > m + 2 and rng.random() < 0.8:            # planted: a slice of a sequence, maybe mutated, maybe reverse-complemented
            a = int(rng.integers(0, len(src) - m))
            p = list(src[a:a + m])
            for _ in range(int(rng.integers(0, 3))):
                p[int(rng.integers(0, m))] = "ACGT"[int(rng.integers(0, 4))]
            p = "".join(p)
            if rng.random() < 0.5:
                p = "".join(COMP[c] for c in reversed(p))
        else:
            p = "".join(rng.choice(list("ACGT"), size=m))
        pats.append(p)
    return seqs, pats


def scan(lib, seqs, pats, max_mm, term):
    ctx = lib.context(0)
    data = np.frombuffer("".join(seqs).encode(), np.uint8)
    off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    codes = iupac.MASK_LUT[np.frombuffer("".join(pats).encode(), np.uint8)]
    poff = np.zeros(len(pats) + 1, np.int32)
    np.cumsum([len(p) for p in pats], out=poff[1:])
    return [tuple(x) for x in ctx.kmm_scan(data, off, codes, poff, max_mm, term, cap=64).tolist()]     # small cap: the regrow path


@pytest.mark.parametrize("seed", range(6))
def test_oracle_scan_equals_brute_force(seed, oracle_lib):
    rng = np.random.default_rng(seed)
    seqs, pats = make_case(rng, 12, 300, 10)
    for max_mm, term in ((0, 0), (1, 4), (2, 3), (1, 30)):
        assert scan(oracle_lib, seqs, pats, max_mm, term) == brute(seqs, pats, max_mm, term)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_hip_scan_equals_oracle(seed, hip_lib, oracle_lib):
    rng = np.random.default_rng(100 + seed)
    seqs, pats = make_case(rng, 40, 20000, 24)                  # sequences longer than a workgroup's 8192-position segment
    seqs[3] = seqs[3][:8192 + 5]                                # a sequence ending just past a segment border
    for max_mm, term in ((1, 4), (2, 0), (0, 6)):
        assert scan(hip_lib, seqs, pats, max_mm, term) == scan(oracle_lib, seqs, pats, max_mm, term)


def test_script_logic_around_the_mapper(oracle_lib, tmp_path, capsys):
    rng = np.random.default_rng(9)
    fwd, rev = "ACGTTGCAAGGCTTACGA", "TTGACCGGTAACGTCAGT"
    rc_rev = "".join(COMP[c] for c in reversed(rev))
    def rnd(n):
        return "".join(rng.choice(list("ACGT"), size=n))
    g1 = rnd(50) + fwd + rnd(300) + rc_rev + rnd(40)                                  # exact sites, product 318 + ...
    mut = list(fwd)
    mut[5] = "A" if mut[5] != "A" else "C"
    g2 = rnd(20) + "".join(mut) + rnd(500) + rc_rev + rnd(10)                         # one mismatch away from the 3' end
    bad = list(fwd)
    bad[-2] = "A" if bad[-2] != "A" else "C"
    g3 = rnd(30) + "".join(bad) + rnd(400) + rc_rev + rnd(10)                         # mismatch inside the 3' term: rejected
    g4 = rnd(900)
    ref = tmp_path / "ref.fa"
    ref.write_text("".join(f">g{i} desc\n{s}\n" for i, s in enumerate((g1, g2, g3, g4), 1)))
    primers = tmp_path / "primers.fa"
    primers.write_text(f">PF\n{fwd[:-1]}R\n>PR\n{rev}\n")                             # PF degenerate in its last base (A/G)
    out = tmp_path / "val.out"
    app = off_targets(primer_file=str(primers), term_length=0, reference_file=str(ref), PCR_product_size="100,1500", mismatch_num=1,
                      outfile=str(out), term_threshold=4, library=oracle_lib)
    assert bowtie2_mismatch_budget(18) == 1 and bowtie2_mismatch_budget(20) == 2
    assert degenerate_seq("ACRTN") == ["ACATA", "ACATT", "ACATG", "ACATC", "ACGTA", "ACGTT", "ACGTG", "ACGTC"]
    app.run()
    term = (tmp_path / "primers.term.fa").read_text().splitlines()
    assert term == [">PF_0", fwd[:-1] + "A", ">PF_1", fwd[:-1] + "G", ">PR_0", rev]
    lines = out.read_text().splitlines()
    assert lines[0].split("\t") == ["Chrom (or Genes)", "Start", "Stop", "Primer_F", "Primer_R", "Product length"]
    got = [l.split("\t") for l in lines[1:]]
    # g1: forward site at 50, reverse-strand hit of PR at 50 + 18 + 300; g2 likewise with one mismatch; g3 rejected; g4 nothing
    assert got == [["g1", "50", str(50 + 18 + 300), "PF", "PR", str(18 + 300 + 1)], ["g2", "20", str(20 + 18 + 500), "PF", "PR", str(18 + 500 + 1)]]
    assert (tmp_path / "val.out.pair.num").read_text().splitlines() == ["Primer_F\tPrimer_R\tPair_num\ttarget accession number", "PF\tPR\t2\t2"]
    assert (tmp_path / "val.out.total.acc.num").read_text() == "total coverage of primer set (PS) is: 2\n"
    msg = capsys.readouterr().out
    assert "Number of genes with candidate primer pairs: 2." in msg
