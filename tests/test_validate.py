"""the k-mismatch primer-site scan (mp_kmm_scan) and the drop-in of primer_coverage_validation_by_BWT.py.

Pinned to the reference: everything the script does around the mapper — tests/golden/validate.json.gz holds what the unmodified
V9 class writes for hand-written and seeded SAM input (make_golden_validate.py); the drop-in must write the same from the same SAM
text.  The mapper's decisions are pinned in tests/test_validate_bwt.py (the reference author's bowtie2 run).  Here: the oracle's scan against a brute-force Python
statement of the acceptance rule in mprime.h, the HIP kernel against the oracle (`-m gpu`; segment borders, both strands, N / lower
case, patterns of several lengths), and scan -> sites == SAM lines of the same alignments -> sites, end to end."""
import os

import numpy as np
import pytest

from multiprime_amd import iupac
from conftest import load_gz_json
from multiprime_amd.validate import TermTable, amplicons, bowtie2_mismatch_budget, degenerate_seq, off_targets, sites_of_sam

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


def make_case(rng, n_rows, max_len, n_pat, pat_len=(6, 25)):
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
        m = int(rng.integers(*pat_len))
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


def test_oracle_scan_equals_brute_force_on_long_patterns(oracle_lib):
    rng = np.random.default_rng(77)
    seqs, pats = make_case(rng, 10, 400, 8, pat_len=(30, 65))           # up to MP_PATTERN_MAX_LEN = 64 bases
    assert max(len(p) for p in pats) > 40
    for max_mm, term in ((2, 4), (4, 0)):
        assert scan(oracle_lib, seqs, pats, max_mm, term) == brute(seqs, pats, max_mm, term)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_hip_scan_equals_oracle_on_long_patterns(seed, hip_lib, oracle_lib):
    rng = np.random.default_rng(300 + seed)
    seqs, pats = make_case(rng, 30, 20000, 16, pat_len=(20, 65))         # the two-word kernel (one pattern beyond 32 bases)
    for max_mm, term in ((2, 4), (5, 0), (0, 8)):
        assert scan(hip_lib, seqs, pats, max_mm, term) == scan(oracle_lib, seqs, pats, max_mm, term)


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


# ---- pinned to the reference: the stages around the mapper --------------------------------------------------------------------
def _canon_read_name(name):
    """A term shared by several primers is named after a set() of them in the reference: compare the names as a sorted group."""
    *owners, index = name.split("_")
    return "_".join(sorted(owners)) + "_" + index


def _write_case(case, tmp_path):
    primers = tmp_path / "primers.fa"
    primers.write_text("".join(f">{n}\n{s}\n" for n, s in case["primers"]))
    (tmp_path / "primers.for.sam").write_text(case["for_sam"])
    (tmp_path / "primers.rev.sam").write_text(case["rev_sam"])
    targets = "None"
    if case["targets"] is not None:
        import pickle
        targets = str(tmp_path / "targets.pkl")
        with open(targets, "wb") as f:
            pickle.dump(case["targets"], f)
    return primers, targets


@pytest.mark.parametrize("index", range(6))
def test_stages_around_the_mapper_equal_the_reference(index, tmp_path, capsys):
    case = load_gz_json("validate.json.gz")["cases"][index]
    want = case["recorded"]
    primers, targets = _write_case(case, tmp_path)
    out = tmp_path / "val.out"
    off_targets(primer_file=str(primers), term_length=case["term_len"], reference_file=str(tmp_path / "unused_index"), PCR_product_size=case["size"],
                mismatch_num=1, outfile=str(out), term_threshold=case["term_threshold"], targets=targets).run()     # SAM files exist: no mapping, no GPU
    # <primers>.term.fa: records in the reference's order, owner names of a shared term as a group
    got_term, want_term = (tmp_path / "primers.term.fa").read_text().splitlines(), want["term_fa"].splitlines()
    assert got_term[1::2] == want_term[1::2]
    assert [_canon_read_name(x[1:]) for x in got_term[0::2]] == [_canon_read_name(x[1:]) for x in want_term[0::2]]
    # <out>: the reference walks the sequences in set() order; inside a sequence the order is fixed
    got, ref = out.read_text().splitlines(), want["out"].splitlines()
    assert got[0] == ref[0]

    def by_gene(lines):
        d = {}
        for line in lines[1:]:
            d.setdefault(line.split("\t")[0], []).append(line)
        return d
    assert by_gene(got) == by_gene(ref)
    # <out>.pair.num: counts in falling order; ties follow the sequence order, so compare as a set of lines
    gp, rp = (tmp_path / "val.out.pair.num").read_text().splitlines(), want["pair_num"].splitlines()
    assert gp[0] == rp[0] and sorted(gp[1:]) == sorted(rp[1:])
    counts = [int(line.split("\t")[2]) for line in gp[1:]]
    assert counts == sorted(counts, reverse=True)
    assert (tmp_path / "val.out.total.acc.num").read_text() == want["total_acc_num"]
    if case["targets"] is not None:
        def records(text):
            return sorted(">" + r for r in text.split(">")[1:])
        assert records((tmp_path / "val.out.unmatched.fa").read_text()) == records(want["unmatched_fa"])
    capsys.readouterr()


def test_md_tag_rule_reads_the_last_two_characters_only(tmp_path):
    """The quirks of build_dict (V9:241-262) one by one; the recorded hand case holds the same lines."""
    def line(md, start=10, read="P_0"):
        tags = ["AS:i:0", "NM:i:0"] + ([md] if md else [])
        return "\t".join([read, "0", "g", str(start), "42", "18M", "*", "0", "0", "A" * 18, "I" * 18] + tags) + "\n"
    sam = tmp_path / "x.sam"
    sam.write_text(line("MD:Z:18", 1) + line("MD:Z:5A12", 2) + line("MD:Z:15C2", 3) + line("MD:Z:17T0", 4) + line("MD:Z:3A9G4", 5) +
                   line("MD:Z:2A10", 6) + line("MD:Z:9^A9", 7) + line(None, 8) + line("MD:Z:18", 9, "Q_1_R_0"))
    got = sites_of_sam(sam, 4)
    # 15C2 and 17T0 end in runs below the threshold; "2A10" ends in a run of 10; the tag stops at '^' (\\w+), leaving "9"
    assert got == {"g": {0: "P", 1: "P", 4: "P", 5: "P", 6: "P", 8: "Q_1_R"}}
    assert sites_of_sam(sam, 0)["g"].keys() == {0, 1, 2, 3, 4, 5, 6, 8}


def test_amplicons_keeps_the_reference_quirks():
    f = {100: "F1", 120: "F2", 5000: "F3", 5100: "F4"}
    r = {400: "R1", 1599: "R2", 1600: "R3", 5300: "R4"}
    # from 100: 400 (301 long); 1599 would be exactly size_hi long — skipped, and not the end of the search.  From 120: 400, 1599, 1600
    assert amplicons(f, r, 100, 1500) == [(100, 400, "F1", "R1", 301), (120, 400, "F2", "R1", 281), (120, 1599, "F2", "R2", 1480),
                                          (120, 1600, "F2", "R3", 1481), (5000, 5300, "F3", "R4", 301), (5100, 5300, "F4", "R4", 201)]
    # the first start without a partner ends the search: 3000 has none, so 5000 and 5100 are never looked at
    f2 = dict(f)
    f2[3000] = "Fx"
    assert amplicons(f2, r, 100, 1500) == [(100, 400, "F1", "R1", 301), (120, 400, "F2", "R1", 281), (120, 1599, "F2", "R2", 1480),
                                           (120, 1600, "F2", "R3", 1481)]
    assert amplicons({10: "F"}, {5000: "R"}, 100, 1500) == [] and amplicons({10: "F"}, {50: "R"}, 100, 1500) == []


# ---- the mapper's replacement, end to end: scan -> sites  ==  SAM lines of the same alignments -> sites ------------------------------
def _sam_of_alignments(seqs, names, reads, read_names, max_mm):
    """Every ungapped alignment with at most max_mm mismatches as a SAM line with its MD:Z tag (reference orientation)."""
    out = ([], [])
    for i, p in enumerate(reads):
        m = len(p)
        for strand in (0, 1):
            q = p if strand == 0 else "".join(COMP[c] for c in reversed(p))
            for g, s in zip(names, seqs):
                s = s.upper()
                for pos in range(0, len(s) - m + 1):
                    mis = [j for j in range(m) if s[pos + j] != q[j]]
                    if len(mis) > max_mm:
                        continue
                    md, prev = "", 0
                    for j in mis:
                        md += str(j - prev) + s[pos + j]
                        prev = j + 1
                    md += str(m - prev)
                    out[strand].append("\t".join([read_names[i], "16" if strand else "0", g, str(pos + 1), "42", f"{m}M", "*", "0", "0", q, "I" * m,
                                                   "AS:i:0", f"NM:i:{len(mis)}", "MD:Z:" + md]) + "\n")
    return out


def _scan_equals_sam(lib, tmp_path, seed):
    rng = np.random.default_rng(seed)
    seqs, pats = make_case(rng, 8, 1500, 6)
    pats = [p[:18] if len(p) > 18 else p for p in pats]
    names = [f"g{i}" for i in range(len(seqs))]
    ref = tmp_path / "ref.fa"
    ref.write_text("".join(f">{n} x\n{s}\n" for n, s in zip(names, seqs)))
    primers = tmp_path / "p.fa"
    primers.write_text("".join(f">P{i}\n{p}\n" for i, p in enumerate(pats)))
    app = off_targets(primer_file=str(primers), term_length=0, reference_file=str(ref), PCR_product_size="50,1200", mismatch_num=1,
                      outfile=str(tmp_path / "o"), term_threshold=4, library=lib, max_mismatch=2)
    table = TermTable(str(primers), 0)
    got_f, got_r = app.scan(table)
    sam_f, sam_r = _sam_of_alignments(seqs, names, list(table.reads), table.names(), 2)
    (tmp_path / "f.sam").write_text("".join(sam_f))
    (tmp_path / "r.sam").write_text("".join(sam_r))
    want_f, want_r = sites_of_sam(tmp_path / "f.sam", 4), sites_of_sam(tmp_path / "r.sam", 4)
    assert {g: set(v) for g, v in got_f.items()} == {g: set(v) for g, v in want_f.items()}
    assert {g: set(v) for g, v in got_r.items()} == {g: set(v) for g, v in want_r.items()}
    assert sum(len(v) for v in want_f.values()) + sum(len(v) for v in want_r.values()) > 0


@pytest.mark.parametrize("seed", range(3))
def test_scan_sites_equal_sam_sites_of_the_same_alignments(seed, oracle_lib, tmp_path):
    _scan_equals_sam(oracle_lib, tmp_path, 40 + seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_scan_sites_equal_sam_sites_on_the_gpu(seed, hip_lib, tmp_path):
    _scan_equals_sam(hip_lib, tmp_path, 40 + seed)


def test_reads_the_scan_cannot_take_are_skipped_by_name(oracle_lib, tmp_path, capsys):
    """The reference hands every read to bowtie2, which ignores what it cannot align: a blank line of the primer file (V9 get_term
    keeps it as read ""), non-ACGT letters.  Here they are skipped with a warning naming them; only a primer file without ANY
    usable read is an error.  A read beyond the scan's width is NOT skipped (bowtie2 would have mapped it): a ValueError names it."""
    ref = tmp_path / "ref.fa"
    ref.write_text(">g\n" + "ACGT" * 40 + "\n")
    primers = tmp_path / "p.fa"
    primers.write_text(">long\n" + "ACGT" * 17 + "\n")                          # 68 nt, whole primer: beyond MP_PATTERN_MAX_LEN
    app = off_targets(primer_file=str(primers), term_length=0, reference_file=str(ref), PCR_product_size="50,1200", mismatch_num=1,
                      outfile=str(tmp_path / "o"), term_threshold=4, library=oracle_lib)
    with pytest.raises(ValueError, match="longer than the scan's 64 bases: 'long_0'"):
        app.run()
    both = tmp_path / "both.fa"
    both.write_text(">f\nACGTACGTACGTACGTAC\n>long\n" + "ACGT" * 17 + "\n")       # a usable read beside it does not make the long one skippable
    with pytest.raises(ValueError, match="long_0"):
        off_targets(primer_file=str(both), term_length=0, reference_file=str(ref), PCR_product_size="50,1200", mismatch_num=1,
                    outfile=str(tmp_path / "o"), term_threshold=4, library=oracle_lib).run()
    off_targets(primer_file=str(primers), term_length=20, reference_file=str(ref), PCR_product_size="50,1200", mismatch_num=1,
                outfile=str(tmp_path / "o"), term_threshold=4, library=oracle_lib).run()      # its 20-base 3' term is fine
    # a blank line and a read with a letter no expansion removes, beside a good primer pair: the run completes, the good pair is found
    mixed = tmp_path / "m.fa"
    mixed.write_text(">f\nACGTACGTACGTACGTAC\n\n>u\nACGUACGTACGTACGTAC\n>r\nGTACGTACGTACGTACGT\n")
    out = tmp_path / "mixed.out"
    off_targets(primer_file=str(mixed), term_length=0, reference_file=str(ref), PCR_product_size="20,1200", mismatch_num=1,
                outfile=str(out), term_threshold=4, library=oracle_lib).run()
    err = capsys.readouterr().err
    assert "not scanned" in err and "u_0" in err
    assert out.exists() and len(out.read_text().splitlines()) > 1
