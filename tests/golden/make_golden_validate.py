#!/usr/bin/env python3
"""Golden vectors for SURVEY §8f-3 (scripts/primer_coverage_validation_by_BWT.py = ..._V9.py): everything the script does AROUND
the mapper, recorded from the unmodified reference class.  bowtie2 / samtools are not installed here, but the script maps only
when <primers>.for.sam / .rev.sam are missing (V9:270-271), so the cases hand it SAM text — hand-written lines covering the
MD:Z quirks of build_dict (V9:241-262: only the last two characters of the tag are looked at) plus seeded random lines — and
record what it writes: <primers>.term.fa (get_term, V9:205-239), <out>, <out>.pair.num, <out>.total.acc.num and, with -d,
<out>.unmatched.fa (PCR_product and the writers, V9:318-398).  The mapper itself (V9:264-300) stays unpinned.

Orders that follow a Python set in the reference (gene order of <out>, id order inside a term name, unmatched records) are
compared as sorted collections by tests/test_validate.py; everything else verbatim.   Run:  python tests/golden/make_golden_validate.py"""
import gzip
import importlib.util
import json
import os
import pickle
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/scripts/primer_coverage_validation_by_BWT_V9.py"


def load_reference():
    spec = importlib.util.spec_from_file_location("bwt_v9", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bwt_v9"] = mod                  # the class is pickled for the reference's worker pool
    spec.loader.exec_module(mod)
    return mod


def sam_line(read, flag, gene, pos1, length, tags):
    return "\t".join([read, str(flag), gene, str(pos1), "42", f"{length}M", "*", "0", "0", "A" * length, "I" * length] + tags) + "\n"


def expansions(seq):
    table = {"R": "AG", "Y": "CT", "M": "AC", "K": "GT", "S": "GC", "W": "AT", "H": "ATC", "B": "GTC", "V": "GAC", "D": "GAT", "N": "ATGC"}
    out = [""]
    for ch in seq:
        out = [a + b for a in out for b in table.get(ch, ch)]
    return out


def read_names(primers, term_len):
    """The read names get_term will give the expanded terms (single primer name per term in these cases)."""
    names = {}
    for name, seq in primers:
        key = seq if term_len == 0 else seq[-term_len:]
        names.setdefault(key, []).append(name)
    reads = []
    for key, ids in names.items():
        ex = expansions(key)
        for j, e in enumerate(ex):
            reads.append(("_".join(dict.fromkeys(ids)) + "_" + str(j), len(e)))
    return reads


def hand_case():
    primers = [("PF", "ACGTTGCAAGGCTTACGR"), ("PR", "TTGACCGGTAACGTCAGT"), ("QF", "GGATCCATGCAAGCTTAC"), ("QR", "CCGGAATTCGGTACCTTA")]
    f, r = [], []
    # MD:Z forms: all matches; mismatch far from the 3' end; trailing run shorter than the threshold; a zero-length trailing run;
    # two mismatches; a two-digit trailing run; a deletion-style tag; no MD tag at all (unaligned read); NM before MD, extra tags after
    f += [sam_line("PF_0", 0, "g1", 51, 18, ["AS:i:0", "XN:i:0", "XM:i:0", "XO:i:0", "XG:i:0", "NM:i:0", "MD:Z:18", "YT:Z:UU"]),
          sam_line("PF_1", 0, "g1", 61, 18, ["AS:i:-6", "XN:i:0", "XM:i:1", "XO:i:0", "XG:i:0", "NM:i:1", "MD:Z:5A12", "YT:Z:UU"]),
          sam_line("PF_0", 0, "g2", 21, 18, ["AS:i:-6", "XN:i:0", "XM:i:1", "XO:i:0", "XG:i:0", "NM:i:1", "MD:Z:15C2", "YT:Z:UU"]),
          sam_line("PF_1", 0, "g2", 31, 18, ["AS:i:-6", "XN:i:0", "XM:i:1", "XO:i:0", "XG:i:0", "NM:i:1", "MD:Z:17T0", "YT:Z:UU"]),
          sam_line("QF_0", 0, "g2", 41, 18, ["AS:i:-12", "XN:i:0", "XM:i:2", "XO:i:0", "XG:i:0", "NM:i:2", "MD:Z:3A9G4", "YT:Z:UU"]),
          sam_line("QF_0", 0, "g3", 5, 18, ["AS:i:-6", "XN:i:0", "XM:i:1", "XO:i:0", "XG:i:0", "NM:i:1", "MD:Z:7G10", "YT:Z:UU"]),
          sam_line("QF_0", 0, "g3", 300, 18, ["AS:i:-8", "XN:i:0", "XM:i:0", "XO:i:1", "XG:i:1", "NM:i:1", "MD:Z:9^A9", "YT:Z:UU"]),
          sam_line("QF_0", 4, "*", 0, 18, ["YT:Z:UU"]),
          sam_line("PF_0", 0, "g4", 11, 18, ["AS:i:0", "NM:i:0", "MD:Z:18"]),
          sam_line("QF_0", 0, "g4", 11, 18, ["AS:i:0", "NM:i:0", "MD:Z:18"])]          # two primers at one start: dict() keeps the last
    r += [sam_line("PR_0", 16, "g1", 400, 18, ["AS:i:0", "XN:i:0", "XM:i:0", "XO:i:0", "XG:i:0", "NM:i:0", "MD:Z:18", "YT:Z:UU"]),
          sam_line("PR_0", 16, "g1", 1700, 18, ["AS:i:0", "NM:i:0", "MD:Z:18", "YT:Z:UU"]),           # beyond the size range of both forward sites
          sam_line("QR_0", 16, "g1", 1549, 18, ["AS:i:0", "NM:i:0", "MD:Z:18", "YT:Z:UU"]),           # product length exactly at / next to the upper bound
          sam_line("QR_0", 16, "g1", 1550, 18, ["AS:i:0", "NM:i:0", "MD:Z:18", "YT:Z:UU"]),
          sam_line("PR_0", 16, "g2", 600, 18, ["AS:i:-6", "NM:i:1", "MD:Z:2A15", "YT:Z:UU"]),
          sam_line("QR_0", 16, "g3", 105, 18, ["AS:i:0", "NM:i:0", "MD:Z:18", "YT:Z:UU"]),            # product length exactly the lower bound + 1
          sam_line("QR_0", 16, "g3", 104, 18, ["AS:i:0", "NM:i:0", "MD:Z:18", "YT:Z:UU"]),
          sam_line("PR_0", 16, "g4", 200, 18, ["AS:i:0", "NM:i:0", "MD:Z:18", "YT:Z:UU"]),
          sam_line("PR_0", 16, "g5", 200, 18, ["AS:i:0", "NM:i:0", "MD:Z:18", "YT:Z:UU"])]            # reverse only: no pair
    targets = {f"g{i}": f">g{i}\nACGT{i}\n" for i in range(1, 8)}
    return {"name": "hand", "primers": primers, "for_sam": "".join(f), "rev_sam": "".join(r), "term_len": 0, "term_threshold": 4,
            "size": "100,1500", "targets": targets}


def random_case(seed, term_len, threshold, size, n_genes, n_lines, with_targets, shared_terms=False):
    rnd = random.Random(seed)
    bases = "ACGT"
    primers = []
    for i in range(6):
        seq = "".join(rnd.choice(bases) for _ in range(rnd.randint(18, 24)))
        if i % 2 == 0:
            p = rnd.randrange(len(seq) - 9, len(seq))          # a degenerate base near the 3' end, inside an 8-base term too
            seq = seq[:p] + rnd.choice("RYMKSWHBVDN") + seq[p + 1:]
        primers.append((f"P{i}{'F' if i % 2 == 0 else 'R'}", seq))
    if shared_terms:                                           # two primers with the same 3' term: one read name carries both ids
        primers.append(("P6F", "".join(rnd.choice(bases) for _ in range(10)) + primers[1][1][-term_len:]))
    reads = read_names(primers, term_len)
    genes = [f"gene{j}.{rnd.randint(1, 9)}" for j in range(n_genes)]

    def md(length):
        r = rnd.random()
        if r < 0.45:
            return f"MD:Z:{length}"
        if r < 0.85:
            p = rnd.randrange(length)
            return f"MD:Z:{p}{rnd.choice(bases)}{length - 1 - p}"
        p, q = sorted(rnd.sample(range(length), 2))
        return f"MD:Z:{p}{rnd.choice(bases)}{q - p - 1}{rnd.choice(bases)}{length - 1 - q}"

    def lines(flag):
        out = []
        for _ in range(n_lines):
            name, length = rnd.choice(reads)
            tags = [f"AS:i:-{rnd.randint(0, 12)}", "XN:i:0", f"NM:i:{rnd.randint(0, 2)}", md(length), "YT:Z:UU"]
            if rnd.random() < 0.05:
                tags = ["YT:Z:UU"]
            out.append(sam_line(name, flag, rnd.choice(genes), rnd.randint(1, 2500), length, tags))
        return "".join(out)

    targets = {g: f">{g}\n{''.join(rnd.choice(bases) for _ in range(30))}\n" for g in genes + ["absent.1", "absent.2"]} if with_targets else None
    return {"name": f"rand{seed}", "primers": primers, "for_sam": lines(0), "rev_sam": lines(16), "term_len": term_len,
            "term_threshold": threshold, "size": size, "targets": targets}


def run_case(mod, case):
    with tempfile.TemporaryDirectory() as td:
        pf = os.path.join(td, "primers.fa")
        open(pf, "w").write("".join(f">{n}\n{s}\n" for n, s in case["primers"]))
        open(os.path.join(td, "primers.for.sam"), "w").write(case["for_sam"])
        open(os.path.join(td, "primers.rev.sam"), "w").write(case["rev_sam"])
        tpath = "None"
        if case["targets"] is not None:
            tpath = os.path.join(td, "targets.pkl")
            pickle.dump(case["targets"], open(tpath, "wb"))
        out = os.path.join(td, "val.out")
        app = mod.off_targets(primer_file=pf, term_length=case["term_len"], reference_file=os.path.join(td, "unused_index"),
                              PCR_product_size=case["size"], mismatch_num=1, outfile=out, term_threshold=case["term_threshold"],
                              bowtie="bowtie2", nproc=2, targets=tpath)
        app.run()
        res = {"term_fa": open(os.path.join(td, "primers.term.fa")).read(), "out": open(out).read(), "pair_num": open(out + ".pair.num").read(),
               "total_acc_num": open(out + ".total.acc.num").read()}
        if case["targets"] is not None:
            res["unmatched_fa"] = open(out + ".unmatched.fa").read()
        return res


def main():
    mod = load_reference()
    cases = [hand_case(),
             random_case(1, 0, 4, "100,1500", 12, 400, True),
             random_case(2, 8, 4, "150,1200", 6, 260, False, shared_terms=True),
             random_case(3, 0, 0, "100,1500", 3, 160, True),
             random_case(4, 12, 6, "50,400", 25, 500, False),
             random_case(5, 0, 4, "100,1500", 40, 120, True)]          # sparse: many genes without a pair, early exits of PCR_product
    for case in cases:
        case["recorded"] = run_case(mod, case)
        print(case["name"], {k: len(v.splitlines()) for k, v in case["recorded"].items()})
    open(os.path.join(HERE, "validate.json.gz"), "wb").write(gzip.compress(json.dumps({"cases": cases}, sort_keys=True).encode(), 9, mtime=0))


if __name__ == "__main__":
    main()
