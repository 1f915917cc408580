#!/usr/bin/env python3
"""Golden vectors for SURVEY §8f-2 (exact in-silico PCR, scripts/extract_PCR_product.py): the unmodified
reference script is run on the shipped inputs and on seeded variants (lower case, N, duplicated forward
site, multi-pair fasta / seq formats); every file it writes is stored by name and sha256, the coverage
table verbatim (pair lines sorted: the reference's order is pool arrival order)."""
import gzip
import hashlib
import json
import os
import random
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
T = os.path.join(REF, "test_data")


def sha(b):
    return hashlib.sha256(b).hexdigest()


def variant_ref(src_lines, seed):
    """A 2-line FASTA derived from real sequences with the quirks the search must respect."""
    rnd = random.Random(seed)
    out = []
    for i in range(0, len(src_lines), 2):
        h, s = src_lines[i], src_lines[i + 1]
        r = rnd.random()
        if r < 0.1:
            s = s.lower()                                   # case-sensitive search: never matches
        elif r < 0.2:
            p = rnd.randrange(len(s))
            s = s[:p] + "N" * rnd.randint(1, 3) + s[p + 1:]
        elif r < 0.45:                                      # repeat the forward primer site 80 nt downstream: the reference's
            import re                                       # str.split then cuts the "Product" before the reverse site
            m = re.search("GGTA[CT]GG[CT][CT]TCAG[AG]CATC", s)
            if m:
                a = m.start()
                chunk = s[max(0, a - 5):a + 30]
                b = min(len(s), a + 80)
                s = s[:b] + chunk + s[b:]
        out += [h, s]
    return out


def run(ref_path, primer_arg, fmt, td, tag):
    od = os.path.join(td, "out_" + tag)
    cov = os.path.join(td, "cov_" + tag + ".xls")
    p = subprocess.run([sys.executable, os.path.join(REF, "scripts", "extract_PCR_product.py"), "-i", primer_arg, "-r", ref_path,
                        "-f", fmt, "-o", od, "-s", cov, "-p", "4"], capture_output=True, text=True)
    files = {}
    for fn in sorted(os.listdir(od)) if os.path.isdir(od) else []:
        b = open(os.path.join(od, fn), "rb").read()
        files[fn] = {"sha256": sha(b), "records": b.count(b">"), "head": b[:160].decode()}
    cov_lines = open(cov).read().splitlines() if os.path.exists(cov) else []
    pair_lines = sorted(l for l in cov_lines if l.startswith("Number of"))
    rest = [l for l in cov_lines if not l.startswith("Number of")]
    print(tag, "rc", p.returncode, "files", len(files), rest)
    return {"returncode": p.returncode, "files": files, "coverage_pairs": pair_lines, "coverage_totals": rest}


def main():
    g = {}
    with tempfile.TemporaryDirectory() as td:
        tfa = os.path.join(T, "results", "Clusters_fa", "Cluster_0_20727.tfa")
        fasta1000 = os.path.join(T, "1000.fasta")
        for name, src in (("Cluster_0_20727.tfa", tfa), ("1000.fasta", fasta1000)):
            open(os.path.join(HERE, "inputs", name + ".gz"), "wb").write(gzip.compress(open(src, "rb").read(), 9, mtime=0))
        xls = os.path.join(T, "results", "Primers_set", "final_maxprimers_set.xls")
        open(os.path.join(HERE, "inputs", "final_maxprimers_set.xls.gz"), "wb").write(gzip.compress(open(xls, "rb").read(), 9, mtime=0))
        g["shipped_xls"] = run(tfa, xls, "xls", td, "shipped_xls")
        g["seq_format"] = run(tfa, "GGTAYGGYYTCAGRCATC,CRACRTATTTCTCDAGGT", "seq", td, "seq")
        # fasta format: the primer pairs the pairing stage found on 1000_fasta.msa (recorded golden) + two hand-made ones
        pg = json.loads(gzip.open(os.path.join(HERE, "pairing.json.gz")).read())
        c0 = pg["results"]["cluster0_v1"]["yaml"]["fa"].split(">")[1:25]           # the 12 best Cluster_0 pairs
        fa = pg["results"]["msa1000_k18_d64"]["yaml"]["fa"] + "".join(">" + x for x in c0)
        fa += ">shipped_F\nGGTAYGGYYTCAGRCATC\n>shipped_R\nCRACRTATTTCTCDAGGT\n"
        # Path.with_suffix drops the reverse record's part of the output file name (PCR:224-226), so pairs that share a
        # forward record overwrite each other's files in pool-arrival order: keep one pair per forward name
        recs = fa.strip().split("\n")
        seen, kept = set(), []
        for i in range(0, len(recs), 4):
            stem = (recs[i].lstrip(">") + "_" + recs[i + 2].lstrip(">")).rsplit(".", 1)[0]
            if stem not in seen:
                seen.add(stem)
                kept += recs[i:i + 4]
        fa = "\n".join(kept) + "\n"
        fa += ">hand_1F\nGGCTTTTAAAAGTTCTGTTCC\n>hand_1R\nCCTCTTACAAAGATGCAGTC\n"
        fa_path = os.path.join(td, "primers.fa")
        open(fa_path, "w").write(fa)
        open(os.path.join(HERE, "inputs", "pcr_primers.fa.gz"), "wb").write(gzip.compress(fa.encode(), 9, mtime=0))
        g["fa_1000"] = run(fasta1000, fa_path, "fa", td, "fa_1000")
        for seed in (1, 2):
            lines = open(tfa).read().splitlines()
            var = "\n".join(variant_ref(lines, seed)) + "\n"
            vp = os.path.join(td, f"variant{seed}.fa")
            open(vp, "w").write(var)
            open(os.path.join(HERE, "inputs", f"pcr_variant{seed}.fa.gz"), "wb").write(gzip.compress(var.encode(), 9, mtime=0))
            g[f"variant{seed}_fa"] = run(vp, fa_path, "fa", td, f"variant{seed}")
    open(os.path.join(HERE, "pcr.json.gz"), "wb").write(gzip.compress(json.dumps(g, sort_keys=True).encode(), 9, mtime=0))


if __name__ == "__main__":
    main()
