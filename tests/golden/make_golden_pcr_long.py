#!/usr/bin/env python3
"""Golden vectors for SURVEY §8f-2 with primers LONGER than 32 nt (adaptor-tailed, 41-52 nt, degenerate bases in the target part):
the unmodified reference script scripts/extract_PCR_product.py on a seeded synthetic database into which amplicons were planted
(forward site, reverse-complemented reverse site; some with a second forward site behind the product, lower case or N inside a
site, a site cut by the sequence end).  Stored like make_golden_pcr.py: every output file by name and sha256, the coverage table
(pair lines sorted).   Run:  python tests/golden/make_golden_pcr_long.py"""
import gzip
import json
import os
import random
import tempfile

from make_golden_pcr import HERE, run

ADAPTOR_F, ADAPTOR_R = "TCTTTCCCTACACGACGCTCTTCCGATCT", "TGGAGTTCAGACGTGTGCTCTTCCGATCT"          # multiPrime.yaml
TABLE = {"R": "AG", "Y": "CT", "M": "AC", "K": "GT", "S": "GC", "W": "AT", "H": "ATC", "B": "GTC", "V": "GAC", "D": "GAT", "N": "ATGC"}
COMP = str.maketrans("ACGT", "TGCA")


def expand_one(rnd, seq):
    return "".join(rnd.choice(TABLE[c]) if c in TABLE else c for c in seq)


def main():
    rnd = random.Random(20250926)
    pairs = []
    for i in range(5):
        f = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(12, 23)))
        r = "".join(rnd.choice("ACGT") for _ in range(rnd.randint(12, 23)))
        if i % 2 == 0:
            p = rnd.randrange(len(f))
            f = f[:p] + rnd.choice("RYKMSW") + f[p + 1:]
            p = rnd.randrange(len(r))
            r = r[:p] + rnd.choice("RYN") + r[p + 1:]
        pairs.append((f"L{i}", ADAPTOR_F + f, ADAPTOR_R + r))
    pairs.append(("S5", "GGTAYGGYYTCAGRCATC", "CRACRTATTTCTCDAGGT"))                       # a short pair in the same call
    fa = "".join(f">{n}_F\n{f}\n>{n}_R\n{r}\n" for n, f, r in pairs)
    seqs = []
    for j in range(160):
        n = rnd.randint(300, 6000)
        s = [rnd.choice("ACGT") for _ in range(n)]
        for _ in range(rnd.randint(0, 3)):
            name, f, r = rnd.choice(pairs)
            fe, re_ = expand_one(rnd, f), expand_one(rnd, r).translate(COMP)[::-1]
            a = rnd.randint(0, max(0, n - 900))
            b = a + len(fe) + rnd.randint(40, 700)
            if b + len(re_) <= n:
                s[a:a + len(fe)] = fe
                s[b:b + len(re_)] = re_
                u = rnd.random()
                if u < 0.2 and b + len(re_) + 80 + len(fe) <= n:
                    s[b + len(re_) + 30:b + len(re_) + 30 + len(fe)] = fe                     # a second forward site behind the product
                elif u < 0.3:
                    s[a + 35] = s[a + 35].lower()                                           # case-sensitive search: no product
                elif u < 0.4:
                    s[b + 3] = "N"
        if j % 40 == 7:                                                                      # a forward site cut by the sequence end
            name, f, r = pairs[0]
            s += list(expand_one(rnd, f)[:-5])
        seqs.append("".join(s))
    ref = "".join(f">seq{j}\n{s}\n" for j, s in enumerate(seqs))
    g = {}
    with tempfile.TemporaryDirectory() as td:
        ref_path, fa_path = os.path.join(td, "long_ref.fa"), os.path.join(td, "long_primers.fa")
        open(ref_path, "w").write(ref)
        open(fa_path, "w").write(fa)
        for name, text in (("pcr_long_ref.fa", ref), ("pcr_long_primers.fa", fa)):
            open(os.path.join(HERE, "inputs", name + ".gz"), "wb").write(gzip.compress(text.encode(), 9, mtime=0))
        g["long_fa"] = run(ref_path, fa_path, "fa", td, "long_fa")
        g["long_seq"] = run(ref_path, pairs[1][1] + "," + pairs[1][2], "seq", td, "long_seq")
        g["long_seq_primers"] = pairs[1][1] + "," + pairs[1][2]
    open(os.path.join(HERE, "pcr_long.json.gz"), "wb").write(gzip.compress(json.dumps(g, sort_keys=True).encode(), 9, mtime=0))


if __name__ == "__main__":
    main()
