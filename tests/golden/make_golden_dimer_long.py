#!/usr/bin/env python3
"""Golden vectors for primers of MORE than 32 bases (adaptor-tailed primers, up to 64 nt) through the reference's finDimer.py and
get_Maxprimerset.py, produced by RUNNING the unmodified scripts in this container (same harness as make_golden_dimer.py).
Usage: python tests/golden/make_golden_dimer_long.py"""
import gzip
import json
import os
import random
import tempfile

import make_golden_dimer as base

HERE = os.path.dirname(os.path.abspath(__file__))
ADAPTORS = ["TCTTTCCCTACACGACGCTCTTCCGATCT", "GTGACTGGAGTTCAGACGTGTGCTCTTCCGATCT", "AATGATACGGCGACCACCGAGATCTACAC", "CAAGCAGAAGACGGCATACGAGAT",
            "ACACTCTTTCCCTACACGACGCTCTTCCGATCTNN"]


def long_primers(seed, n, planted):
    rnd = random.Random(seed)
    recs = base.synth_primers(seed, n, planted)[:-2]
    out = []
    for i, (name, s) in enumerate(recs):
        kind = rnd.random()
        if kind < 0.65:
            s = rnd.choice(ADAPTORS) + s                 # 5' tail: the 3' end keeps the planted complementarity
        elif kind < 0.75:
            s = s + base.rc(rnd.choice(ADAPTORS))[:rnd.randint(10, 30)]
        s = s[:64] if len(s) > 64 else s
        out.append((name, s))
    out.append((">tail_only", ADAPTORS[1]))
    out.append((">selfcomp_long", "GGATCCGGATCCAAGCTTAAGCTTGGATCCGGATCCAAGCTTAAGCTT"))
    return out


def main():
    g = {}
    for name, seed, n, planted in (("findimer_long_a", 11, 110, 140), ("findimer_long_b", 12, 60, 120)):
        fa = os.path.join(tempfile.gettempdir(), name + ".fa")
        recs = long_primers(seed, n, planted)
        base.write_fa(fa, recs)
        open(os.path.join(HERE, "inputs", name + ".fa.gz"), "wb").write(gzip.compress(open(fa, "rb").read(), 9, mtime=0))
        g[name] = base.run_findimer(fa)
        g[name + "_t3"] = base.run_findimer(fa, 3.0)
        g[name + "_lengths"] = sorted({len(s) for _, s in recs})
    # get_Maxprimerset on clusters whose candidate primers carry the adaptors
    rnd = random.Random(5)
    for seed in (1, 2):
        rows = base.fake_clusters(seed)
        tailed = []
        for r in rows:
            r = list(r)
            for i in range(1, len(r) - 4, 5):          # fields: F, R, product:Tm:coverage, ... (stride 5); primers are fields 0 and 1 of a pair
                for j in (i, i + 1):
                    if rnd.random() < 0.7:
                        r[j] = rnd.choice(ADAPTORS[:4]) + r[j]
            tailed.append(r)
        g[f"maxset_long{seed}_rows"] = tailed
        g[f"maxset_long{seed}_T"] = base.run_maxset(tailed, "T")
        g[f"maxset_long{seed}_F"] = base.run_maxset(tailed, "F")
    raw = json.dumps(g, sort_keys=True).encode()
    open(os.path.join(HERE, "dimer_long.json.gz"), "wb").write(gzip.compress(raw, 9, mtime=0))
    for k, v in g.items():
        if k.startswith("findimer") and isinstance(v, dict):
            print(k, len(v["hits"]), "hits")
        elif k.endswith("_lengths"):
            print(k, v)
        elif isinstance(v, dict):
            print(k, "rc", v["returncode"], "rows", (v["out"] or "").count("\n"), "next", (v["next"] or "").count("\n"), v["stdout"][:2])


if __name__ == "__main__":
    main()
