#!/usr/bin/env python3
"""More goldens for SURVEY §8a row M (scripts/get_Maxprimerset.py), produced by RUNNING the unmodified reference script:
multi-cluster candidate files whose primers really dimerise ACROSS clusters, so that the greedy cover has to skip pairs
(-m T writes clusters to .next.xls) and the maximum-set search (-m F) has to back-track to an earlier cluster and take
its next pair — or give up with exit status 1.  Seeded synthetic inputs; the scenarios are built so that both outcomes
occur.  Usage: python tests/golden/make_golden_maxset2.py   (writes tests/golden/maxset_multi.json.gz)"""
import gzip
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_dimer import rc, run_maxset  # noqa: E402


def rnd_primer(rnd, n=18):
    return "".join(rnd.choice("ACGT") for _ in range(n))


def pair(rnd, i):
    f, r = rnd_primer(rnd), rnd_primer(rnd)
    return [f, r, f"{rnd.randint(150, 900)}:{rnd.uniform(45, 62):.2f}:{rnd.uniform(0.7, 1):.3f}", str(rnd.randint(200, 500)),
            f"{rnd.randint(10, 500)}:{rnd.randint(600, 1500)}"]


def plant(victim, source, ln, d2):
    """Make `victim` carry the reverse complement of source's 3' end, d2 bases away from its own 3' end."""
    t = rc(source[-ln:])
    pos = len(victim) - ln - d2
    return victim[:pos] + t + victim[pos + ln:]


def scenario(seed):
    rnd = random.Random(seed)
    n_clusters = rnd.randint(6, 12)
    rows = []
    for c in range(n_clusters):
        n = rnd.choice([1, 2, 2, 3, 4, 6])
        rows.append([f"/syn/Cluster_{seed}_{c}.candidate.primers.txt"] + [x for i in range(n) for x in pair(rnd, i)])
    # cross-cluster conflicts: every pair of cluster b dimerises with the FIRST pair of cluster a (a sorts before b),
    # so -m F must return to a and take its second pair; -m T drops b or the later pairs
    for _ in range(rnd.randint(2, 4)):
        a, b = sorted(rnd.sample(range(n_clusters), 2), key=lambda i: len(rows[i]))
        if len(rows[a]) < 11 or len(rows[a]) == len(rows[b]):
            continue
        src = rows[a][1]
        for col in range(1, len(rows[b]) - 4, 5):
            which = col + rnd.randint(0, 1)
            rows[b][which] = plant(rows[b][which], src, rnd.randint(5, 8), rnd.choice([0, 0, 1]))
    # some conflicts that hit every pair of a cluster from a single-pair cluster: no way out for -m F (exit 1)
    if seed % 3 == 0:
        singles = [i for i, r in enumerate(rows) if len(r) == 6]
        multi = [i for i, r in enumerate(rows) if len(r) > 6]
        if singles and multi:
            a, b = singles[0], multi[-1]
            for col in range(1, len(rows[b]) - 4, 5):
                rows[b][col] = plant(rows[b][col], rows[a][2], 7, 0)
    if seed % 2 == 0:
        rows.insert(rnd.randrange(len(rows)), [f"/syn/Cluster_{seed}_empty.candidate.primers.txt"])
    # a few degenerate symbols
    for r in rows:
        for col in range(1, len(r) - 4, 5):
            for which in (col, col + 1):
                s = list(r[which])
                for p in range(2, 10):
                    if rnd.random() < 0.04:
                        s[p] = rnd.choice("RYMKSW")
                r[which] = "".join(s)
    return rows


def main():
    g = {}
    for seed in range(1, 9):
        rows = scenario(seed)
        g[f"maxset_multi{seed}_rows"] = rows
        for method in ("T", "F"):
            g[f"maxset_multi{seed}_{method}"] = run_maxset(rows, method)
            v = g[f"maxset_multi{seed}_{method}"]
            print(seed, method, "rc", v["returncode"], "rows", (v["out"] or "").count("\n"), "next", (v["next"] or "").count("\n"), v["stdout"][:3])
    open(os.path.join(HERE, "maxset_multi.json.gz"), "wb").write(gzip.compress(json.dumps(g, sort_keys=True).encode(), 9, mtime=0))


if __name__ == "__main__":
    main()
