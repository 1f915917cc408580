#!/usr/bin/env python3
"""Golden vectors for the second tier of SURVEY §8a — row D (scripts/finDimer.py) and row M
(scripts/get_Maxprimerset.py) — produced by RUNNING the unmodified reference scripts in this
container on (a) the reference's own shipped inputs and (b) seeded synthetic inputs written
here.  Row order of finDimer's output is arrival order of a process pool, so the
hit lines are stored sorted.  Usage: python tests/golden/make_golden_dimer.py
"""
import gzip
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
T = os.path.join(REF, "test_data", "results")
IUPAC = "RYMKSWHBVDN"
COMP = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s.translate(COMP)[::-1]


def synth_primers(seed, n, planted):
    rnd = random.Random(seed)
    prim = []
    for i in range(n):
        L = rnd.choice([12, 15, 17, 18, 18, 18, 19, 20, 22, 24])
        s = [rnd.choice("ACGT") for _ in range(L)]
        prim.append(s)
    for _ in range(planted):                      # plant 3'-end complementarity between random pairs
        i, j = rnd.randrange(n), rnd.randrange(n)
        ln = rnd.randint(5, 11)
        end = "".join(prim[i][-ln:])
        tgt = list(rc(end))
        d2 = rnd.choice([0, 0, 0, 1, 2, 3])
        pos = len(prim[j]) - ln - d2
        if pos >= 0:
            prim[j][pos:pos + ln] = tgt
    for s in prim:                                # sprinkle degenerate symbols afterwards
        for p in range(len(s)):
            if rnd.random() < 0.05:
                s[p] = rnd.choice(IUPAC[:10])
    out = []
    for i, s in enumerate(prim):
        out.append((f">p{i:04d}", "".join(s)))
    out.append((">dup_of_p0003", out[3][1]))      # duplicate sequence: the dict keeps the last name
    out.append((">selfcomp", "ACGTTGCATGCAACGT"))
    return out


def write_fa(path, recs):
    with open(path, "w") as f:
        for name, s in recs:
            f.write(name + "\n" + s + "\n")


def run_findimer(fa, thr=None):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "out.tsv")
        cmd = [sys.executable, os.path.join(REF, "scripts", "finDimer.py"), "-i", fa, "-o", out, "-n", "8"]
        if thr is not None:
            cmd += ["-t", str(thr)]
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
        num = open(out + ".dimer_num").read().splitlines()
        return {"header": lines[0], "hits": sorted(lines[1:]), "dimer_num_header": num[0], "dimer_num": sorted(num[1:])}


def fake_clusters(seed):
    """A multi-cluster candidate file carved out of the shipped single-cluster one: rows of
    different length (stride-5 fields per pair), including a cluster with no pair at all."""
    rnd = random.Random(seed)
    line = open(os.path.join(T, "Primers_set", "candidate_primers_sets.txt")).read().strip().split("\t")
    fields = line[1:]
    pairs = [fields[i:i + 5] for i in range(0, len(fields) - 4, 5)]
    rnd.shuffle(pairs)
    rows, k = [], 0
    for c in range(36):
        n = rnd.choice([1, 2, 3, 5, 8, 13, 40])
        rows.append([f"/fake/Cluster_{c}.candidate.primers.txt"] + [x for p in pairs[k:k + n] for x in p])
        k += n
    rows.insert(7, ["/fake/Cluster_empty.candidate.primers.txt"])
    return rows


def run_maxset(rows, method):
    with tempfile.TemporaryDirectory() as td:
        inp = os.path.join(td, "cand.txt")
        with open(inp, "w") as f:
            for r in rows:
                f.write("\t".join(r) + "\n")
        out = os.path.join(td, "final.xls")
        p = subprocess.run([sys.executable, os.path.join(REF, "scripts", "get_Maxprimerset.py"), "-i", inp, "-s", "5",
                            "-m", method, "-o", out], capture_output=True, text=True)
        res = {"returncode": p.returncode, "stdout": p.stdout.splitlines()}
        res["out"] = open(out).read() if os.path.exists(out) else None
        nxt = out.rstrip(".xls") + ".next.xls"
        res["next"] = open(nxt).read() if os.path.exists(nxt) else None
        res["sort"] = open(os.path.join(td, "sort.cand.txt")).read()
        return res


def main():
    os.makedirs(os.path.join(HERE, "inputs"), exist_ok=True)
    g = {}
    shipped = os.path.join(T, "Clusters_cprimer", "Cluster_0_20727.candidate.primers.txt.fa")
    with open(shipped, "rb") as f:
        open(os.path.join(HERE, "inputs", "cluster0_candidates.fa.gz"), "wb").write(gzip.compress(f.read(), 9, mtime=0))
    g["findimer_cluster0"] = run_findimer(shipped)
    g["findimer_cluster0_t3"] = run_findimer(shipped, 3.0)
    for name, seed, n, planted in (("findimer_syn_a", 101, 260, 160), ("findimer_syn_b", 202, 90, 200)):
        fa = os.path.join(tempfile.gettempdir(), name + ".fa")
        recs = synth_primers(seed, n, planted)
        write_fa(fa, recs)
        open(os.path.join(HERE, "inputs", name + ".fa.gz"), "wb").write(gzip.compress(open(fa, "rb").read(), 9, mtime=0))
        g[name] = run_findimer(fa)
        g[name + "_t3"] = run_findimer(fa, 3.0)
    cand = os.path.join(T, "Primers_set", "candidate_primers_sets.txt")
    open(os.path.join(HERE, "inputs", "candidate_primers_sets.txt.gz"), "wb").write(gzip.compress(open(cand, "rb").read(), 9, mtime=0))
    shipped_rows = [l.rstrip("\n").split("\t") for l in open(cand)]
    shipped_rows = [[x for x in r if x] for r in shipped_rows]
    g["maxset_shipped_T"] = run_maxset(shipped_rows, "T")
    g["maxset_shipped_F"] = run_maxset(shipped_rows, "F")
    for seed in (1, 2, 3):
        rows = fake_clusters(seed)
        g[f"maxset_fake{seed}_rows"] = rows
        g[f"maxset_fake{seed}_T"] = run_maxset(rows, "T")
        g[f"maxset_fake{seed}_F"] = run_maxset(rows, "F")
    raw = json.dumps(g, sort_keys=True).encode()
    open(os.path.join(HERE, "dimer_maxset.json.gz"), "wb").write(gzip.compress(raw, 9, mtime=0))
    for k, v in g.items():
        if k.startswith("findimer"):
            print(k, len(v["hits"]), "hits")
        elif isinstance(v, dict):
            print(k, "rc", v["returncode"], "rows", (v["out"] or "").count("\n"), "next", (v["next"] or "").count("\n"), v["stdout"][:2])


if __name__ == "__main__":
    main()
