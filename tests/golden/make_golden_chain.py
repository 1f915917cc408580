#!/usr/bin/env python3
"""Goldens for BASELINE config 5 (multi-cluster run), produced by RUNNING the unmodified reference chain as its
Snakemake rules would (no reference source is copied; the scripts are started as subprocesses):

  rule multiPrime                  scripts/multiPrime-core.py     multiPrime.py:200-207    {i}.top.primer.out (+ 2 JSON)
  rule get_multiPrime              scripts/get_multiPrime.py      multiPrime.py:232-238    {i}.candidate.primers.txt
  rule aggregate_candidate_primers cat                            multiPrime.py:253-256    candidate_primers_sets.txt
  rule get_Maxprimerset            scripts/get_Maxprimerset.py    multiPrime.py:277-295    final_maxprimers_set.xls (+ .next.xls)
  rule all_mfeprimer_check         scripts/primerset_format.py,
                                   scripts/finDimer.py            multiPrime.py:396-415    final_maxprimers_set.fa(.findimer, .dimer_num)

on eight small seeded synthetic clusters (multiprime_amd.synth, the generator of SURVEY §8d input 4/5; sizes kept small
because the reference core takes ~0.1 s per window).  Flags are multiPrime.yaml's.  Stored: the gzipped cluster FASTA files
and every file the chain wrote, in tests/golden/chain.json.gz.  Environment as SURVEY Appendix A-14.

Usage: python tests/golden/make_golden_chain.py                                      -> chain.json.gz (-l 18, all eight clusters)
       python tests/golden/make_golden_chain.py --length 36 --clusters 0,1,3,5,7    -> chain_k36.json.gz (primers longer than one 32-bit word)
"""
import argparse
import gzip
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
S = os.path.join(REF, "scripts")
ENV = dict(os.environ, PYTHONHASHSEED="0",
           NPY_DISABLE_CPU_FEATURES="AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3")
ADAPTOR = "TCTTTCCCTACACGACGCTCTTCCGATCT,TGGAGTTCAGACGTGTGCTCTTCCGATCT"
# (rows, columns, seed, generator settings): a clean cluster, one with ragged edges, one with IUPAC codes, one deeper
CLUSTERS = [
    (48, 520, 101, dict()),
    (72, 610, 202, dict(edge_frac=0.3, p_gap=0.004)),
    (96, 480, 303, dict(p_iupac=4e-4)),
    (150, 700, 404, dict(p_sub=0.02)),
    (30, 450, 505, dict(p_sub=0.03, var_frac=0.1)),
    (64, 560, 606, dict(edge_frac=0.2)),
    (110, 640, 707, dict(p_gap=0.006, p_iupac=2e-4)),
    (40, 500, 808, dict(p_sub=0.01)),
]


def run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout.decode(errors="replace"), r.stderr.decode(errors="replace")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--length", type=int, default=18, help="-l of the core step (multiPrime.yaml: 18)")
    ap.add_argument("--clusters", default=None, help="comma-separated indices into CLUSTERS (default: all)")
    args = ap.parse_args()
    picked = [int(x) for x in args.clusters.split(",")] if args.clusters else list(range(len(CLUSTERS)))
    sys.path.insert(0, REPO)
    from multiprime_amd.synth import synth_block, to_fasta
    out = {"flags": "multiPrime.yaml" + ("" if args.length == 18 else f" with -l {args.length}"), "primer_length": args.length, "clusters": [], "files": {}, "stdout": {}}
    with tempfile.TemporaryDirectory() as wd:
        names = []
        for i, (n, L, seed, kw) in enumerate(CLUSTERS):
            if i not in picked:
                continue
            name = f"Cluster_{i}_{n}"
            names.append(name)
            data = to_fasta(synth_block(0, n, L, seed, block_rows=256, **kw))
            # the rule chain hands the SAME records to the core step (aligned, .tmsa) and to the pairing step (.tfa, unaligned there;
            # get_multiPrime only counts its '>' lines): the aligned file serves as both
            with open(os.path.join(wd, name + ".tfa"), "wb") as f:
                f.write(data)
            out["clusters"].append({"name": name, "rows": n, "columns": L, "seed": seed, "settings": kw,
                                    "fasta_gz_hex": gzip.compress(data, 9, mtime=0).hex()})

        def core_and_pairing(name):
            fa = os.path.join(wd, name + ".tfa")
            top = os.path.join(wd, name + ".top.primer.out")
            rc1 = run([sys.executable, os.path.join(S, "multiPrime-core.py"), "-i", fa, "-n", "4", "-d", "10", "-v", "1", "-c", "2,3,-1",
                       "-g", "0.2,0.7", "-s", "150", "-l", str(args.length), "-e", "3.6", "-o", top, "-f", "0.7", "-p", "1"], wd)
            cand = os.path.join(wd, name + ".candidate.primers.txt")
            rc2 = run([sys.executable, os.path.join(S, "get_multiPrime.py"), "-i", top, "-r", fa, "-f", "0.7", "-s", "150,1200",
                       "-g", "0.2,0.7", "-e", "4", "-d", "4", "-a", ADAPTOR, "-m", "0", "-o", cand, "-p", "1"], wd)
            return name, rc1[0], rc2[0]

        with ThreadPoolExecutor(8) as ex:
            for name, a, b in ex.map(core_and_pairing, names):
                print(name, "core exit", a, "pairing exit", b, flush=True)
                out["stdout"][name] = {"core_exit": a, "pairing_exit": b}
        agg = os.path.join(wd, "candidate_primers_sets.txt")
        with open(agg, "wb") as f:
            for name in names:
                p = os.path.join(wd, name + ".candidate.primers.txt")
                if os.path.exists(p):
                    f.write(open(p, "rb").read())
        final = os.path.join(wd, "final_maxprimers_set.xls")
        rc = run([sys.executable, os.path.join(S, "get_Maxprimerset.py"), "-i", agg, "-s", "5", "-m", "T", "-o", final], wd)
        out["stdout"]["get_Maxprimerset"] = {"exit": rc[0]}
        fa = os.path.join(wd, "final_maxprimers_set.fa")
        rc = run([sys.executable, os.path.join(S, "primerset_format.py"), "-i", final, "-o", fa], wd)
        out["stdout"]["primerset_format"] = {"exit": rc[0]}
        rc = run([sys.executable, os.path.join(S, "finDimer.py"), "-i", fa, "-o", fa + ".findimer"], wd)
        out["stdout"]["finDimer"] = {"exit": rc[0]}
        for fn in sorted(os.listdir(wd)):
            if fn.endswith(".tfa"):
                continue
            raw = open(os.path.join(wd, fn), "rb").read()
            if fn.endswith("_json"):
                # key order of the side files depends on the hash seed: stored parsed
                out["files"][fn] = {"json": json.loads(raw)}
            else:
                out["files"][fn] = {"text": raw.decode().replace(wd, "@WD")}
    path = os.path.join(HERE, "chain.json.gz" if args.length == 18 else f"chain_k{args.length}.json.gz")
    with open(path, "wb") as f:
        f.write(gzip.compress(json.dumps(out, sort_keys=True).encode(), 9, mtime=0))
    print("wrote", path, os.path.getsize(path), "bytes;", {k: len(v.get("text", "")) for k, v in out["files"].items() if "text" in v})


if __name__ == "__main__":
    main()
