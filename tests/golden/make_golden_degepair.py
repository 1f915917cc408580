#!/usr/bin/env python3
"""Goldens for the get_degePrimer.py drop-in (multiprime_amd/degepair.py), produced by RUNNING the unmodified reference
script on the DEGEPRIME tables shipped under test_data/variation_effect/identity90/ with several flag sets.
Usage: python tests/golden/make_golden_degepair.py   (writes tests/golden/degepair.json.gz + gz copies of the inputs)"""
import gzip
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
T = os.path.join(REF, "test_data", "variation_effect", "identity90")
INPUTS = {"dege10": os.path.join(T, "degeneracy_10", "1000_fasta.dege.out"), "dege8": os.path.join(T, "degeneracy_8", "1000_fasta.dege.out")}
FLAGS = {
    "yaml": ["-f", "0.5", "-s", "150,1200", "-g", "0.2,0.7", "-e", "4", "-d", "4", "-a", "TCTTTCCCTACACGACGCTCTTCCGATCT,TGGAGTTCAGACGTGTGCTCTTCCGATCT", "-m", "500"],
    "default": [],
    "loose": ["-f", "0.05", "-s", "100,400", "-e", "0", "-d", "3", "-a", ",", "-m", "2000"],
    "tight": ["-f", "0.3", "-s", "200,300", "-e", "6", "-m", "100"],
}


def main():
    os.makedirs(os.path.join(HERE, "inputs"), exist_ok=True)
    g = {"flags": FLAGS, "results": {}}
    for name, path in INPUTS.items():
        if not os.path.exists(path):
            continue
        raw = open(path, "rb").read()
        open(os.path.join(HERE, "inputs", f"degeprime_{name}.out.gz"), "wb").write(gzip.compress(raw, 9, mtime=0))
        g["results"][name] = {}
        for fs, flags in FLAGS.items():
            with tempfile.TemporaryDirectory() as td:
                ref = os.path.join(td, "ref.fa")
                open(ref, "w").write("".join(f">s{i}\nACGT\n" for i in range(1000)))
                out = os.path.join(td, "cand.txt")
                p = subprocess.run([sys.executable, os.path.join(REF, "scripts", "get_degePrimer.py"), "-i", path, "-r", ref, "-o", out] + flags,
                                   capture_output=True, text=True)
                txt = open(out).read().replace(out, "<OUT>") if os.path.exists(out) else None
                lines = [l for l in p.stdout.splitlines() if not l.startswith("INFO ")]
                g["results"][name][fs] = {"returncode": p.returncode, "txt": txt, "stdout": lines}
                print(name, fs, p.returncode, (len(txt.split("\t")) - 2) // 5 if txt else None, lines[:2], p.stderr[-200:])
    open(os.path.join(HERE, "degepair.json.gz"), "wb").write(gzip.compress(json.dumps(g, sort_keys=True).encode(), 9, mtime=0))


if __name__ == "__main__":
    main()
