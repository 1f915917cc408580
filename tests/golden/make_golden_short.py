#!/usr/bin/env python3
"""Pins what the REFERENCE does on an alignment holding a row with fewer than k residues (V20:683-687 leaves that row's
k-mer short): it dies with a ValueError inside Y_distance (V20:230) as soon as a window whose universe holds the short
k-mer reaches mis_primer_check.  Records input, exit status and the exception line as tests/golden/short_row.json.
Test infrastructure; run in the authoring container (needs /root/reference):  python tests/golden/make_golden_short.py"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
V20 = "/root/reference/scripts/multiPrime-core.py"
ENV = dict(os.environ, NPY_DISABLE_CPU_FEATURES="AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3",
           PYTHONHASHSEED="0")


def make_input():
    rng = np.random.default_rng(5)
    root = rng.choice(list("ACGT"), size=150)
    rows = []
    for _ in range(40):
        s = root.copy()
        m = rng.random(150) < 0.02
        s[m] = rng.choice(list("ACGT"), size=int(m.sum()))
        rows.append("".join(s))
    rows[7] = rows[7][:10]      # only 10 residues: every window past column 10 leaves this row fewer than k = 18
    return "".join(f">s{i}\n{r}\n" for i, r in enumerate(rows))


def main():
    text = make_input()
    flags = ["-l", "18", "-n", "4", "-d", "10", "-v", "1", "-e", "3.6", "-g", "0.2,0.7", "-s", "60", "-f", "0.8", "-c", "2,3,-1", "-p", "1"]
    with tempfile.TemporaryDirectory() as td:
        inp = os.path.join(td, "short.fa")
        open(inp, "w").write(text)
        r = subprocess.run([sys.executable, V20, "-i", inp, "-o", os.path.join(td, "o")] + flags, env=ENV, capture_output=True, text=True)
        out_exists = os.path.exists(os.path.join(td, "o"))
    last = [l for l in r.stderr.strip().splitlines() if l.strip()][-1]
    rec = {"input": text, "flags": flags, "reference_returncode": r.returncode, "reference_last_stderr_line": last,
           "reference_wrote_tsv": out_exists}
    json.dump(rec, open(os.path.join(HERE, "short_row.json"), "w"), indent=1)
    print({k: v for k, v in rec.items() if k != "input"})


if __name__ == "__main__":
    main()
