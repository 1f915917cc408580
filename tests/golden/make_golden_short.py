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


def base_rows(seed=5, n=40, L=150):
    rng = np.random.default_rng(seed)
    root = rng.choice(list("ACGT"), size=L)
    rows = []
    for _ in range(n):
        s = root.copy()
        m = rng.random(L) < 0.02
        s[m] = rng.choice(list("ACGT"), size=int(m.sum()))
        rows.append("".join(s))
    return rows


def fasta(rows):
    return "".join(f">s{i}\n{r}\n" for i, r in enumerate(rows))


def make_input():
    rows = base_rows()
    rows[7] = rows[7][:10]      # only 10 residues: every window past column 10 leaves this row fewer than k = 18
    return fasta(rows)


def more_inputs():
    """Other shapes around the same input class (round 3): a row with fewer than k residues in its RECORD makes the reference fail; a
    row with fewer than k residues between gaps does not (its slices keep their length, V20:668-687) and must give the same TSV."""
    out = {}
    rows = base_rows(6)
    rows[11] = ""                                                   # an empty record
    out["empty_record"] = (fasta(rows), 11)
    rows = base_rows(7)
    rows[3] = "-" * 60 + rows[3][60:77] + "-" * (150 - 77)          # 17 residues in the middle of the alignment, gaps around them
    out["seventeen_residues_between_gaps"] = (fasta(rows), 3)
    rows = base_rows(8)
    for r in (37, 38, 39):
        rows[r] = rows[r][:12]                                      # the last three records break off after 12 residues
    out["ragged_tail"] = (fasta(rows), 37)
    return out


def run_reference(text, flags):
    with tempfile.TemporaryDirectory() as td:
        inp = os.path.join(td, "short.fa")
        open(inp, "w").write(text)
        r = subprocess.run([sys.executable, V20, "-i", inp, "-o", os.path.join(td, "o")] + flags, env=ENV, capture_output=True, text=True)
        out_exists = os.path.exists(os.path.join(td, "o"))
        tsv = open(os.path.join(td, "o")).read() if out_exists else None
    last = [l for l in r.stderr.strip().splitlines() if l.strip()][-1]
    rec = {"input": text, "flags": flags, "reference_returncode": r.returncode, "reference_last_stderr_line": last,
           "reference_wrote_tsv": out_exists}
    if tsv is not None and r.returncode == 0:
        rec["reference_tsv"] = tsv                 # the reference got through: its output is the golden
    return rec


def main():
    flags = ["-l", "18", "-n", "4", "-d", "10", "-v", "1", "-e", "3.6", "-g", "0.2,0.7", "-s", "60", "-f", "0.8", "-c", "2,3,-1", "-p", "1"]
    rec = run_reference(make_input(), flags)
    json.dump(rec, open(os.path.join(HERE, "short_row.json"), "w"), indent=1)
    print({k: v for k, v in rec.items() if k not in ("input", "reference_tsv")})
    more = {}
    for name, (text, row) in more_inputs().items():
        more[name] = dict(run_reference(text, flags), first_short_row=row)
        print(name, {k: v for k, v in more[name].items() if k not in ("input", "reference_tsv")})
    json.dump(more, open(os.path.join(HERE, "short_rows_more.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
