#!/usr/bin/env python3
"""Fuzz corpus for the core step: small random alignments x random flag sets, each run through the
UNMODIFIED reference (multiPrime-core_V20.py, stable-argsort environment of SURVEY A-14); inputs,
flags, exit code, TSV and canonical digests of the two JSON files are stored in fuzz.json.gz.
Usage: python tests/golden/make_golden_fuzz.py [--n 48]"""
import argparse
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from multiprime_amd.synth import synth_block  # noqa: E402

REF = "/root/reference"
ENV = dict(os.environ, PYTHONHASHSEED="0",
           NPY_DISABLE_CPU_FEATURES="AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3")


def canon(obj):
    return hashlib.sha256(json.dumps(obj, sort_keys=True, separators=(",", ":")).encode()).hexdigest()


def canon_noncov(d):
    return {str(k): [{km: sorted(ids) for km, ids in sorted(side.items())} for side in v] for k, v in d.items()}


def canon_gap(d):
    return {str(k): {km: list(ids) for km, ids in sorted(v.items())} for k, v in d.items()}


def make_case(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(4, 70))
    L = int(rng.integers(70, 220))
    rows = synth_block(0, n, L, 500 + seed, p_sub=float(rng.choice([0.004, 0.01, 0.03])), p_var=float(rng.choice([0.1, 0.3])),
                       var_frac=float(rng.choice([0.02, 0.08])), p_gap=float(rng.choice([0.0, 0.01, 0.05])),
                       edge_frac=float(rng.choice([0.0, 0.2, 0.6])), edge_mean=float(rng.choice([4, 12])),
                       p_iupac=float(rng.choice([0.0, 1e-3, 1e-2])), block_rows=128)
    if rng.random() < 0.3:                       # a few junk / lower-case characters
        junk = np.frombuffer(b"nN*acgtRY", dtype=np.uint8)
        hit = rng.random(rows.shape) < 0.01
        rows = np.where(hit, junk[rng.integers(0, len(junk), rows.shape)], rows).astype(np.uint8)
    ragged = rng.random() < 0.25
    parts = []
    for i in range(n):
        s = rows[i].tobytes()
        if ragged and i % 3 == 1:
            s = s[: int(rng.integers(L // 2, L))]
        parts.append(b">q%03d\n" % i + s + b"\n")
    k = int(rng.integers(10, 25))
    flags = {"l": k, "v": int(rng.integers(0, 4)), "d": int(rng.choice([4, 10, 24, 64, 1000])), "n": int(rng.choice([2, 4, 18])),
             "f": float(rng.choice([0.5, 0.7, 0.8, 0.95, 1.0])), "e": float(rng.choice([2.0, 3.6, 5.0])),
             "c": str(rng.choice(["1,2,-1", "2,3,-1", "1,-1", "-2", "3", "0", "2,-3,5"])),
             "g": str(rng.choice(["0.2,0.7", "0.4,0.6", "0.0,1.0"])), "s": int(rng.choice([20, 30, 40, 60, 60, 300])),
             "a": int(rng.choice([3, 4, 6]))}
    if flags["v"] >= k:
        flags["v"] = 1
    return b"".join(parts), flags


def run_case(seed):
    fasta, fl = make_case(seed)
    with tempfile.TemporaryDirectory() as td:
        inp = os.path.join(td, "in.fa")
        open(inp, "wb").write(fasta)
        out = os.path.join(td, "out.tsv")
        cmd = [sys.executable, os.path.join(REF, "scripts", "multiPrime-core.py"), "-i", inp, "-o", out, "-p", "1"]
        for key, val in fl.items():
            cmd += ["-" + key, str(val)]
        p = subprocess.run(cmd, capture_output=True, text=True, env=ENV)
        rec = {"seed": seed, "flags": fl, "fasta": fasta.decode("latin-1"), "returncode": p.returncode,
               "stdout_head": [l for l in p.stdout.splitlines() if not l.startswith("INFO ")][:2],
               "crashed": "Traceback" in p.stderr}
        if p.returncode == 0 and os.path.exists(out):
            rec["tsv"] = open(out).read()
            rec["noncov_sha"] = canon(canon_noncov(json.load(open(out + ".non_coverage_seq_id_json"))))
            rec["gap_sha"] = canon(canon_gap(json.load(open(out + ".gap_seq_id_json"))))
        return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=48)
    a = ap.parse_args()
    with ThreadPoolExecutor(8) as ex:
        recs = list(ex.map(run_case, range(a.n)))
    for r in recs:
        print(r["seed"], "rc", r["returncode"], "crashed" if r["crashed"] else "", "rows", (r.get("tsv") or "").count("\n") - 1,
              {k: r["flags"][k] for k in ("l", "v", "d", "f", "c")})
    raw = json.dumps(recs, sort_keys=True).encode()
    open(os.path.join(HERE, "fuzz.json.gz"), "wb").write(gzip.compress(raw, 9, mtime=0))
    print("written", len(raw), "bytes raw;", sum(1 for r in recs if r["returncode"] == 0), "ok,",
          sum(1 for r in recs if r["crashed"]), "reference crashes")


if __name__ == "__main__":
    main()
