#!/usr/bin/env python3
"""Golden vectors for SURVEY §8f-1 (the pairing stage, scripts/get_multiPrime.py): the unmodified
reference core (multiPrime-core.py) is run on a fixture, then the unmodified reference pairing
script on its three output files with several flag sets; the three files it writes and a digest of
its stdout are stored.  Usage: python tests/golden/make_golden_pairing.py"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
ADAPT = "TCTTTCCCTACACGACGCTCTTCCGATCT,TGGAGTTCAGACGTGTGCTCTTCCGATCT"
ENV = dict(os.environ, PYTHONHASHSEED="0",
           NPY_DISABLE_CPU_FEATURES="AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3")

# pairing flag sets (multiPrime.yaml:100-118 first)
FLAGS = {
    "yaml": ["-f", "0.7", "-s", "150,1200", "-g", "0.2,0.7", "-e", "4", "-d", "4", "-a", ADAPT, "-m", "0"],
    "default": [],
    "noadaptor_t2": ["-a", ",", "-e", "0", "-t", "2", "-s", "200,600", "-f", "0.8", "-m", "100"],
    "tight": ["-f", "0.99", "-s", "150,400", "-e", "2", "-d", "3"],
}
CORE = {   # fixture -> (input under tests/golden/inputs, core flags)
    "cluster0_v1": ("Cluster_0_20727.tmsa", ["-n", "4", "-d", "10", "-v", "1", "-c", "2,3,-1", "-g", "0.2,0.7", "-s", "150", "-l", "18",
                                             "-e", "3.6", "-f", "0.7"]),
    "msa1000_k18_d64": ("1000_fasta.msa", ["-l", "18", "-d", "64", "-v", "1", "-n", "4", "-f", "0.8", "-c", "2,3,-1", "-e", "3.6",
                                           "-g", "0.2,0.7", "-s", "150"]),
    "ivc_v1": ("IV_C.msa", ["-v", "1"]),
}


def run_fixture(name):
    inp_name, core_flags = CORE[name]
    res = {}
    with tempfile.TemporaryDirectory() as td:
        inp = os.path.join(td, inp_name)
        raw = gzip.open(os.path.join(HERE, "inputs", inp_name + ".gz")).read()
        open(inp, "wb").write(raw)
        core_out = os.path.join(td, name + ".top.primer.out")
        subprocess.check_call([sys.executable, os.path.join(REF, "scripts", "multiPrime-core.py"), "-i", inp, "-o", core_out,
                               "-p", "1"] + core_flags, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=ENV)
        n_seq = raw.count(b">")
        ref = os.path.join(td, "ref.tfa")                   # only its line count is used (get_number)
        with open(ref, "w") as f:
            for i in range(n_seq):
                f.write(f">s{i}\nACGT\n")
        for fname, fl in FLAGS.items():
            od = os.path.join(td, fname)
            os.makedirs(od)
            out = os.path.join(od, name + ".candidate.primers.txt")
            p = subprocess.run([sys.executable, os.path.join(REF, "scripts", "get_multiPrime.py"), "-i", core_out, "-r", ref,
                                "-o", out, "-p", "1"] + fl, capture_output=True, text=True, env=ENV)
            lines = [l for l in p.stdout.splitlines() if not l.startswith("INFO ")]
            r = {"returncode": p.returncode, "n_seq": n_seq, "stdout_lines": len(lines),
                 "stdout_sha256": hashlib.sha256("\n".join(lines).encode()).hexdigest(),
                 "stdout_head": lines[:3], "n_dimer_msgs": sum(l.startswith("Dimer detection") for l in lines)}
            for ext, path in (("txt", out), ("xls", out.strip(".txt") + ".xls"), ("fa", out.strip(".txt") + ".fa")):
                txt = open(path).read() if os.path.exists(path) else None
                if txt is not None:
                    txt = txt.replace(out, "<OUT>")
                r[ext] = txt
            r["n_pairs"] = (r["xls"].count("\n") - 1) if r["xls"] else 0
            res[fname] = r
            print(name, fname, "rc", p.returncode, "pairs", r["n_pairs"], "dimer msgs", r["n_dimer_msgs"], flush=True)
    return name, res


def main():
    with ThreadPoolExecutor(3) as ex:
        g = dict(ex.map(run_fixture, list(CORE)))
    raw = json.dumps({"flags": FLAGS, "core": {k: v[1] for k, v in CORE.items()}, "results": g}, sort_keys=True).encode()
    open(os.path.join(HERE, "pairing.json.gz"), "wb").write(gzip.compress(raw, 9, mtime=0))
    print("written", len(raw), "bytes raw")


if __name__ == "__main__":
    main()
