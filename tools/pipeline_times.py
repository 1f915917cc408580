#!/usr/bin/env python3
"""End-to-end wall time of the drop-in core step on the committed fixtures, on the GPU box
(whole program: parse, pack, windows, histograms, host control flow, evaluation, filters, files).
The reference's wall time on the same input/flags (1 core, authoring container) is in the
golden trace's meta."""
import gzip
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multiprime_amd._abi import prefer_staged_copies  # noqa: E402
prefer_staged_copies()            # a program's own decision, before the HIP runtime starts (multiprime_amd/_abi.py)
from multiprime_amd.core import NN_degenerate  # noqa: E402

G = os.path.join(REPO, "tests", "golden")
for name in sys.argv[1:] or ["msa1000_k18_d64", "msa1000_k20_d64", "msa1000_k22_d64", "cluster0_v1", "cluster0_v2", "ivc_v1", "testfa"]:
    meta = json.loads(gzip.open(os.path.join(G, name + ".trace.json.gz")).read())["meta"]
    fl = meta["flags"]
    with tempfile.TemporaryDirectory() as td:
        inp = os.path.join(td, "in.fa")
        open(inp, "wb").write(gzip.open(os.path.join(G, "inputs", meta["input"] + ".gz")).read())
        best = None
        for rep in range(2):
            t0 = time.time()
            app = NN_degenerate(seq_file=inp, primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"],
                                score_of_dege_bases=fl["d"], raw_entropy_threshold=fl["e"], product_len=fl["s"],
                                position=fl["c"], variation=fl["v"], distance=fl["a"], GC=fl["g"], nproc=1,
                                outfile=os.path.join(td, "out.tsv"))
            app.run()
            dt = time.time() - t0
            best = dt if best is None else min(best, dt)
        same = open(os.path.join(td, "out.tsv"), "rb").read() == open(os.path.join(G, name + ".tsv"), "rb").read()
        st = {k: round(v, 3) for k, v in app.stats.items() if isinstance(v, float)}
        # the drop-in script as a user runs it: a fresh process (interpreter, numpy, HIP runtime start-up included)
        import subprocess
        t0 = time.time()
        subprocess.check_call([sys.executable, os.path.join(REPO, "scripts", "multiPrime-core.py"), "-i", inp, "-o", os.path.join(td, "cli.tsv"),
                               "-l", str(fl["l"]), "-n", str(fl["n"]), "-d", str(fl["d"]), "-v", str(fl["v"]), "-e", str(fl["e"]), "-g", fl["g"],
                               "-s", str(fl["s"]), "-f", str(fl["f"]), "-c", fl["c"], "-a", str(fl["a"]), "-p", "1"], stdout=subprocess.DEVNULL)
        cli = time.time() - t0
        same = same and open(os.path.join(td, "cli.tsv"), "rb").read() == open(os.path.join(G, name + ".tsv"), "rb").read()
        print(json.dumps({"fixture": name, "n_seq": meta["n_seq"], "windows": meta["n_windows"], "wall_s": round(best, 3),
                          "cli_process_s": round(cli, 2),
                          "reference_wall_s": meta["reference_wall_s"], "speedup": round(meta["reference_wall_s"] / best, 1),
                          "tsv_identical": same, "n_candidates": app.stats.get("n_candidates"), "phases": st}), flush=True)
