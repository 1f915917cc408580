#!/usr/bin/env python3
"""cProfile of the drop-in core step on one committed fixture (second run in the process), top of the cumulative list.
usage: python tools/profile_fixture.py [fixture]"""
import cProfile
import gzip
import json
import os
import pstats
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multiprime_amd._abi import prefer_staged_copies  # noqa: E402
prefer_staged_copies()            # a program's own decision, before the HIP runtime starts (multiprime_amd/_abi.py)
from multiprime_amd.core import NN_degenerate  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cluster0_v2"
G = os.path.join(REPO, "tests", "golden")
meta = json.loads(gzip.open(os.path.join(G, name + ".trace.json.gz")).read())["meta"]
fl = meta["flags"]
with tempfile.TemporaryDirectory() as td:
    inp = os.path.join(td, "in.fa")
    open(inp, "wb").write(gzip.open(os.path.join(G, "inputs", meta["input"] + ".gz")).read())

    def go():
        app = NN_degenerate(seq_file=inp, primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"], score_of_dege_bases=fl["d"],
                            raw_entropy_threshold=fl["e"], product_len=fl["s"], position=fl["c"], variation=fl["v"], distance=fl["a"],
                            GC=fl["g"], nproc=1, outfile=os.path.join(td, "out.tsv"))
        app.run()
        return app
    go()
    pr = cProfile.Profile()
    pr.enable()
    app = go()
    pr.disable()
    print({k: round(v, 4) for k, v in app.stats.items() if isinstance(v, float)})
    pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
