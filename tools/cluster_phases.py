#!/usr/bin/env python3
"""Where a 500-sequence cluster's core step spends its time (GPU box): phase sums over C clusters in one process, one at a time."""
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multiprime_amd._abi import prefer_staged_copies  # noqa: E402
prefer_staged_copies()            # a program's own decision, before the HIP runtime starts (multiprime_amd/_abi.py)
from multiprime_amd.core import NN_degenerate  # noqa: E402
from multiprime_amd.synth import synth_block, to_fasta  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(20250303)
cols = rng.integers(800, 2000, size=C)
tot = {}
with tempfile.TemporaryDirectory() as wd:
    fas = []
    for i in range(C):
        fa = os.path.join(wd, f"c{i}.fa")
        open(fa, "wb").write(to_fasta(synth_block(0, 500, int(cols[i]), 20250303 + 1000 * (i + 1))))
        fas.append(fa)
    for rep in range(2):
        tot = {}
        t0 = time.time()
        for i, fa in enumerate(fas):
            t1 = time.time()
            app = NN_degenerate(seq_file=fa, primer_length=18, coverage=0.7, number_of_dege_bases=4, score_of_dege_bases=10, raw_entropy_threshold=3.6,
                                product_len=150, position="2,3,-1", variation=1, distance=4, GC="0.2,0.7", nproc=1, outfile=os.path.join(wd, f"o{i}.tsv"))
            t2 = time.time()
            app.run()
            t3 = time.time()
            app.ctx.close()
            t4 = time.time()
            tot["ctor"] = tot.get("ctor", 0) + t2 - t1
            tot["run"] = tot.get("run", 0) + t3 - t2
            tot["close"] = tot.get("close", 0) + t4 - t3
            for k, v in app.stats.items():
                if isinstance(v, float):
                    tot["s." + k] = tot.get("s." + k, 0) + v
        wall = time.time() - t0
    print(json.dumps({"clusters": C, "wall_s": round(wall, 3), "ms_per_cluster": round(1e3 * wall / C, 2),
                      "phase_ms_per_cluster": {k: round(1e3 * v / C, 2) for k, v in sorted(tot.items())}}))
