#!/usr/bin/env python3
"""cProfile of NN_degenerate(...).run() on a deep synthetic alignment (GPU box): where the HOST time of the core step goes."""
import cProfile
import os
import pstats
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multiprime_amd._abi import prefer_staged_copies  # noqa: E402
prefer_staged_copies()
from multiprime_amd.core import NN_degenerate  # noqa: E402
from multiprime_amd.synth import synth_block, to_fasta  # noqa: E402

rows_n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
k_len = int(sys.argv[2]) if len(sys.argv) > 2 else 18
with tempfile.TemporaryDirectory() as td:
    fa = os.path.join(td, "syn.fa")
    open(fa, "wb").write(to_fasta(synth_block(0, rows_n, 1000, 20250303)))

    def go():
        app = NN_degenerate(seq_file=fa, primer_length=k_len, coverage=0.8, number_of_dege_bases=4, score_of_dege_bases=10, raw_entropy_threshold=3.6,
                            product_len=150, position="2,3,-1", variation=1, distance=4, GC="0.2,0.7", nproc=1, outfile=os.path.join(td, "o.tsv"),
                            write_json=False, keep_bitsets=True)
        app.run()
        return app

    go().ctx.close()
    pr = cProfile.Profile()
    pr.enable()
    app = go()
    pr.disable()
    print({k: round(v * 1e3, 2) for k, v in app.stats.items() if isinstance(v, float)})
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
