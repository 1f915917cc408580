#!/usr/bin/env python3
"""Whole drop-in core step on a large synthetic alignment (default 131072 x 1000, the bench shard)
on the GPU box: writes the FASTA, runs scripts/multiPrime-core.py's code path with --no-json and
prints the per-phase timings.  There is no reference output at this size (the Python reference would
need ~1 h); the run checks the internal host/device perfect-coverage invariant on every candidate."""
import argparse
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
from multiprime_amd.synth import synth_block, to_fasta  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=131072)
ap.add_argument("--cols", type=int, default=1000)
ap.add_argument("--k", type=int, default=18)
ap.add_argument("--file", action="store_true", help="hand the coverage bitsets to the pairing stage through <out>.coverage_bitsets.npz (two "
                "processes' worth of work) instead of leaving them on the device")
a = ap.parse_args()
with tempfile.TemporaryDirectory() as td:
    t0 = time.time()
    rows = synth_block(0, a.rows, a.cols, 20250303)
    fa = os.path.join(td, "syn.fa")
    with open(fa, "wb") as f:
        f.write(to_fasta(rows))
    t_gen = time.time() - t0
    t0 = time.time()
    app = NN_degenerate(seq_file=fa, primer_length=a.k, coverage=0.8, number_of_dege_bases=4, score_of_dege_bases=10,
                        raw_entropy_threshold=3.6, product_len=150, position="2,3,-1", variation=1, distance=4, GC="0.2,0.7",
                        nproc=1, outfile=os.path.join(td, "out.tsv"), write_json=False, write_bitsets=a.file, keep_bitsets=not a.file)
    app.run()
    if os.environ.get("MP_REPEAT_UNIQUE"):                      # debugging: the histogram read-back again, in the same process
        for _ in range(3):
            t0 = time.time()
            app.ctx.window_unique(want_labels=False, sort=False)
            print("window_unique again: %.4f s" % (time.time() - t0), file=sys.stderr)
    wall = time.time() - t0
    n_out = sum(1 for _ in open(os.path.join(td, "out.tsv"))) - 1
    # pairing stage straight from the coverage bitsets (no JSON exists at this depth)
    import contextlib
    import io
    from multiprime_amd.pairing import Primers_filter
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        pf = Primers_filter(ref_file=fa, primer_file=os.path.join(td, "out.tsv"), outfile=os.path.join(td, "syn.candidate.primers.txt"),
                            adaptor="TCTTTCCCTACACGACGCTCTTCCGATCT,TGGAGTTCAGACGTGTGCTCTTCCGATCT", rep_seq_number=0, distance=4,
                            size="150,600", position=4, fraction=0.7, diff_Tm=4, core=None if a.file else app)
        pf.run()
    pair_wall = time.time() - t0
    n_pairs = sum(1 for _ in open(os.path.join(td, "syn.candidate.primers.xls"))) - 1
    print(json.dumps({"pairing_wall_s": round(pair_wall, 2), "pairs": n_pairs, "hand_off": "file" if a.file else "device-resident bitsets",
                      "bitset_file_bytes": os.path.getsize(os.path.join(td, "out.tsv.coverage_bitsets.npz")) if a.file else 0,
                      "pairing_stats": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in pf.stats.items()}}))
    print(json.dumps({"rows": a.rows, "cols": a.cols, "k": a.k, "generate_s": round(t_gen, 2), "wall_s": round(wall, 2),
                      "windows": app.n_windows, "windows_past_the_gates": app.stats.get("windows_planned"), "rows_out": n_out, "n_candidates": app.stats.get("n_candidates"),
                      "phases": {k: round(v, 3) for k, v in app.stats.items() if isinstance(v, float)},
                      "device_bytes": app.ctx.device_bytes()}))
