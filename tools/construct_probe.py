#!/usr/bin/env python3
"""The constructor (FASTA parse + mp_load_msa_fasta) and run() of NN_degenerate, repeated in one process on ONE kept context (as bench.py's
`pipeline` block and a --batch worker do), with and without a pause between a run() and the next constructor (GPU box): is a slow
constructor (0.15 s instead of 0.05 s at 10^6 rows) the pool's CPU quota catching up with the process after run()'s 96 threads?
usage: tools/construct_probe.py ROWS PAUSE_S"""
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multiprime_amd._abi import Library, prefer_staged_copies  # noqa: E402
prefer_staged_copies()
from multiprime_amd.core import NN_degenerate  # noqa: E402
from multiprime_amd.synth import synth_block, to_fasta  # noqa: E402


def throttled():
    try:
        with open("/sys/fs/cgroup/cpu.stat") as f:
            d = dict(line.split() for line in f)
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except OSError:
        return None


rows_n, pause = int(sys.argv[1]), float(sys.argv[2])
lib = Library()
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
    fa = os.path.join(td, "syn.fa")
    with open(fa, "wb") as f:
        f.write(to_fasta(synth_block(0, rows_n, 1000, 20250303)))
    ctx = None
    for rep in range(8):
        th0 = throttled()
        t0 = time.perf_counter()
        app = NN_degenerate(seq_file=fa, primer_length=18, coverage=0.8, number_of_dege_bases=4, score_of_dege_bases=10, raw_entropy_threshold=3.6,
                            product_len=150, position="2,3,-1", variation=1, distance=4, GC="0.2,0.7", nproc=1, outfile=os.path.join(td, "o.tsv"),
                            write_json=False, keep_bitsets=True, library=lib, device=0, context=ctx)
        ctx = app.ctx
        t1 = time.perf_counter()
        th1 = throttled()
        app.run()
        t2 = time.perf_counter()
        th2 = throttled()
        print(json.dumps({"rep": rep, "pause_s": pause, "construct_ms": round((t1 - t0) * 1e3, 1), "run_ms": round((t2 - t1) * 1e3, 1),
                          "parse_ms": round(app.stats.get("parse_s", 0) * 1e3, 1), "load_ms": round(app.stats.get("load_s", 0) * 1e3, 1),
                          "throttled_during_construct": None if th0 is None else [th1[0] - th0[0], th1[1] - th0[1]],
                          "throttled_during_run": None if th0 is None else [th2[0] - th1[0], th2[1] - th1[1]]}), flush=True)
        del app
        time.sleep(pause)
    ctx.close()
