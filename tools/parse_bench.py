#!/usr/bin/env python3
"""Host-only: the native FASTA parser (csrc/fasta.cpp) on the bench alignment written to /dev/shm, under thread counts (MP_HOST_THREADS)
with the parser's own laps (MP_HOST_TRACE).  usage: tools/parse_bench.py [ROWS] [threads ...]"""
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
from multiprime_amd import host  # noqa: E402
from multiprime_amd.synth import synth_block, to_fasta  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
threads = [int(x) for x in sys.argv[2:]] or [0, 16, 32, 64, 128]
rows = np.concatenate([synth_block(r0, min(32768, n - r0), 1000, 20250303) for r0 in range(0, n, 32768)])
with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
    fa = os.path.join(td, "syn.fa")
    with open(fa, "wb") as f:
        f.write(to_fasta(rows))
    os.environ["MP_HOST_TRACE"] = "1"
    for t in threads:
        if t:
            os.environ["MP_HOST_THREADS"] = str(t)
        else:
            os.environ.pop("MP_HOST_THREADS", None)
        for rep in range(2):
            t0 = time.perf_counter()
            x = host.Fasta(fa)
            dt = time.perf_counter() - t0
            print(f"threads {t or 'default'} rep {rep}: parse {dt * 1e3:.1f} ms, rows {x.n_rows}", file=sys.stderr)
            x.close()
