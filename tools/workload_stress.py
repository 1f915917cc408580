#!/usr/bin/env python3
"""The place where two of 57 `bench.py` runs of round 6 died (the first seconds of the `k_sweep` block): bench's Workload — a fresh context,
mp_load_msa of the 10^6-row alignment, mp_build_windows, the exception list, the extra rows, two evaluations — created and closed over
and over on the same rows, k alternating (GPU box).  MPRIME_LIBRARY / MP_HOST_LIB select the build.  Exit status of the child = the finding."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import argparse, os, sys, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
from multiprime_amd._abi import Library, prefer_staged_copies
prefer_staged_copies()
import numpy as np
import torch
from bench_common import Workload, synth_rows, time_launches
lib = Library()
rows_n, n_iter = int(sys.argv[1]), int(sys.argv[2])
rows = synth_rows(0, rows_n, 1000, 20250303)
t0 = time.time()
for i in range(n_iter):
    a = argparse.Namespace(k=(18, 20, 22, 36)[i %% 4], v=1, cands=8, cols=1000, seed=20250303)
    w = Workload(lib, 0, torch, 0, rows_n, a, rows=rows, win_part=(i %% 3, 3) if i %% 5 == 4 else (0, 1))
    buf = torch.zeros((w.n_cand, 3), dtype=torch.int64, device="cuda")
    time_launches(w.ctx, torch, buf.data_ptr(), 3, 1)
    w.ctx.close(); w.ctx = None
    del buf
    if i %% 7 == 0:
        torch.cuda.empty_cache()
print("done %%.1f s" %% (time.time() - t0))
''' % (REPO, REPO)

if __name__ == "__main__":
    rows_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", CHILD, str(rows_n), str(n)], capture_output=True, text=True, timeout=2400,
                       env=dict(os.environ, MP_DEBUG_TERMINATE="1"))
    print(json.dumps({"library": os.environ.get("MPRIME_LIBRARY", "product"), "rows": rows_n, "workloads": n, "exit": r.returncode,
                      "stdout": r.stdout.strip()[-60:], "stderr_tail": r.stderr[-2500:] if r.returncode else ""}), flush=True)
