#!/usr/bin/env python3
"""Phase times of hist_kernel's workgroups (debug aid, GPU box): MP_HIST_PROF stamps -> medians in shader-clock ticks."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from multiprime_amd._abi import Library  # noqa: E402

rows_n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
ctx = Library().context(0)
rows = bench.synth_rows(0, rows_n, 1000, 20250303)
ctx.load_msa(rows.reshape(-1), np.arange(rows_n + 1, dtype=np.int64) * 1000)
ctx.build_windows(16, 950, 18, 1)
ctx.window_unique(sort=False)
os.environ["MP_HIST_PROF"] = "/tmp/hist_prof.bin"
ctx.build_windows(16, 950, 18, 1)
ctx.window_unique(sort=False)
t = np.fromfile("/tmp/hist_prof.bin", np.uint64).reshape(-1, 8).astype(np.int64)
t = t[t[:, 0] > 0]
print("workgroups", len(t))
for name, a, b in (("init", 0, 1), ("row loop", 1, 2), ("patch rows + mid flush", 2, 3), ("final flush", 3, 4), ("total", 0, 4)):
    d = t[:, b] - t[:, a]
    print(f"{name:24s} median {np.median(d):9.0f}  p90 {np.percentile(d, 90):9.0f}  max {d.max():9.0f}")
print("LDS entries after the row loop: median", np.median(t[:, 6]), "max", t[:, 6].max(), "; at the final flush: median", np.median(t[:, 7]), "max", t[:, 7].max())
