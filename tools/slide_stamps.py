#!/usr/bin/env python3
"""Where a workgroup of eval_slide_kernel spends its life (GPU box): runs one synchronous launch of the bench workload with
MP_EXPERIMENT_STAMPS (thread 0 of every workgroup stores the 100 MHz wall clock at its phase borders) and prints, in microseconds
relative to the first workgroup's start: when workgroups start, and the median / max duration of every phase.
    python tools/slide_stamps.py --rows 131072"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=131072)
ap.add_argument("--slices", type=int, default=0, help="row slices of the launch (wc_pad): rows / 32 / 512 at 2 words per lane, padded to 8")
a = ap.parse_args()
with tempfile.TemporaryDirectory() as td:
    f = os.path.join(td, "stamps.bin")
    env = dict(os.environ, MP_EXPERIMENT_STAMPS=f, MP_EVAL_SLIDE="1")
    subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--rows", str(a.rows), "--steps", "3", "--warmup", "2", "--no-cpu", "--no-variants",
                    "--no-pipeline", "--no-shard"], env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    raw = np.fromfile(f, dtype=np.uint64)
n_slide, n_patch = int(raw[0]), int(raw[1])
st = raw[2:].reshape(-1, 8).astype(np.float64) / 100.0          # 100 MHz -> us
live = st[:, 0] > 0
t0 = st[live, 0].min()
st = np.where(st > 0, st - t0, np.nan)
s = st[:n_slide]
s = s[~np.isnan(s[:, 6])]                                       # workgroups that ran a band (padding slices return early)
p = st[n_slide:]
p = p[~np.isnan(p[:, 7])]
print(f"rows {a.rows}: {len(s)} sliding workgroups, {len(p)} patch workgroups; times in us after the first workgroup's start")
print("sliding workgroups start: median %.2f  max %.2f" % (np.median(s[:, 0]), s[:, 0].max()))
names = ["entry -> table zeroed, band record read", "-> records / iteration words requested", "-> warm-up columns in", "-> band done (items)", "-> workgroup barrier",
         "-> flush (global atomics)"]
for i, nm in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print("  %-48s median %6.2f  max %6.2f" % (nm, np.median(d), np.nanmax(d)))
print("sliding workgroups end:   median %.2f  max %.2f" % (np.median(s[:, 6]), s[:, 6].max()))
# who is slow?  durations by band (blockIdx // slices), by slice, by XCD (blockIdx % 8)
dur = st[:n_slide, 6] - st[:n_slide, 0]
idx = np.arange(n_slide)
ok = ~np.isnan(dur)
for name, key in (("XCD (blockIdx % 8)", idx % 8), ("band", idx // a.slices if a.slices else idx * 0), ("slice", idx % a.slices if a.slices else idx * 0)):
    ks = np.unique(key[ok])
    med = [np.median(dur[ok & (key == k)]) for k in ks]
    print("  duration by %-20s min-of-medians %.1f  max-of-medians %.1f   %s" % (name, min(med), max(med),
          " ".join("%d:%.0f" % (k, m) for k, m in zip(ks[:24], med[:24]))))
if len(p):
    print("patch workgroups start:   median %.2f  min %.2f  max %.2f;  duration median %.2f max %.2f;  last end %.2f" % (
        np.median(p[:, 0]), p[:, 0].min(), p[:, 0].max(), np.median(p[:, 7] - p[:, 0]), (p[:, 7] - p[:, 0]).max(), p[:, 7].max()))
