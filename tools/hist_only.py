#!/usr/bin/env python3
"""The histogram stage alone (mp_load_msa, mp_build_windows, mp_window_unique x 3) on the bench alignment: for kernel traces and
experiment builds whose tables are not meant to be right.  usage: tools/hist_only.py ROWS K"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402,F401
from multiprime_amd._abi import Library  # noqa: E402
from multiprime_amd.synth import synth_block  # noqa: E402

n, k = int(sys.argv[1]), int(sys.argv[2])
L = 1000
rows = np.concatenate([synth_block(r0, min(32768, n - r0), L, 20250303) for r0 in range(0, n, 32768)])
ctx = Library().context(0)
ctx.load_msa(rows.reshape(-1), np.arange(n + 1, dtype=np.int64) * L)
ctx.build_windows(0, L - k, k, 1)
ctx.set_entropy_gate(3.6)
for _ in range(3):
    print(ctx.window_unique_device())
ctx.close()
