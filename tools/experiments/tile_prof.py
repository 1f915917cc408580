#!/usr/bin/env python3
"""Phase times of one eval_tile_kernel launch from the per-wave shader-clock stamps the library writes with MP_TILE_PROF=<file>
(debug aid, GPU box):  python tools/tile_prof.py [--rows N]  -> medians over workgroups, in cycles of the stamp counter."""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=131072)
    ap.add_argument("--tile", default="4")
    a = ap.parse_args()
    import torch
    from multiprime_amd._abi import Library
    from multiprime_amd.synth import synth_block, synth_root
    k, v, C, L = 18, 1, 8, 1000
    ctx = Library().context(0)
    rows = synth_block(0, a.rows, L, 20250303)
    ctx.load_msa(rows.reshape(-1), np.arange(a.rows + 1, dtype=np.int64) * L)
    p0, W = 16, L - 32 - k
    n_ex = ctx.build_windows(p0, W, k, v)
    bench.expand_exceptions(ctx, n_ex, k, v)
    root_codes = np.array([1, 2, 4, 8], np.uint8)[synth_root(L, 20250303)]
    cw, codes = bench.make_candidates(root_codes, p0, W, k, C, 20250303)
    sF = sum(1 << y for y in {2, 3, k} if 0 <= y < k)
    sR = sum(1 << y for y in {2, k - 3, k - 2} if 0 <= y < k)
    os.environ["MP_EVAL_TILE"] = a.tile
    ctx.eval_upload(cw, codes, sF, sR)
    out = torch.zeros((len(cw), 3), dtype=torch.int64, device="cuda")
    for _ in range(5):
        ctx.eval_launch(out.data_ptr())
    torch.cuda.synchronize()
    path = "/tmp/tile_prof.bin"
    os.environ["MP_TILE_PROF"] = path
    ctx.eval_launch(out.data_ptr())
    torch.cuda.synchronize()
    del os.environ["MP_TILE_PROF"]
    raw = open(path, "rb").read()
    grid, waves, slots, n_slices = np.frombuffer(raw[:16], np.int32)
    t = np.frombuffer(raw[16:], np.uint64).reshape(grid, waves, slots).astype(np.int64)
    t0 = t[:, :, 0].min()
    n = int((t[0, 0] > 0).sum())
    print(f"grid={grid} waves={waves} stamps per wave={n} slices={n_slices}")
    print(f"kernel span (first stamp of any wave -> last stamp of any wave): {t[:, :, :n].max() - t0} ticks")
    start = t[:, :, 0] - t0
    print(f"wave start: median {np.median(start):.0f}  max {start.max()}")
    names = ["arith done->", "barrier A ->", "stores    ->", "barrier B ->"]
    rounds = (n - 2) // 4
    for r in range(rounds):
        b = 1 + 4 * r
        seg = [np.median(t[:, :, b] - t[:, :, b - 1])] + [np.median(t[:, :, b + i + 1] - t[:, :, b + i]) for i in range(3)]
        worst = [(t[:, :, b] - t[:, :, b - 1]).max()] + [(t[:, :, b + i + 1] - t[:, :, b + i]).max() for i in range(3)]
        print(f"round {r}: " + "  ".join(f"{nm}{s:8.0f} (max {w})" for nm, s, w in zip(["arith(prev) ", "barrierA ", "stores ", "barrierB "], seg, worst)))
    print(f"last arithmetic: median {np.median(t[:, :, n - 1] - t[:, :, n - 2]):.0f} max {(t[:, :, n - 1] - t[:, :, n - 2]).max()}")
    print(f"end of wave: median {np.median(t[:, :, n - 1] - t0):.0f} max {(t[:, :, n - 1] - t0).max()}")


if __name__ == "__main__":
    main()
