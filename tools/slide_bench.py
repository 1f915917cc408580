#!/usr/bin/env python3
"""Sliding evaluation kernel against the first-pass kernels on the bench workload: same candidates, counters compared one by one,
HIP-event kernel time per launch for a list of settings (MP_EVAL_SLIDE, MP_SLIDE_GW, MP_SLIDE_BAND are read at upload time).

  python tools/slide_bench.py --rows 1048576 --set "slide=0" --set "slide=1,gw=2,band=16" ...
"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1048576)
    ap.add_argument("--cols", type=int, default=1000)
    ap.add_argument("--k", type=int, default=18)
    ap.add_argument("--v", type=int, default=1)
    ap.add_argument("--cands", type=int, default=8)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--set", action="append", default=[], help="slide=0|1[,gw=G][,band=B][,prog=0|1]")
    a = ap.parse_args()
    import torch
    import bench
    from multiprime_amd._abi import Library
    os.environ["MP_EVAL_TIMING_EVERY"] = "1"
    lib = Library()
    seed = 20250303
    rows = bench.synth_rows(0, a.rows, a.cols, seed)
    ctx = lib.context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.load_msa(rows.reshape(-1), np.arange(a.rows + 1, dtype=np.int64) * a.cols)
    k, v = a.k, a.v
    p0, W = 16, a.cols - 32 - k
    n_ex = ctx.build_windows(p0, W, k, v)
    bench.expand_exceptions(ctx, n_ex, k, v)
    from multiprime_amd.synth import synth_root
    root_codes = np.array([1, 2, 4, 8], np.uint8)[synth_root(a.cols, seed)]
    cw, codes = bench.make_candidates(root_codes, p0, W, k, a.cands, seed)
    sF = sum(1 << y for y in (2, 3, k) if 0 <= y < k)
    sR = sum(1 << y for y in (2, k - 3, k - 2) if 0 <= y < k)
    dev = torch.device("cuda", 0)
    out = torch.zeros((len(cw), 3), dtype=torch.int64, device=dev)
    ref = None
    for spec in a.set or ["slide=0", "slide=1"]:
        kv = dict(x.split("=") for x in spec.split(","))
        for key in ("MP_EVAL_SLIDE", "MP_SLIDE_GW", "MP_SLIDE_BAND", "MP_EVAL_PROG"):
            os.environ.pop(key, None)
        if "slide" in kv:
            os.environ["MP_EVAL_SLIDE"] = kv["slide"]
        if "gw" in kv:
            os.environ["MP_SLIDE_GW"] = kv["gw"]
        if "band" in kv:
            os.environ["MP_SLIDE_BAND"] = kv["band"]
        if "prog" in kv:
            os.environ["MP_EVAL_PROG"] = kv["prog"]
        ctx.eval_upload(cw, codes, sF, sR)
        t = bench.time_launches(ctx, torch, out.data_ptr(), a.launches, 3)
        got = out.cpu().numpy().copy()
        if ref is None:
            ref = got
        same = bool(np.array_equal(got, ref))
        print(json.dumps({"set": spec, "rows": a.rows, "kernel_ms_mean": round(t["mean_ms"], 5), "kernel_ms_median": round(t["median_ms"], 5),
                          "kernel_ms_max": round(t["max_ms"], 5), "equal_to_first": same, "checksum": got.sum(axis=0).tolist(),
                          "different_candidates": int((got != ref).any(axis=1).sum())}), flush=True)


if __name__ == "__main__":
    main()
