#!/usr/bin/env python3
"""A/B of the evaluation kernels compiled into libmprime_hip.so on the bench workload (same shard, windows and
candidates as bench.py); checks that every variant returns the counters of the first one.  Runs on the GPU box.

  --mode rows   MP_EVAL_VARIANT numbers of the row-per-lane kernel (profiles/r01_variants.txt)
  --mode bits   bN = symbol-table kernel (MP_EVAL_BITS=N), cN = nested-chain kernel shape (MP_EVAL_CHAIN=N)
                tN = LDS-tiled sweep with N row words per lane (MP_EVAL_TILE=N), tN:G with G workgroups aimed at
                e.g. --variants b1 b2 c3 c7 t4 t2   (profiles/r01_variants_v4.txt)
  --rows / --cands / --v   shard depth, candidates per window (a chain of that length), mismatch tolerance"""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=131072)
    ap.add_argument("--cols", type=int, default=1000)
    ap.add_argument("--k", type=int, default=18)
    ap.add_argument("--v", type=int, default=1)
    ap.add_argument("--cands", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--variants", nargs="*", default=["0", "1", "2", "3", "4"],
                    help="rows mode: MP_EVAL_VARIANT numbers; bits mode: bN = MP_EVAL_BITS=N (symbol-table kernel), "
                         "cN = nested-chain kernel shape MP_EVAL_CHAIN=N")
    ap.add_argument("--generic-v", action="store_true")
    ap.add_argument("--wins", type=int, default=0, help="evaluate only the first N windows (scaling experiments)")
    ap.add_argument("--mode", choices=["rows", "bits"], default="rows",
                    help="rows: MP_EVAL_VARIANT of the row-per-lane kernel; bits: MP_EVAL_BITS shapes of the bit-sliced kernel")
    a = ap.parse_args()
    import torch
    from multiprime_amd._abi import Library
    from multiprime_amd.synth import synth_block, synth_root
    k, v, C, L = a.k, a.v, a.cands, a.cols
    ctx = Library().context(0)
    rows = synth_block(0, a.rows, L, 20250303)
    ctx.load_msa(rows.reshape(-1), np.arange(a.rows + 1, dtype=np.int64) * L)
    p0, W = 16, L - 32 - k
    if a.wins:
        W = min(W, a.wins)
    n_ex = ctx.build_windows(p0, W, k, v)
    bench.expand_exceptions(ctx, n_ex, k, v)
    root_codes = np.array([1, 2, 4, 8], np.uint8)[synth_root(L, 20250303)]
    cw, codes = bench.make_candidates(root_codes, p0, W, k, C, 20250303)
    sF = sum(1 << y for y in {2, 3, k} if 0 <= y < k)
    sR = sum(1 << y for y in {2, k - 3, k - 2} if 0 <= y < k)
    alln = ctx.eval_candidates(np.arange(W, dtype=np.int32), np.full((W, k), 15, np.uint8), 0, 0)
    evals = int((alln[:, 0] + alln[:, 1]).sum()) * C
    ctx.eval_upload(cw, codes, sF, sR)
    out = torch.zeros((len(cw), 3), dtype=torch.int64, device="cuda")
    ref = None
    if a.generic_v:
        os.environ["MP_EVAL_GENERIC_V"] = "1"
    for var in a.variants:
        os.environ["MP_EVAL_MODE"] = a.mode
        if a.mode == "rows":
            os.environ["MP_EVAL_VARIANT"] = var
        elif var.startswith("t"):                        # LDS-tiled sweep, tN[:groups] = N row words per lane
            os.environ["MP_EVAL_BITS"] = "0"
            os.environ.pop("MP_EVAL_CHAIN", None)
            os.environ["MP_EVAL_TILE"] = var[1:].split(":")[0]
            if ":" in var:
                os.environ["MP_EVAL_TILE_GROUPS"] = var.split(":")[1]
            else:
                os.environ.pop("MP_EVAL_TILE_GROUPS", None)
            ctx.eval_upload(cw, codes, sF, sR)           # a new plan
        elif var[0] in "cpq":  # cN: eval_chain_kernel shape N; pN: the program-driven kernel, same shapes (+ 9); qN: the same with a wave per item
            os.environ["MP_EVAL_BITS"] = "0"
            os.environ["MP_EVAL_TILE"] = "0"
            os.environ["MP_EVAL_PROG"] = "1" if var[0] in "pq" else "0"
            os.environ["MP_EVAL_QUAD"] = "1" if var[0] == "q" else "0"
            os.environ["MP_EVAL_CHAIN"] = var[1:]
            ctx.eval_upload(cw, codes, sF, sR)           # the programs are written at upload time
        else:
            os.environ["MP_EVAL_BITS"] = var[1:]
        for _ in range(3):
            ctx.eval_launch(out.data_ptr())
        ctx.eval_timing(reset=True)
        for _ in range(a.steps):
            ctx.eval_launch(out.data_ptr())
        ms, n = ctx.eval_timing(reset=True)
        res = out.cpu().numpy()
        if ref is None:
            ref = res
        same = bool((res == ref).all())
        per = ms / n
        print(json.dumps({"mode": a.mode, "variant": var, "ms": round(per, 4), "evals_per_s": evals / per * 1e3, "identical": same,
                          "checksum": res.sum(axis=0).tolist(), "C": C, "rows": a.rows, "v": v, "generic_v": a.generic_v}), flush=True)


if __name__ == "__main__":
    main()
