#!/usr/bin/env python3
"""Benchmark of the hot path: batched candidate x sequence coverage evaluation
(mp_eval_launch -> eval_kernel) on a synthetic alignment shard per GPU (SURVEY §8d input 4 /
BASELINE.json configs[3]: 1M x 1 kb sequences sharded over 8 GPUs = 131072 rows per GPU).

One step = one pass of the evaluation over every window of the shard with C candidates per
window, plus (N > 1) the RCCL all-reduce of the per-candidate coverage counters.  Inputs are
resident in HBM (window words built once, untimed); weak scaling: the shard per GPU is fixed.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` (algorithmic
bytes 3k/8 per evaluation over the HIP-event kernel time, vs 8 TB/s) and `cpu_baseline` (the
plain-C oracle on one host core, bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable


KERNELS = {
    "chain": "eval_chain_kernel (bit-sliced one-hot column planes, nested refinement chains; patch rows ride in the same launch)",
    "table": "eval_bits_kernel (bit-sliced one-hot column planes, symbol table per position; patch rows ride in the same launch)",
    "rows": "eval_kernel (row-per-lane window words)",
}


def eval_mode():
    """Which evaluation kernels the library's environment switches select (defaults: the nested-chain kernel)."""
    if os.environ.get("MP_EVAL_MODE") == "rows":
        return "rows"
    if os.environ.get("MP_EVAL_BITS", "0") in ("1", "2") or os.environ.get("MP_EVAL_GROUP") == "plain":
        return "table"
    return "chain"


def make_candidates(root_codes, p0, W, k, C, seed):
    """C candidates per window: the root k-mer of the window, then progressively more degenerate
    versions (one more random base at one more random position each), seeded per window — the
    shape of a refinement chain (SURVEY §8d micro-benchmark)."""
    rng = np.random.default_rng(seed)
    cw = np.repeat(np.arange(W, dtype=np.int32), C)
    codes = np.empty((W, C, k), np.uint8)
    cur = np.stack([root_codes[p0 + w: p0 + w + k] for w in range(W)]).astype(np.uint8)
    codes[:, 0] = cur
    for c in range(1, C):
        pos = rng.integers(0, k, size=W)
        add = (1 << rng.integers(0, 4, size=W)).astype(np.uint8)
        cur = cur.copy()
        cur[np.arange(W), pos] |= add
        codes[:, c] = cur
    return cw, codes.reshape(W * C, k)


def expand_exceptions(ctx, n_ex, k, v):
    from multiprime_amd import iupac
    if not n_ex:
        return 0
    ew, er, ec = ctx.get_exceptions(n_ex)
    raw = iupac.strings_of(iupac.SYMBOL_LUT[ec])
    xw, xk = [], []
    for w_, s in zip(ew.tolist(), raw):
        if s.count("-") <= v:
            for e in iupac.expand(s):
                xw.append(w_)
                xk.append(e)
    if xw:
        chars = np.frombuffer("".join(xk).encode(), np.uint8).reshape(len(xk), k)
        ctx.set_extra_rows(np.asarray(xw, np.int32), iupac.words_of_kmers(chars))
    return len(xw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=131072, help="sequences per GPU (weak scaling)")
    ap.add_argument("--cols", type=int, default=1000)
    ap.add_argument("--k", type=int, default=18)
    ap.add_argument("--v", type=int, default=1)
    ap.add_argument("--cands", type=int, default=8, help="candidates per window")
    ap.add_argument("--seed", type=int, default=20250303)
    ap.add_argument("--cpu-rows", type=int, default=131072, help="rows of the shard timed on the CPU oracle (~13 s for the whole shard)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--bucket", type=int, default=4, help="steps whose counters share one all-reduce (N > 1)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from multiprime_amd import iupac
    from multiprime_amd._abi import Library
    from multiprime_amd.synth import synth_block, synth_root

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # MP_BENCH_BACKEND=gloo (testing only): several ranks on whatever GPUs the box has, collectives through the host —
    # exercises the N > 1 control flow of this file on a 1-GPU box; RCCL itself refuses two ranks on one device
    backend = os.environ.get("MP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    k, v, C, L = a.k, a.v, a.cands, a.cols
    # the library brackets every n-th mp_eval_launch with a HIP-event pair on its stream (live kernel time for the
    # roofline); a pair idles the stream for ~6 us, so the bench samples one launch in four
    os.environ.setdefault("MP_EVAL_TIMING_EVERY", "4")
    t_setup = time.time()
    lib = Library()                                   # the HIP library or an error: no fallback
    ctx = lib.context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    rows = synth_block(rank * a.rows, a.rows, L, a.seed)
    ctx.load_msa(rows.reshape(-1), np.arange(a.rows + 1, dtype=np.int64) * L)
    p0, W = 16, L - 32 - k                            # same windows on every rank
    n_ex = ctx.build_windows(p0, W, k, v)
    n_extra = expand_exceptions(ctx, n_ex, k, v)
    root_codes = np.array([1, 2, 4, 8], np.uint8)[synth_root(L, a.seed)]
    cw, codes = make_candidates(root_codes, p0, W, k, C, a.seed)
    f_set, r_set = {2, 3, k}, {2, k - 3, k - 2}        # -c 2,3,-1 (multiPrime.yaml), get_Y V20:1091
    sF = sum(1 << y for y in f_set if 0 <= y < k)
    sR = sum(1 << y for y in r_set if 0 <= y < k)
    n_cand = len(cw)
    # universe size per window (sequences with <= v gaps, plus expansions): an all-N candidate
    # matches every non-gap symbol, so perfect + F_mis under empty strict masks counts it
    alln = ctx.eval_candidates(np.arange(W, dtype=np.int32), np.full((W, k), 15, np.uint8), 0, 0)
    universe = alln[:, 0] + alln[:, 1]
    evals_local = int(universe.sum()) * C
    ctx.eval_upload(cw, codes, sF, sR)
    setup_s = time.time() - t_setup

    # N > 1: the counters of every step are all-reduced over RCCL, bucketed and overlapped (dist.StepBuckets):
    # `--bucket` consecutive steps write into one [bucket][n_cand][3] buffer that is reduced with ONE collective
    # (xGMI rings are latency-bound at 180 KB per step), on RCCL's stream, while the next bucket's steps evaluate
    # into the other buffer.  Every step's counters are reduced; all reductions complete inside the timed region.
    from multiprime_amd.dist import StepBuckets
    sb = StepBuckets(n_cand, a.bucket, dev, world)

    def step():
        ctx.eval_launch(sb.begin_step().data_ptr())
        sb.end_step()

    drain = sb.drain

    for _ in range(a.warmup):
        step()
    drain()
    ctx.eval_timing(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = ctx.eval_timing(reset=True)

    # measured device-copy bandwidth (SURVEY §8d asks for it next to the spec peak): 1 GiB d2d, read + write
    copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    ev = torch.tensor([evals_local], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ev, op=dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    evals_total = int(ev.item())
    checksum = sb.block_of(a.steps - 1).sum(dim=0).tolist()

    if rank == 0:
        traffic = None
        try:     # measured in a separate PMC pass of this same command; reported only on an exact config match
            for e in json.load(open(os.path.join(REPO, "profiles", "hbm_traffic.json")))["entries"]:
                if (e["rows"], e["cols"], e["k"], e["v"], e["cands"], e.get("mode")) == (a.rows, L, k, v, C, eval_mode()):
                    traffic = e["traffic_bytes"]
                    break
        except (OSError, KeyError, ValueError):
            pass
        per_launch_ms = kern_ms / max(kern_n, 1)
        alg_bytes = evals_local * 3 * k / 8.0         # 3k/8 bytes per evaluation
        achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9
        res = {
            "metric": "candidate x sequence coverage evals/s",
            "value": evals_total * a.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 bit-planes", "data": "synthetic",
            "config": {"workload": f"synthetic MSA {a.rows} x {L} per GPU (SURVEY 8d input 4; BASELINE configs[3] shard), "
                                   f"k={k}, v={v}, {C} candidates/window, {W} windows, strict -c 2,3,-1",
                       "rows_per_gpu": a.rows, "cols": L, "k": k, "variation": v, "candidates_per_window": C,
                       "windows": W, "evals_per_step_per_gpu": evals_local, "iupac_extra_rows": n_extra,
                       "parallelism": f"row shards x{world}, RCCL all-reduce of every step's [{n_cand}x3] int64 counters, {sb.B} steps per collective, overlapped with the next bucket"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": alg_bytes,
                         "kernel": KERNELS[eval_mode()] + "; timed region = counter memset + kernel", "eval_mode": eval_mode(), "kernel_ms": per_launch_ms, "launches_timed": kern_n, "timed_every": int(os.environ["MP_EVAL_TIMING_EVERY"]),
                         "algorithmic_bytes_per_eval": 3 * k / 8.0},
            "measured_copy_GBs": copy_gbs, "setup_s": setup_s, "device_bytes": ctx.device_bytes(), "counter_checksum": checksum,
        }
        if world == 1 and not a.no_cpu:
            res["cpu_baseline"] = cpu_baseline(rows[: a.cpu_rows], L, p0, W, k, v, cw, codes, sF, sR, C)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(rows, L, p0, W, k, v, cw, codes, sF, sR, C):
    """The oracle (plain-C restatement of the reference, one thread) on a bounded sample of the
    same workload: the first `cpu_rows` sequences of rank 0's shard, same windows and candidates.
    This is the only place bench.py touches oracle/."""
    from multiprime_amd._abi import Library
    so = os.path.join(REPO, "oracle", "_build", "libmprime_oracle.so")
    if not os.path.exists(so):
        return {"value": None, "unit": "evals/s", "cores": 1, "kind": "port", "sample": "oracle library not built"}
    ora = Library(so).context(0)
    n = rows.shape[0]
    ora.load_msa(rows.reshape(-1), np.arange(n + 1, dtype=np.int64) * L)
    n_ex = ora.build_windows(p0, W, k, v)
    expand_exceptions(ora, n_ex, k, v)
    alln = ora.eval_candidates(np.arange(W, dtype=np.int32), np.full((W, k), 15, np.uint8), 0, 0)
    evals = int((alln[:, 0] + alln[:, 1]).sum()) * C
    t0 = time.perf_counter()
    ora.eval_candidates(cw, codes, sF, sR)
    dt = time.perf_counter() - t0
    return {"value": evals / dt, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": f"first {n} sequences of the shard, all {W} windows x {C} candidates, {evals} evals in {dt:.1f} s "
                      f"(oracle/mprime_oracle.c, 1 thread; the Python reference itself measures 2-3e5 evals/s "
                      f"inside mis_primer_check, BASELINE.md)"}


if __name__ == "__main__":
    main()
