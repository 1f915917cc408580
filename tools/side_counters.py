#!/usr/bin/env python3
"""Runs on the GPU box: rocprofv3 kernel trace + two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/side_bench.py; per kernel
of the dimer / PCR / k-mismatch blocks: dispatches, average duration, bytes fetched from and written to the fabric (HBM / Infinity Cache)
per dispatch.  Writes <out>/side_kernels.json — copy to profiles/r06_side_kernels.json, which side_bench.py merges into its blocks
(`kernels`).  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes: x 2 (MI355X_MICROARCH.md, HBM section);
WRITE_SIZE is uncalibrated there and reported as counted.
usage: python tools/side_counters.py --out gpurun_out/r06/side_counters"""
import argparse
import glob
import json
import os
import sqlite3
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "r06", "side_counters"))
a = ap.parse_args()
out = os.path.abspath(a.out)
os.makedirs(out, exist_ok=True)
cmd = [sys.executable, os.path.join(REPO, "tools", "side_bench.py")]
env = dict(os.environ, TMPDIR="/tmp")
for tag, extra in (("trace", ["--kernel-trace", "--stats"]), ("fetch", ["--pmc", "FETCH_SIZE"]), ("write", ["--pmc", "WRITE_SIZE"])):
    with open(os.path.join(out, tag + ".log"), "w") as f:
        subprocess.call(["rocprofv3"] + extra + ["-d", os.path.join(out, tag), "-o", "side", "--"] + cmd, stdout=f, stderr=subprocess.STDOUT, cwd="/tmp", env=env, timeout=900)


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def db_of(tag):
    g = glob.glob(os.path.join(out, tag, "**", "*.db"), recursive=True)
    return sqlite3.connect(g[0]) if g else None


kern = {}
t = db_of("trace")
if t:
    for name, calls, total, avg in t.execute("select name,total_calls,total_duration,average from top_kernels order by total_duration desc"):
        kern[short(name)] = {"dispatches": calls, "avg_us": avg}
for tag, key in (("fetch", "fetch_bytes"), ("write", "write_bytes")):
    d = db_of(tag)
    if not d:
        continue
    q = """select name, avg(v) from (select k.name as name, p.dispatch_id as d, sum(p.counter_value) as v from pmc_events p join kernels k
           on p.dispatch_id = k.dispatch_id group by k.name, p.dispatch_id) group by name"""
    for name, v in d.execute(q):
        # KB; FETCH_SIZE counts a 128-byte request as 64 bytes on gfx950 (x 2, MI355X_MICROARCH.md: HBM); WRITE_SIZE is uncalibrated there: as counted
        kern.setdefault(short(name), {})[key] = v * 1024.0 * (2.0 if key == "fetch_bytes" else 1.0)
blocks = {"dimer_scan": ("dimer_rows_kernel", "dimer_group_kernel", "prim_kernel"), "pcr_scan": ("pcr_block_kernel", "seq_pack_kernel"), "kmm_scan": ("kmm_kernel",)}
res = {b: {k: v for k, v in kern.items() if k.startswith(names)} for b, names in blocks.items()}
res["_note"] = ("rocprofv3 of tools/side_bench.py: --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE in runs of their own; bytes per dispatch "
                "(averages over all dispatches of a kernel in the script: <..., false> = text per call, <..., true> = resident store)")
with open(os.path.join(out, "side_kernels.json"), "w") as f:
    json.dump(res, f, indent=1)
for b, ks in res.items():
    if b.startswith("_"):
        continue
    for k, v in ks.items():
        print(f"{b:11s} {k:36s} n={v.get('dispatches', 0):3d} avg {v.get('avg_us', 0):9.1f} us  fetch {v.get('fetch_bytes', 0) / 1e6:9.2f} MB  write {v.get('write_bytes', 0) / 1e6:9.2f} MB")
