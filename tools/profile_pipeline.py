#!/usr/bin/env python3
"""Runs on the GPU box: rocprofv3 kernel trace + one SQ counter pass of the WHOLE drop-in core step on the synthetic bench shard
(tools/pipeline_scale.py), and prints per kernel: calls, total / average duration, VALU / SALU / VMEM wave-instructions, busy and
issue-stall cycles, and the VALU issue fraction against the measured ceiling (profiles/r02_ubench.json).
usage: python tools/profile_pipeline.py [--rows N] [--out gpurun_out/r02/prof_pipe] > profiles/r02_pipeline_kernels.txt"""
import argparse
import glob
import json
import os
import sqlite3
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=131072)
ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "r02", "prof_pipe"))
ap.add_argument("--json", default="", help="also write the per-kernel rows {kernel, calls, avg_us, bytes_compulsory, frac} to this file (merged by rows: "
                                             "bench.py's pipeline block reads profiles/r06_pipeline_kernels.json)")
a = ap.parse_args()
out = os.path.abspath(a.out)
os.makedirs(out, exist_ok=True)
cmd = [sys.executable, os.path.join(REPO, "tools", "pipeline_scale.py"), "--rows", str(a.rows)]
env = dict(os.environ, TMPDIR="/tmp")
for tag, extra in (("trace", ["--kernel-trace", "--stats"]),
                   ("pmc", ["--pmc", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"])):
    with open(os.path.join(out, tag + ".log"), "w") as f:
        subprocess.call(["rocprofv3"] + extra + ["-d", os.path.join(out, tag), "-o", "pipe", "--"] + cmd, stdout=f, stderr=subprocess.STDOUT, cwd="/tmp", env=env, timeout=900)


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]


ceil = 8.5e8
try:
    ub = json.load(open(os.path.join(REPO, "profiles", "r02_ubench.json")))
    ceil = max(r["wave_instr_per_s_per_simd"] for r in ub["valu"] if r["op"] in ("v_bitop3_b32", "v_add_u32", "v_xor_b32"))
except (OSError, ValueError, KeyError):
    pass
tdb = glob.glob(os.path.join(out, "trace", "**", "*.db"), recursive=True)
pdb = glob.glob(os.path.join(out, "pmc", "**", "*.db"), recursive=True)
pmc = {}
if pdb:
    q = """select name, counter_name, avg(v) from (select k.name as name, p.counter_name as counter_name, p.dispatch_id as d, sum(p.counter_value) as v
           from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name, p.dispatch_id) group by name, counter_name"""
    for name, cname, v in sqlite3.connect(pdb[0]).execute(q):
        pmc.setdefault(name, {})[cname] = v
print(f"# whole drop-in core step, synthetic {a.rows} x 1000, k=18 v=1 (tools/pipeline_scale.py) under rocprofv3: kernel trace, then one SQ counter pass")
print(f"# valu_frac = VALU wave-instructions / (avg duration x 1024 SIMDs x {ceil:.3g} wave-instr/s/SIMD measured by tools/ubench.hip)")
print(f"{'kernel':44s} {'calls':>5s} {'total_us':>10s} {'avg_us':>9s} {'VALU':>11s} {'SALU':>11s} {'VMEM_RD':>9s} {'LDS':>9s} {'waves':>8s} {'wait_inst/busy':>14s} {'valu_frac':>9s}")
if tdb:
    for name, calls, total, avg in sqlite3.connect(tdb[0]).execute("select name,total_calls,total_duration,average from top_kernels order by total_duration desc"):
        c = pmc.get(name, {})
        vf = c.get("SQ_INSTS_VALU", 0) / (avg * 1e-6 * 1024 * ceil) if avg else 0
        ratio = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_BUSY_CYCLES"] if c.get("SQ_BUSY_CYCLES") else 0
        print(f"{short(name):44s} {calls:5d} {total:10.1f} {avg:9.1f} {c.get('SQ_INSTS_VALU', 0):11.0f} {c.get('SQ_INSTS_SALU', 0):11.0f} "
              f"{c.get('SQ_INSTS_VMEM_RD', 0):9.0f} {c.get('SQ_INSTS_LDS', 0):9.0f} {c.get('SQ_WAVES', 0):8.0f} {ratio:14.2f} {vf:9.3f}")


# ---- per-kernel rows with the bytes an ideal kernel of the formulation must move (N rows x L = 1000 columns, W windows of k = 18) -------------
if a.json and tdb:
    N, L, k = float(a.rows), 1000.0, 18
    W = L - k - 36                                   # windows of the primer region on the synthetic alignment (982 at the bench sizes: region 18..1000)
    slots = 16384.0
    while slots < N / 16:
        slots *= 2
    model = [   # (kernel-name prefix, bytes per call, what)
        ("pack_kernel", N * L * 1.5, "characters in, 4 one-bit planes out"),
        ("row_scan_kernel", N * L * (0.5 + 0.125 + 0.5), "planes in; prefix counts and gap-free nibbles out"),
        ("colplane_kernel", N * L * 1.0, "planes in, column planes out"),
        ("classify_kernel", N * L * 0.5 + W * N / 8, "planes in, exclusion words out (per pass)"),
        ("hist", N * L * 3 / 8, "SURVEY 8d (ii): the packed planes once (every window re-reads its 8 plane words: 36 B per row and window through L2)"),
        ("table_sums_kernel", W * slots * 12, "every slot's key and count once"),
        ("compact_kernel", W * 0.45 * slots * 16, "the tables of the windows the gate leaves (~45 %) once"),
        ("window_stats", N * L * 0.5 + W * N / 8, "column planes and exclusion words once (window_stats_group_kernel from 32768 rows on)"),
        # the slow (window, row) pairs — edge-gap repair, ragged end, IUPAC: 0.55 % of the pairs of the synthetic alignment
        ("repair_kernel", N * W * 0.0055 * (32 + 8 + 12 + 4), "per slow pair: its 8 plane words, two prefix counts, 3 window words + row out"),
        ("plain_planes_kernel", N * W * 0.0055 * (32 + 18 * 4 / 8.0), "per slow pair: its 8 plane words in, k x 4 plane bits out"),
        ("patch_planes_kernel", N * W * 0.0055 * (12 + 18 * 4 / 8.0), "per slow pair: 3 window words in, k x 4 plane bits out"),
        ("mask_patch_kernel", N * W * 0.0055 * (12 + 4), "per slow pair of an output window: its window words in, one verdict bit pair out (only the ~410 output windows: an upper bound)"),
        ("eval_slide_kernel", N * L * 3 / 8, "SURVEY 8d (ii)"), ("eval_chain_kernel", N * L * 3 / 8, "SURVEY 8d (ii)"), ("eval_bits_kernel", N * L * 3 / 8, "SURVEY 8d (ii)"),
    ]
    rows = []
    for name, calls, total, avg in sqlite3.connect(tdb[0]).execute("select name,total_calls,total_duration,average from top_kernels order by total_duration desc"):
        sn = short(name)
        m = next((x for x in model if sn.startswith(x[0]) or (" " + x[0]) in sn or sn.split("<")[0].endswith(x[0])), None)
        row = {"kernel": sn, "calls": calls, "avg_us": avg, "total_us": total}
        if m:
            row.update({"bytes_compulsory": m[1], "frac": m[1] / (avg * 1e-6) / 8e12, "bytes_model": m[2]})
        c = pmc.get(name, {})
        if c.get("SQ_INSTS_VALU"):
            row["valu_frac"] = c["SQ_INSTS_VALU"] / (avg * 1e-6 * 1024 * ceil)
        rows.append(row)
    try:
        with open(a.json) as f:
            db = json.load(f)
    except (OSError, ValueError):
        db = {}
    db[f"rows_{a.rows}"] = {"what": "rocprofv3 --kernel-trace of the whole core step (tools/pipeline_scale.py); frac = bytes_compulsory / avg duration / 8 TB/s", "kernels": rows}
    with open(a.json, "w") as f:
        json.dump(db, f, indent=1)
