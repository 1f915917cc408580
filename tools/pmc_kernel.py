#!/usr/bin/env python3
"""Runs on the GPU box: several SQ counter passes of the whole drop-in core step (tools/pipeline_scale.py), raw per-kernel sums of
every counter for the kernels whose name contains --kernel (debug aid: what a pipeline kernel waits on).
usage: python tools/pmc_kernel.py --kernel hist_kernel [--rows N] [--out gpurun_out/r03/pmc_k]"""
import argparse
import glob
import os
import sqlite3
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=131072)
ap.add_argument("--kernel", default="hist_kernel")
ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "r03", "pmc_k"))
a = ap.parse_args()
out = os.path.abspath(a.out)
os.makedirs(out, exist_ok=True)
cmd = [sys.executable, os.path.join(REPO, "tools", "pipeline_scale.py"), "--rows", str(a.rows)]
env = dict(os.environ, TMPDIR="/tmp")
GROUPS = {
    "cycles": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INST_CYCLES_SALU", "SQ_ACTIVE_INST_ANY"],
    "lds": ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_ADDR_CONFLICT", "SQ_INSTS_VMEM_WR", "SQ_WAVES"],
    "mem": ["SQ_INST_CYCLES_VMEM_RD", "SQ_ACTIVE_INST_VMEM", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS", "SQ_LEVEL_WAVES", "SQ_INSTS_BRANCH", "SQ_ACTIVE_INST_MISC", "SQ_CYCLES"],
}
for tag, counters in GROUPS.items():
    with open(os.path.join(out, tag + ".log"), "w") as f:
        subprocess.call(["rocprofv3", "--pmc"] + counters + ["-d", os.path.join(out, tag), "-o", "pipe", "--"] + cmd, stdout=f, stderr=subprocess.STDOUT, cwd="/tmp", env=env,
                        timeout=600)
    for db in glob.glob(os.path.join(out, tag, "**", "*.db"), recursive=True):
        q = """select k.name, p.counter_name, sum(p.counter_value), count(distinct p.dispatch_id) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id
               group by k.name, p.counter_name"""
        for name, cname, v, n in sqlite3.connect(db).execute(q):
            if a.kernel in name:
                print(f"{name.split('(')[0][-40:]:40s} {cname:28s} {v / max(n, 1):16.0f}")
