#!/usr/bin/env python3
"""Per-kernel averages of every counter found under a tools/pmc_variants.sh output directory (rocpd sqlite files).
usage: tools/pmc_report.py gpurun_out/prof_<tag> [name-filter]"""
import glob
import os
import sqlite3
import sys


def main(root, flt="eval_"):
    acc = {}
    for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*.db"), recursive=True)):
        db = sqlite3.connect(f)
        q = """select k.name, k.grid_x, p.counter_name, p.dispatch_id, sum(p.counter_value) from pmc_events p join kernels k
               on p.dispatch_id = k.dispatch_id group by k.name, k.grid_x, p.counter_name, p.dispatch_id"""
        for name, grid, cname, _, val in db.execute(q):
            if flt in name:
                acc.setdefault((name.replace("(anonymous namespace)::", "").split("(")[0], grid), {}).setdefault(cname, []).append(val)
    for (name, grid), ctrs in sorted(acc.items()):
        print(f"== {name}  grid_x={grid}")
        for cname, vals in sorted(ctrs.items()):
            print(f"   {cname:28s} n={len(vals):3d} avg={sum(vals) / len(vals):16.1f}")


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]))
