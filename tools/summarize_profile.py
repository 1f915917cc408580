#!/usr/bin/env python3
"""Condense the rocprofv3 (rocpd sqlite) outputs of tools/profile_bench.sh into a text summary
for profiles/: per-kernel call count / total / average duration (the `--kernel-trace --stats`
view), and per-kernel averages of every PMC counter collected in the separate --pmc passes.

usage: tools/summarize_profile.py gpurun_out/prof_<tag> > profiles/<name>.txt
"""
import glob
import json
import os
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0][:70]


def main(root):
    print(f"# rocprofv3 summary of {root}")
    tdb = glob.glob(os.path.join(root, "trace", "*.db"))
    if tdb:
        db = sqlite3.connect(tdb[0])
        print("\n## kernel trace (rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu)")
        print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
        for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
            print(f"{short(name):70s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}")
        rows = list(db.execute("select name, duration, grid_x, grid_y, workgroup_x, vgpr_count, sgpr_count, lds_size from kernels where name like '%eval_%' order by start"))
        if rows:
            d = [r[1] for r in rows]
            print(f"\neval_kernel dispatches: n={len(d)} min={min(d)/1e3:.1f}us median={sorted(d)[len(d)//2]/1e3:.1f}us max={max(d)/1e3:.1f}us "
                  f"grid=({rows[-1][2]},{rows[-1][3]}) wg={rows[-1][4]} vgpr={rows[-1][5]} sgpr={rows[-1][6]} lds={rows[-1][7]}")
        bj = os.path.join(root, "trace_bench.json")
        if os.path.exists(bj):
            line = [l for l in open(bj) if l.startswith("{")]
            if line:
                j = json.loads(line[-1])
                print(f"bench.py under the tracer: value={j['value']:.4g} {j['unit']} ms_per_step={j['ms_per_step']:.4f} "
                      f"hip-event kernel_ms={j['roofline']['kernel_ms']:.4f} achieved={j['roofline']['achieved']:.1f} GB/s frac={j['roofline']['frac']:.3f}")
    for pdir in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        dbs = glob.glob(os.path.join(pdir, "*.db"))
        if not os.path.isdir(pdir) or not dbs:
            continue
        db = sqlite3.connect(dbs[0])
        print(f"\n## PMC pass {os.path.basename(pdir)} (separate run; per-dispatch average over the dispatches of each kernel)")
        # a counter is reported once per hardware instance (XCD / SE): sum the instances of a dispatch,
        # then average over the dispatches of the kernel
        q = """select name, counter_name, count(*), avg(v), min(v), max(v) from
               (select k.name as name, p.counter_name as counter_name, p.dispatch_id as d, sum(p.counter_value) as v
                from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name, p.dispatch_id)
               group by name, counter_name order by name"""
        try:
            for name, cname, n, avg, mn, mx in db.execute(q):
                if "eval_" in name or "unique" in name or "build_windows" in name or "dimer" in name:
                    note = ""
                    if cname == "FETCH_SIZE":
                        note = f"  -> {avg * 1024 * 2 / 1e9:.3f} GB/launch HBM read (KB x 2: gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM)"
                    if cname == "WRITE_SIZE":
                        note = f"  -> {avg * 1024 / 1e9:.4f} GB/launch written (uncalibrated)"
                    print(f"{short(name):40s} {cname:20s} dispatches={n:3d} avg={avg:16.1f} min={mn:16.1f} max={mx:16.1f}{note}")
        except sqlite3.Error as e:
            print("query failed:", e)


if __name__ == "__main__":
    main(sys.argv[1])
