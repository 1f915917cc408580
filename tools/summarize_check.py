#!/usr/bin/env python3
"""Turns the outputs of tools/check_run.sh (gpurun_out/r02/check/) into the tracked summaries: profiles/r02_pipeline_times.txt,
profiles/r02_dimer_pcr.txt, profiles/r02_pytest_gpu.log."""
import json
import os
import shutil
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(REPO, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02/check")
P = os.path.join(REPO, "profiles")


def last_json(path):
    return json.loads([ln for ln in open(path).read().strip().split("\n") if ln.startswith("{")][-1])


def phases(d):
    return ", ".join(f"{k} {v}" for k, v in d.items())


rows = [json.loads(ln) for ln in open(os.path.join(O, "pipeline_times.txt")) if ln.startswith("{")]
out = ["# End-to-end wall time of the drop-in core step on one MI355X box, round 2 (tools/check_run.sh -> tools/summarize_check.py).",
       "# wall_s = constructor + run() in a process that has the library loaded (best of two); a cold process adds the HIP runtime start-up",
       "# (context_s of tools/load_bench.py below).  reference = multiPrime-core_V20.py, 1 core, authoring container (golden trace meta).",
       "# Every TSV byte-identical to the reference's.  Round-1 figures: profiles/r01_pipeline_times.txt; at mid-round (Python JSON writer,",
       "# scalar Tm / filters) the same fixtures took 0.07-1.3 s.",
       "# cli_s = `python scripts/multiPrime-core.py ...` as a fresh process (interpreter, numpy, HIP start-up included); the reference's figure is its own process.",
       "fixture             n_seq  windows   cands  wall_s   cli_s    ref_s  speedup  phases >= 5 ms"]
for r in rows:
    ph = ", ".join(f"{k} {v}" for k, v in r["phases"].items() if v >= 0.005)
    out.append(f"{r['fixture']:18s} {r['n_seq']:6d} {r['windows']:8d} {r['n_candidates']:7d} {r['wall_s']:7.3f} {r.get('cli_process_s', float('nan')):7.2f} {r['reference_wall_s']:8.2f} "
               f"{r['speedup']:7.1f}x  {ph}")
a, b = last_json(os.path.join(O, "scale_131k.txt")), last_json(os.path.join(O, "scale_1m.txt"))
out.append("#")
for d in (a, b):
    out.append(f"# synthetic {d['rows']} x {d['cols']} (k=18, v=1, -d 10 -f 0.8 -c 2,3,-1, --no-json, device-resident bitsets): core wall {d['wall_s']} s "
               f"(incl. context creation beside the parse), {d['windows']} windows, {d['windows_past_the_gates']} past the gates, {d['rows_out']} primers, "
               f"device {d['device_bytes'] / 1e9:.2f} GB")
    out.append("#   phases: " + phases(d["phases"]))
for name in ("scale_131k.txt", "scale_1m.txt"):
    for ln in open(os.path.join(O, name)):
        if ln.startswith('{"pairing_wall_s"'):
            out.append(f"#   pairing stage after {name[6:-4]} rows: " + ln.strip())
lb = os.path.join(O, "load_bench.txt")
if os.path.exists(lb):
    out.append("#")
    out.append("# tools/load_bench.py (10^6 x 1000, 1 GB FASTA): " + open(lb).read().strip())
m = last_json(os.path.join(O, "multi_cluster.txt"))
out.append("#")
out.append("# 64 clusters through the whole rule chain (tools/multi_cluster.py, BASELINE config 5), one GPU (round 1: core_s 25.4, pairing_s 5.1):")
out.append("#   " + json.dumps(m))
open(os.path.join(P, "r02_pipeline_times.txt"), "w").write("\n".join(out) + "\n")

out = ["# All-pairs dimer scan (finDimer / get_Maxprimerset) and in-silico PCR kernels on one MI355X, round 2 (tools/check_run.sh).",
       "# dimer_rows_kernel = the all-pairs triangle scan (2024 / 8000 / 20000 primers in one launch each); the round-1 thread-per-pair kernel took",
       "# 8 / 96 / 580 ms on the same inputs, the same hits.  dimer_group_kernel = short lists (warm-up calls of 50 primers here).",
       "# pcr_block_kernel: round-1 pcr_kernel took 35 ms on the same database."]
for tag, cmd in (("dimer", "tools/dimer_bench.py"), ("pcr", "tools/pcr_bench.py")):
    db = sqlite3.connect(os.path.join(O, f"prof_{tag}", f"{tag}_results.db"))
    out.append(f"\n## rocprofv3 --kernel-trace --stats -- python {cmd}")
    out.append(f"{'kernel':60s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s}")
    for name, calls, total, avg in db.execute("select name,total_calls,total_duration,average from top_kernels order by total_duration desc"):
        name = name.replace("(anonymous namespace)::", "").split("(")[0][:60]
        out.append(f"{name:60s} {calls:6d} {total:12.1f} {avg:10.2f}")
    if tag == "dimer":
        per = [f"{d / 1e3:.0f} us" for (d,) in db.execute("select duration from kernels where name like '%dimer_rows%' order by start")]
        out.append("dimer_rows_kernel per launch: " + " / ".join(per))
    out.append(open(os.path.join(O, f"{tag}_bench.txt")).read().strip())
open(os.path.join(P, "r02_dimer_pcr.txt"), "w").write("\n".join(out) + "\n")
shutil.copy(os.path.join(O, "pytest_gpu.log"), os.path.join(P, "r02_pytest_gpu.log"))
print(open(os.path.join(P, "r02_pipeline_times.txt")).read()[:2500])
