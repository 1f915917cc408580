#!/usr/bin/env python3
"""Copies what tools/final_run.sh and tools/check_run.sh left under gpurun_out/r03/ into profiles/ (the tracked summaries).
usage: python tools/collect_profiles.py [--round r03]"""
import argparse
import json
import os
import shutil

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--round", default="r03")
a = ap.parse_args()
R = a.round
src = os.path.join(REPO, "gpurun_out", R)
dst = os.path.join(REPO, "profiles")


def copy(rel, name):
    p = os.path.join(src, rel)
    if os.path.exists(p):
        shutil.copyfile(p, os.path.join(dst, name))
        print("copied", rel, "->", name)
    else:
        print("MISSING", rel)


copy("final/r03_counters.json", f"{R}_counters.json")
copy("final/prof_bench/summary.txt", f"{R}_bench_eval.txt")
copy("final/prof_bench_1m/summary.txt", f"{R}_bench_eval_1m.txt")
copy("final/pipeline_kernels.txt", f"{R}_pipeline_kernels.txt")
copy("check/pytest_gpu.log", f"{R}_pytest_gpu.log")
p = os.path.join(src, "final", "bench.json")
if os.path.exists(p):
    lines = [l for l in open(p) if l.startswith("{")]
    if lines:
        open(os.path.join(dst, f"{R}_bench.json"), "w").write(lines[-1])
        print("bench line ->", f"{R}_bench.json")
# whole-step timings: one table from the fixture lines, then the two synthetic depths
p = os.path.join(src, "check", "pipeline_times.txt")
if os.path.exists(p):
    rows = [json.loads(l) for l in open(p) if l.startswith("{")]
    out = ["# End-to-end wall time of the drop-in core step on one MI355X box (tools/check_run.sh).  wall_s = constructor + run() in a process that has",
           "# the library loaded; cli_process_s = `python scripts/multiPrime-core.py ...` as a fresh process; reference = multiPrime-core_V20.py, 1 core, authoring container.",
           "# Every TSV byte-identical to the reference's.  Batch mode (one process for many clusters): profiles/r03_batch.txt.",
           f"{'fixture':18s} {'n_seq':>6s} {'windows':>8s} {'cands':>7s} {'wall_s':>7s} {'cli_process_s':>14s} {'ref_s':>8s} {'speedup':>8s}  tsv_identical"]
    for r in rows:
        out.append(f"{r['fixture']:18s} {r['n_seq']:6d} {r['windows']:8d} {r['n_candidates']:7d} {r['wall_s']:7.3f} {r['cli_process_s']:14.2f} "
                   f"{r['reference_wall_s']:8.2f} {r['speedup']:8.1f}  {r['tsv_identical']}")
    out += ["", "# synthetic deep alignments, whole core step incl. native FASTA parse (tools/pipeline_scale.py; --no-json, device-resident bitsets)"]
    for f in ("scale_131k.txt", "scale_1m.txt"):
        q = os.path.join(src, "check", f)
        if os.path.exists(q):
            out += [l.rstrip() for l in open(q) if l.startswith("{")][-1:]
    open(os.path.join(dst, f"{R}_pipeline_times.txt"), "w").write("\n".join(out) + "\n")
    print("pipeline times ->", f"{R}_pipeline_times.txt")
soak = []
for f, what in (("soak_parity.txt", "tools/soak_parity.py --seconds 150: random alignments / k / v / windows / candidate lists under every kernel setting; no difference"),
                ("soak_primers.txt", "tools/soak_primers.py --seconds 60: random primer lists through the dimer / pair-coverage / PCR kernels; no difference")):
    q = os.path.join(src, "check", f)
    if os.path.exists(q):
        soak += ["# " + what] + [l.rstrip() for l in open(q) if l.startswith("{")][-1:]
if soak:
    open(os.path.join(dst, f"{R}_soak.txt"), "w").write("# randomised HIP-vs-oracle soaks on the GPU box (tools/check_run.sh)\n" + "\n".join(soak) + "\n")
    print("soaks ->", f"{R}_soak.txt")
