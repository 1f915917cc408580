#!/usr/bin/env python3
"""Runs on the GPU box (via gpurun): rocprofv3 kernel trace + SEPARATE --pmc passes of `python bench.py` (and of the read
kernels of tools/_build/ubench, whose byte counts are known, to calibrate bytes per L2 request), then condenses the
per-dispatch counters of the timed evaluation kernel into one JSON entry keyed by the bench configuration:

    python tools/collect_counters.py [--rows 131072] [--out gpurun_out/r02/prof_bench] [--tag r02]

writes <out>/counters.json (copy to profiles/r03_counters.json: bench.py reads it for roofline.valu_frac / l2_frac /
hbm_frac / traffic on an exact match of configuration AND kernel source hash) and <out>/summary.txt (tools/summarize_profile.py view).
Counter passes never combine --pmc with tracing domains other than the kernel trace.
"""
import argparse
import glob
import json
import os
import sqlite3
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PASSES = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "sq": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY"],
    "l2": ["TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"],
    "grbm": ["GRBM_GUI_ACTIVE", "TCC_READ_sum"],
}


def run(cmd, log):
    with open(log, "w") as f, open(log + ".err", "w") as e:
        # (MP_BENCH_STREAMS=1: one launch at a time on the chip — a kernel's duration and counters are its own)
        return subprocess.call(cmd, stdout=f, stderr=e, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", MP_BENCH_STREAMS="1"), timeout=600)


def per_kernel(dbfile):
    """{(kernel name, counter): [value per dispatch, in dispatch order]} — a counter is reported once per hardware instance
    (XCD / SE): the instances of a dispatch are summed."""
    db = sqlite3.connect(dbfile)
    q = """select k.name, p.counter_name, p.dispatch_id, sum(p.counter_value) from pmc_events p join kernels k
           on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name, p.dispatch_id order by p.dispatch_id"""
    out = {}
    for name, cname, _, val in db.execute(q):
        out.setdefault((name, cname), []).append(val)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=131072)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "r04", "prof_bench"))
    ap.add_argument("--bench-args", default="")
    ap.add_argument("--merge", default="", help="an earlier counters.json whose entries for OTHER configurations are kept")
    a = ap.parse_args()
    a.out = os.path.abspath(a.out)                 # rocprofv3 runs from /tmp
    os.makedirs(a.out, exist_ok=True)
    bench = [sys.executable, os.path.join(REPO, "bench.py"), "--rows", str(a.rows), "--no-cpu", "--no-variants", "--no-full", "--no-pipeline", "--no-side", "--no-shapes", "--no-ksweep"] + a.bench_args.split()
    # kernel trace of the bench command itself
    run(["rocprofv3", "--kernel-trace", "--stats", "-d", os.path.join(a.out, "trace"), "-o", "bench", "--"] + bench + ["--steps", "20", "--warmup", "3"],
        os.path.join(a.out, "trace_bench.json"))
    for tag, ctrs in PASSES.items():
        run(["rocprofv3", "--pmc"] + ctrs + ["-d", os.path.join(a.out, "pmc_" + tag), "-o", "bench", "--"] + bench + ["--steps", "5", "--warmup", "1"],
            os.path.join(a.out, f"pmc_{tag}_bench.json"))
    ub = os.path.join(REPO, "tools", "_build", "ubench")
    if os.path.exists(ub):
        run(["rocprofv3", "--pmc"] + PASSES["l2"] + ["-d", os.path.join(a.out, "cal_l2"), "-o", "ub", "--", ub], os.path.join(a.out, "cal_l2_ubench.json"))
        run(["rocprofv3", "--pmc", "FETCH_SIZE", "-d", os.path.join(a.out, "cal_fetch"), "-o", "ub", "--", ub], os.path.join(a.out, "cal_fetch_ubench.json"))

    # ---- condense ----
    line = None
    for l in open(os.path.join(a.out, "trace_bench.json")):
        if l.startswith("{"):
            line = json.loads(l)
    sys.path.insert(0, REPO)
    import bench as bench_mod
    entry = {"rows": a.rows, "source_hash": bench_mod.kernel_source_hash()}
    if line:
        c = line["config"]
        entry.update({"cols": c["cols"], "k": c["k"], "v": c["variation"], "cands": c["candidates_per_window"], "mode": line["roofline"]["eval_mode"]})
    tdb = glob.glob(os.path.join(a.out, "trace", "**", "*.db"), recursive=True)
    kern = None
    if tdb:
        db = sqlite3.connect(tdb[0])
        rows = list(db.execute("select name, count(*), avg(duration), min(duration), max(duration) from kernels where name like '%eval_%' group by name order by sum(duration) desc"))
        if rows:
            kern = rows[0][0]
            entry["kernel"] = kern.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip() + " (counters: summed with the other eval_ kernels of a step)"
            entry["trace_all"] = [{"kernel": r[0].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip(), "dispatches": r[1], "avg_us": r[2] / 1e3} for r in rows]
            entry["trace"] = {"dispatches": rows[0][1], "avg_us": rows[0][2] / 1e3, "min_us": rows[0][3] / 1e3, "max_us": rows[0][4] / 1e3,
                              "note": "rocprofv3 --kernel-trace of `bench.py --steps 20 --warmup 3`"}
    # a step = every evaluation kernel of one mp_eval_launch (the sliding kernel + eval_chain_kernel on the patch planes): per
    # kernel the per-dispatch average, summed over the kernels; one dispatch of each per step
    raw, per_kernel_raw = {}, {}
    for tag in PASSES:
        dbs = glob.glob(os.path.join(a.out, "pmc_" + tag, "**", "*.db"), recursive=True)
        if not dbs:
            continue
        pk = per_kernel(dbs[0])
        most = max([len(v) for (name, _), v in pk.items() if "eval_" in name] or [0])
        for (name, cname), vals in pk.items():
            if "eval_" in name and 2 * len(vals) >= most:                 # (a kernel that ran once belongs to the set-up, not to the steps)
                short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
                per_kernel_raw.setdefault(short, {})[cname] = sum(vals) / len(vals)
                raw[cname] = raw.get(cname, 0.0) + sum(vals) / len(vals)
    entry["kernels_of_a_step"] = per_kernel_raw
    entry["raw_counters_per_launch"] = raw
    # calibration: bytes per TCP->TCC read request on coalesced dword / dwordx4 reads of known size
    cal = {}
    ubj = None
    try:
        ubj = json.load(open(os.path.join(a.out, "cal_l2_ubench.json")))
    except (OSError, ValueError):
        pass
    dbs = glob.glob(os.path.join(a.out, "cal_l2", "**", "*.db"), recursive=True)
    if ubj and dbs:
        pk = per_kernel(dbs[0])
        cases = ubj["reads"]
        for vec, label in ((1, "dword"), (4, "dwordx4")):
            names = [n for (n, cn) in pk if "read_kernel" in n and f"<{vec}>" in n and cn == "TCP_TCC_READ_REQ_sum"]
            if not names:
                continue
            vals = pk[(names[0], "TCP_TCC_READ_REQ_sum")]
            my_cases = [cs for cs in cases if (cs["case"].endswith("dwordx4")) == (vec == 4)]
            # every case launches its kernel 5 times (2 warm-ups, 3 timed); the HBM case 4 times
            i = 0
            for cs in my_cases:
                reps = 4 if cs["case"].startswith("hbm") else 5
                if i + reps > len(vals):
                    break
                req = sum(vals[i:i + reps]) / reps
                i += reps
                if cs.get("bytes_per_launch") and req:
                    cal[cs["case"]] = {"bytes_per_launch": cs["bytes_per_launch"], "TCP_TCC_READ_REQ": req, "bytes_per_request": cs["bytes_per_launch"] / req}
    entry["calibration"] = cal
    bpr = cal.get("l2_dword", {}).get("bytes_per_request") or 64.0
    entry["l2_bytes_per_request"] = bpr
    if "SQ_INSTS_VALU" in raw:
        entry["valu_insts"] = raw["SQ_INSTS_VALU"]
    if "TCP_TCC_READ_REQ_sum" in raw:
        entry["l2_read_bytes"] = raw["TCP_TCC_READ_REQ_sum"] * bpr
    if "FETCH_SIZE" in raw:
        entry["hbm_read_bytes"] = raw["FETCH_SIZE"] * 1024 * 2      # KB; gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)
    if "WRITE_SIZE" in raw:
        entry["hbm_write_bytes"] = raw["WRITE_SIZE"] * 1024
    entry["source"] = "tools/collect_counters.py: separate rocprofv3 --pmc passes of `python bench.py --steps 5 --warmup 1 --no-cpu --no-variants`, per-dispatch average of the timed kernel"
    entries = [entry]
    if a.merge and os.path.exists(a.merge):          # keep the entries of other configurations collected earlier
        old = json.load(open(a.merge)).get("entries", [])
        entries += [e for e in old if (e.get("rows"), e.get("mode")) != (entry.get("rows"), entry.get("mode"))]
    json.dump({"entries": entries}, open(os.path.join(a.out, "counters.json"), "w"), indent=1)
    print(json.dumps(entry, indent=1))
    with open(os.path.join(a.out, "summary.txt"), "w") as f:
        subprocess.call([sys.executable, os.path.join(REPO, "tools", "summarize_profile.py"), a.out], stdout=f)


if __name__ == "__main__":
    main()
