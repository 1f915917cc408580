#!/usr/bin/env python3
"""BASELINE config 5's core step two ways (GPU box): C synthetic clusters (SURVEY §8d input 5; the workflow caps a cluster at 500
sequences, multiPrime.yaml:61-67) through the drop-in `scripts/multiPrime-core.py`
  per_cluster   one process per cluster, as Snakemake rule `multiPrime` launches them (multiPrime.py:200-207)
  batch         ONE process for all of them (--batch): interpreter, numpy and HIP start-up paid once
and checks that both write the same files.  Prints one JSON line (clusters per second of either mode).

  python tools/batch_bench.py --clusters 64 --rows 500"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multiprime_amd.synth import synth_block, to_fasta  # noqa: E402

SCRIPT = os.path.join(REPO, "scripts", "multiPrime-core.py")
FLAGS = ["-l", "18", "-n", "4", "-d", "10", "-v", "1", "-e", "3.6", "-g", "0.2,0.7", "-s", "150", "-f", "0.7", "-c", "2,3,-1", "-p", "1"]


def digest(path):
    h = hashlib.sha256()
    for suffix in ("", ".non_coverage_seq_id_json", ".gap_seq_id_json"):
        with open(path + suffix, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clusters", type=int, default=64)
    ap.add_argument("--rows", type=int, default=500)
    ap.add_argument("--seed", type=int, default=20250303)
    ap.add_argument("--no-per-cluster", action="store_true", help="skip the one-process-per-cluster leg")
    ap.add_argument("--workers", default="0", help="comma list of --batch-workers[xPROCS] settings to time (0 = the default), e.g. 2x4")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    cols = rng.integers(800, 2000, size=a.clusters)
    with tempfile.TemporaryDirectory() as wd:
        inputs = []
        for i in range(a.clusters):
            fa = os.path.join(wd, f"Cluster_{i}.tmsa")
            with open(fa, "wb") as f:
                f.write(to_fasta(synth_block(0, a.rows, int(cols[i]), a.seed + 1000 * (i + 1))))
            inputs.append(fa)
        subprocess.run([sys.executable, SCRIPT, "-i", inputs[0], "-o", os.path.join(wd, "warm.out")] + FLAGS, check=True, capture_output=True)
        t0 = time.time()
        for i, fa in enumerate(inputs if not a.no_per_cluster else inputs[:4]):
            subprocess.run([sys.executable, SCRIPT, "-i", fa, "-o", os.path.join(wd, f"p{i}.out")] + FLAGS, check=True, capture_output=True)
        per_cluster_s = (time.time() - t0) * (1 if not a.no_per_cluster else a.clusters / 4)
        batch = os.path.join(wd, "batch.tsv")
        with open(batch, "w") as f:
            f.writelines(f"{fa}\t{os.path.join(wd, f'b{i}.out')}\n" for i, fa in enumerate(inputs))
        by_workers = {}
        for wk in a.workers.split(","):
            t0 = time.time()
            wt, _, procs = wk.partition("x")
            r = subprocess.run([sys.executable, SCRIPT, "--batch", batch, "--batch-workers", wt, "--batch-procs", procs or "1"] + FLAGS, check=True,
                               capture_output=True, text=True)
            batch_s = time.time() - t0
            inner = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{") and ("processes" in line or not procs)][-1]
            by_workers[wk] = {"process_s": round(batch_s, 2), "clusters_per_s": round(a.clusters / batch_s, 2),
                              "inner_clusters_per_s": inner["clusters_per_s"], "workers": inner.get("workers")}
        same = all(digest(os.path.join(wd, f"p{i}.out")) == digest(os.path.join(wd, f"b{i}.out")) for i in range(a.clusters if not a.no_per_cluster else 4))
        inner = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")][-1]
        primers = sum(sum(1 for _ in open(os.path.join(wd, f"b{i}.out"))) - 1 for i in range(a.clusters))
    print(json.dumps({"clusters": a.clusters, "rows_per_cluster": a.rows, "columns": [int(cols.min()), int(cols.max())], "flags": " ".join(FLAGS),
                      "per_cluster_processes_s": round(per_cluster_s, 2), "per_cluster_clusters_per_s": round(a.clusters / per_cluster_s, 2),
                      "batch_process_s": round(batch_s, 2), "batch_clusters_per_s": round(a.clusters / batch_s, 2),
                      "batch_inner_clusters_per_s": inner["clusters_per_s"], "batch_by_workers": by_workers, "identical_outputs": same, "primers_written": primers,
                      "reference_note": "the reference takes 50-62 s per 500-sequence cluster (BASELINE.md)"}))


if __name__ == "__main__":
    main()
