#!/usr/bin/env python3
"""BASELINE config 5 in one process per GPU: many clusters through the whole rule chain with this build's drop-ins.

For each of C synthetic clusters (different roots and seeds, 500 .. 50 000 sequences; SURVEY §8d input 5):
  rule multiPrime      -> multiprime_amd.core.NN_degenerate          ({i}.top.primer.out)
  rule get_multiPrime  -> multiprime_amd.pairing.Primers_filter      ({i}.candidate.primers.txt)
then `cat` (rule aggregate_candidate_primers), rule get_Maxprimerset -> multiprime_amd.maxset, the
primerset_format shim and rule all_mfeprimer_check's finDimer -> multiprime_amd.dimer.  Flags are multiPrime.yaml's.
Clusters are independent: with WORLD_SIZE > 1 rank r takes clusters r, r + N, ... (no data-path collective, one
barrier before rank 0 aggregates).  Prints one JSON line with the per-stage wall times.

  python tools/multi_cluster.py --clusters 64
  python -m torch.distributed.run --nproc-per-node 8 tools/multi_cluster.py --clusters 64
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multiprime_amd.core import NN_degenerate  # noqa: E402
from multiprime_amd.synth import synth_block, to_fasta  # noqa: E402

ADAPTOR = "TCTTTCCCTACACGACGCTCTTCCGATCT,TGGAGTTCAGACGTGTGCTCTTCCGATCT"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clusters", type=int, default=64)
    ap.add_argument("--min-rows", type=int, default=500)
    ap.add_argument("--max-rows", type=int, default=50000)
    ap.add_argument("--seed", type=int, default=20250303)
    ap.add_argument("--workdir", default=None, help="shared directory (default: a temporary one; required for N > 1)")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        if a.workdir is None:
            raise SystemExit("--workdir (a directory all ranks see) is required with more than one rank")
    ctxm = tempfile.TemporaryDirectory() if a.workdir is None else contextlib.nullcontext(a.workdir)
    with ctxm as wd:
        os.makedirs(wd, exist_ok=True)
        rng = np.random.default_rng(a.seed)
        sizes = np.exp(rng.uniform(np.log(a.min_rows), np.log(a.max_rows), size=a.clusters)).astype(int)
        cols = rng.integers(600, 1200, size=a.clusters)
        t = {"generate_s": 0.0, "core_s": 0.0, "pairing_s": 0.0}
        n_primers = n_pairs = rows_total = 0
        phases = {}
        from multiprime_amd.pairing import Primers_filter
        for i in range(rank, a.clusters, world):
            t0 = time.time()
            fa = os.path.join(wd, f"Cluster_{i}.tfa")
            with open(fa, "wb") as f:
                f.write(to_fasta(synth_block(0, int(sizes[i]), int(cols[i]), a.seed + 1000 * (i + 1))))
            t["generate_s"] += time.time() - t0
            rows_total += int(sizes[i])
            t0 = time.time()
            top = os.path.join(wd, f"Cluster_{i}.top.primer.out")
            deep = sizes[i] > 2000                      # the JSON side files are O(windows x sequences): device-resident bitsets instead
            app = NN_degenerate(seq_file=fa, primer_length=18, coverage=0.7, number_of_dege_bases=4, score_of_dege_bases=10,
                          raw_entropy_threshold=3.6, product_len=150, position="2,3,-1", variation=1, distance=4,
                          GC="0.2,0.7", nproc=1, outfile=top, device=local, write_json=not deep, keep_bitsets=deep)
            app.run()
            t["core_s"] += time.time() - t0
            for key, val in app.stats.items():
                if isinstance(val, float):
                    phases["core." + key] = phases.get("core." + key, 0.0) + val
            n_primers += sum(1 for _ in open(top)) - 1
            t0 = time.time()
            cand = os.path.join(wd, f"Cluster_{i}.candidate.primers.txt")
            with contextlib.redirect_stdout(io.StringIO()):
                pf = Primers_filter(ref_file=fa, primer_file=top, outfile=cand, adaptor=ADAPTOR, rep_seq_number=0, distance=4,
                               size="150,1200", position=4, fraction=0.7, diff_Tm=4, core=app if deep else None)
                pf.run()
            t["pairing_s"] += time.time() - t0
            for key, val in pf.stats.items():
                if isinstance(val, float):
                    phases["pairing." + key] = phases.get("pairing." + key, 0.0) + val
            app.ctx.close()
            if os.path.exists(cand):
                n_pairs += sum(max(0, len(line.rstrip("\n").split("\t")) - 1) for line in open(cand))
        if world > 1:
            dist.barrier()
        res = {"clusters": a.clusters, "n_gpus": world, "rows_this_rank": rows_total, "primers_this_rank": n_primers,
               "pairs_this_rank": n_pairs, **{k: round(v, 2) for k, v in t.items()},
               "phase_sums": {k: round(v, 2) for k, v in sorted(phases.items()) if v >= 0.05}}
        if rank == 0:
            t0 = time.time()
            agg = os.path.join(wd, "candidate_primers_sets.txt")
            with open(agg, "wb") as out:                                      # rule aggregate_candidate_primers: cat
                for i in range(a.clusters):
                    p = os.path.join(wd, f"Cluster_{i}.candidate.primers.txt")
                    if os.path.exists(p):
                        out.write(open(p, "rb").read())
            final = os.path.join(wd, "final_maxprimers_set.xls")
            from multiprime_amd import maxset
            with contextlib.redirect_stdout(io.StringIO()):
                try:
                    maxset.main(["-i", agg, "-s", "5", "-m", "T", "-o", final])
                except SystemExit as e:                                        # the reference exits 1 when it cannot back-track
                    res["maxset_exit"] = e.code
            res["maxset_s"] = round(time.time() - t0, 2)
            if os.path.exists(final):
                t0 = time.time()
                fa = os.path.join(wd, "final_maxprimers_set.fa")              # primerset_format.py: name_F / name_R records
                n_set = 0
                with open(final) as In, open(fa, "w") as out:
                    for line in In:
                        if line.startswith("#"):
                            continue
                        info = line.strip().split("/")[-1].replace(".candidate.primers.txt", "").split("\t")
                        out.write(f">{info[0]}_F\n{info[2]}\n>{info[0]}_R\n{info[3]}\n")
                        n_set += 1
                from multiprime_amd.dimer import Dimer
                with contextlib.redirect_stdout(io.StringIO()):
                    Dimer(primer_file=fa, threshold=3.96, outfile=fa + ".findimer", nproc=1, device=local).run()
                res["findimer_s"] = round(time.time() - t0, 2)
                res["final_set_pairs"] = n_set
                res["dimer_hits"] = sum(1 for _ in open(fa + ".findimer"))
            print(json.dumps(res), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
