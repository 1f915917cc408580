#!/usr/bin/env python3
"""BASELINE config 5 in one process per GPU: many clusters through the whole rule chain with this build's drop-ins.

For each of C synthetic clusters (different roots and seeds, 500 .. 50 000 sequences; SURVEY §8d input 5):
  rule multiPrime      -> multiprime_amd.core.NN_degenerate          ({i}.top.primer.out)            multiPrime.py:200-207
  rule get_multiPrime  -> multiprime_amd.pairing.Primers_filter      ({i}.candidate.primers.txt)     multiPrime.py:232-238
then `cat` (rule aggregate_candidate_primers), rule get_Maxprimerset -> multiprime_amd.maxset (multiPrime.py:277-295), the
primerset_format shim and rule all_mfeprimer_check's finDimer -> multiprime_amd.dimer (multiPrime.py:396-415).  Flags are
multiPrime.yaml's.  Clusters are independent: with WORLD_SIZE > 1 rank r takes clusters r, r + N, ... (no data-path collective,
one barrier before rank 0 aggregates).  Prints one JSON line with the per-stage wall times.

  python tools/multi_cluster.py --clusters 64
  python -m torch.distributed.run --nproc-per-node 8 tools/multi_cluster.py --clusters 64
  python tools/multi_cluster.py --clusters 16 --max-rows 5000 --check      # parity: the chain twice, HIP library and CPU oracle

--check (test infrastructure: the one place outside tests/ that loads the oracle, like bench.py's cpu_baseline leg) runs the
whole chain a second time with the plain-C oracle behind the same ABI and requires every file of the chain to be identical:
{i}.top.primer.out (+ its JSON side files where they are written), {i}.candidate.primers.txt/.xls/.fa, candidate_primers_sets.txt,
sort.candidate_primers_sets.txt, final_maxprimers_set.xls, .next.xls, .fa, .fa.findimer, .fa.findimer.dimer_num.
tests/test_chain.py holds the same comparison against files recorded from the unmodified reference chain.
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
from multiprime_amd._abi import prefer_staged_copies  # noqa: E402
prefer_staged_copies()            # a program's own decision, before the HIP runtime starts (multiprime_amd/_abi.py)
from multiprime_amd.core import NN_degenerate  # noqa: E402
from multiprime_amd.synth import synth_block, to_fasta  # noqa: E402

ADAPTOR = "TCTTTCCCTACACGACGCTCTTCCGATCT,TGGAGTTCAGACGTGTGCTCTTCCGATCT"
DEEP_ROWS = 2000        # above this the JSON side files (O(windows x sequences)) give way to device-resident bitsets


def cluster_stage(fa, wd, name, library=None, device=0, deep=False, timings=None, phases=None, primer_length=18):
    """Rules multiPrime and get_multiPrime for one cluster: {name}.top.primer.out and {name}.candidate.primers.txt in wd.
    Returns (primers written, candidate pairs written)."""
    from multiprime_amd.pairing import Primers_filter
    t0 = time.time()
    top = os.path.join(wd, name + ".top.primer.out")
    app = NN_degenerate(seq_file=fa, primer_length=primer_length, coverage=0.7, number_of_dege_bases=4, score_of_dege_bases=10,
                        raw_entropy_threshold=3.6, product_len=150, position="2,3,-1", variation=1, distance=4,
                        GC="0.2,0.7", nproc=1, outfile=top, device=device, library=library, write_json=not deep, keep_bitsets=deep)
    app.run()
    if timings is not None:
        timings["core_s"] += time.time() - t0
        for key, val in app.stats.items():
            if isinstance(val, float):
                phases["core." + key] = phases.get("core." + key, 0.0) + val
    n_primers = sum(1 for _ in open(top)) - 1
    t0 = time.time()
    cand = os.path.join(wd, name + ".candidate.primers.txt")
    with contextlib.redirect_stdout(io.StringIO()):
        pf = Primers_filter(ref_file=fa, primer_file=top, outfile=cand, adaptor=ADAPTOR, rep_seq_number=0, distance=4,
                            size="150,1200", position=4, fraction=0.7, diff_Tm=4, core=app if deep else None,
                            library=library, device=device)
        try:
            pf.run()
        except SystemExit:              # an empty primer file: the reference's pairing step exits 1 there too and writes nothing
            pass
    if timings is not None:
        timings["pairing_s"] += time.time() - t0
        for key, val in pf.stats.items():
            if isinstance(val, float):
                phases["pairing." + key] = phases.get("pairing." + key, 0.0) + val
    if not deep:
        pf.ctx.close()
    app.ctx.close()
    n_pairs = 0
    if os.path.exists(cand):
        n_pairs = sum(max(0, len(line.rstrip("\n").split("\t")) - 1) for line in open(cand)) // 5
    return n_primers, n_pairs


def tail_stage(wd, names, library=None, device=0):
    """Rules aggregate_candidate_primers, get_Maxprimerset (-s 5 -m T), primerset_format and finDimer over the clusters' candidate
    files in wd.  Returns the stage record (times, exit status of the set cover, sizes)."""
    import types
    from multiprime_amd import maxset
    from multiprime_amd.dimer import Dimer
    res = {}
    t0 = time.time()
    agg = os.path.join(wd, "candidate_primers_sets.txt")
    with open(agg, "wb") as out:                                      # rule aggregate_candidate_primers: cat
        for name in names:
            p = os.path.join(wd, name + ".candidate.primers.txt")
            if os.path.exists(p):
                out.write(open(p, "rb").read())
    final = os.path.join(wd, "final_maxprimers_set.xls")
    with contextlib.redirect_stdout(io.StringIO()):
        try:
            maxset.run(types.SimpleNamespace(input=agg, step=5, method="T", out=final, device=device), library=library)
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
        with contextlib.redirect_stdout(io.StringIO()):
            Dimer(primer_file=fa, threshold=3.96, outfile=fa + ".findimer", nproc=1, device=device, library=library).run()
        res["findimer_s"] = round(time.time() - t0, 2)
        res["final_set_pairs"] = n_set
        res["dimer_hits"] = sum(1 for _ in open(fa + ".findimer")) - 1
    return res


def chain_files(wd):
    """{file name: bytes} of everything the chain wrote into wd (inputs excluded), the directory's own path replaced by @WD
    (the candidate files carry their absolute path in column 1, get_multiPrime_V8.py:606)."""
    out = {}
    for fn in sorted(os.listdir(wd)):
        p = os.path.join(wd, fn)
        if fn.endswith((".tfa", ".npz")) or os.path.isdir(p):
            continue
        out[fn] = open(p, "rb").read().replace(os.path.abspath(wd).encode(), b"@WD")
    return out


def run_chain(wd, fastas, library=None, device=0, deep_rows=DEEP_ROWS, primer_length=18):
    """The whole chain over {cluster name: (FASTA path, rows)} into wd; returns chain_files(wd)."""
    os.makedirs(wd, exist_ok=True)
    for name, (fa, rows) in fastas.items():
        cluster_stage(fa, wd, name, library=library, device=device, deep=rows > deep_rows, primer_length=primer_length)
    tail_stage(wd, list(fastas), library=library, device=device)
    return chain_files(wd)


def compare_chains(a, b):
    """Names of the files that differ between two chain_files() results (missing on one side counts)."""
    return [fn for fn in sorted(set(a) | set(b)) if a.get(fn) != b.get(fn)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clusters", type=int, default=64)
    ap.add_argument("--min-rows", type=int, default=500)
    ap.add_argument("--max-rows", type=int, default=50000)
    ap.add_argument("--seed", type=int, default=20250303)
    ap.add_argument("--workdir", default=None, help="shared directory (default: a temporary one; required for N > 1)")
    ap.add_argument("--check", action="store_true", help="run the chain a second time on the CPU oracle and compare every file")
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
        names = [f"Cluster_{i}" for i in range(a.clusters)]
        fastas = {}
        for i in range(rank, a.clusters, world):
            t0 = time.time()
            fa = os.path.join(wd, f"Cluster_{i}.tfa")
            with open(fa, "wb") as f:
                f.write(to_fasta(synth_block(0, int(sizes[i]), int(cols[i]), a.seed + 1000 * (i + 1))))
            fastas[names[i]] = (fa, int(sizes[i]))
            t["generate_s"] += time.time() - t0
            rows_total += int(sizes[i])
            p, q = cluster_stage(fa, wd, names[i], device=local, deep=sizes[i] > DEEP_ROWS, timings=t, phases=phases)
            n_primers += p
            n_pairs += q
        if world > 1:
            dist.barrier()
        res = {"clusters": a.clusters, "n_gpus": world, "rows_this_rank": rows_total, "primers_this_rank": n_primers,
               "pairs_this_rank": n_pairs, **{k: round(v, 2) for k, v in t.items()},
               "phase_sums": {k: round(v, 2) for k, v in sorted(phases.items()) if v >= 0.05}}
        if rank == 0:
            res.update(tail_stage(wd, names, device=local))
        if a.check:
            # the same clusters through the plain-C oracle (every rank its own; rank 0 the tail), file by file
            from multiprime_amd._abi import Library
            ora = Library(os.path.join(REPO, "oracle", "_build", "libmprime_oracle.so"))
            t0 = time.time()
            od = os.path.join(wd, "oracle")
            os.makedirs(od, exist_ok=True)
            for name, (fa, rows) in fastas.items():
                cluster_stage(fa, od, name, library=ora, deep=rows > DEEP_ROWS)
            if world > 1:
                dist.barrier()
            if rank == 0:
                tail_stage(od, names, library=ora)
                mine, theirs = chain_files(wd), chain_files(od)
                diff = compare_chains(mine, theirs)
                res["check"] = {"files_compared": len(set(mine) | set(theirs)), "different": diff, "identical": not diff,
                                "oracle_s": round(time.time() - t0, 2)}
        if rank == 0:
            print(json.dumps(res), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if a.check and rank == 0 and res["check"]["different"]:
            raise SystemExit(1)


if __name__ == "__main__":
    main()
