"""Command line of the core step — the exact flag set of scripts/multiPrime-core.py
(parseArg, V20:60-102), so Snakemake rule `multiPrime` (multiPrime.py:200-207) can call this
program instead of the reference script without touching the rule:

    python {scripts_dir}/multiPrime-core.py -i {tmsa} -n 4 -d 10 -v 1 -c 2,3,-1 -g 0.2,0.7 -s 150 \\
           -l 18 -e 3.6 -o {out} -f 0.7 -p 1

Extra, optional flags (additions must stay optional):
  --device D     GPU ordinal (default 0).  With several ranks it is the ordinal of LOCAL_RANK 0: rank r of the node opens D + r
                 (MP_SHARE_DEVICE=1: every rank opens D — several ranks on one GPU, for tests)
  --no-json      skip the two O(windows x sequences) JSON side files, which stop being writable at ~10^5 sequences
  --bitsets      also write <out>.coverage_bitsets.npz (the bitset form of those files)
  --ngpu N       run on N GPUs of this node: the program re-launches itself as N ranks (torch.distributed.run, 127.0.0.1).
                 The same happens without the flag when a launcher started it (WORLD_SIZE > 1).  What the ranks share:
                   one alignment (-i / -o):  its ROWS — contiguous blocks, one all-reduce of the coverage counters over
                                             RCCL (multiprime_amd/dist.py); rank 0 writes the files;
                   --batch:                  the CLUSTERS — rank r takes lines r, r + N, ... of the batch file, no collective.
  --batch FILE   many alignments in ONE process per GPU: FILE holds one `input<TAB>output` pair per line, every pair runs
                 with this command's flags.  The fixed cost of a process (interpreter, numpy, HIP runtime: 0.3-0.4 s, more
                 than the work of a 500-sequence cluster) is paid once instead of once per cluster; a Snakemake workflow
                 replaces the per-cluster rule by one rule over the cluster list.  Prints the reference's closing line per
                 pair and one JSON summary (clusters per second) at the end.  --batch-workers T: clusters in flight per GPU
                 (threads with a context each; default up to 8): a small cluster is host-bound, several of them overlap.
                 --batch-procs P: P processes per GPU share its clusters (the per-cluster Python does not scale over threads).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import threading
import subprocess
import sys
import time


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="For degenerate primer design (MI355X-native core step)")
    p.add_argument("-i", "--input", type=str, default=None, metavar="<file>",
                   help="Input file: multi-alignment output (muscle or others).")
    p.add_argument("-l", "--plen", type=int, default=18, metavar="<int>", help="Length of primer. Default: 18.")
    p.add_argument("-n", "--dnum", type=int, default=4, metavar="<int>", help="Max number of degenerate. Default: 4.")
    p.add_argument("-d", "--degeneracy", type=int, default=10, metavar="<int>", help="Max degeneracy of primer. Default: 10.")
    p.add_argument("-v", "--variation", type=int, default=1, metavar="<int>", help="Max mismatch number of primer. Default: 1")
    p.add_argument("-e", "--entropy", type=float, default=3.6, metavar="<float>",
                   help="Entropy threshold of a primer-length window. Default: 3.6.")
    p.add_argument("-g", "--gc", type=str, default="0.2,0.7", metavar="<str>", help="Filter primers by GC content. Default [0.2,0.7].")
    p.add_argument("-s", "--size", type=int, default=100, metavar="<int>", help="Filter primers by mini PRODUCT size. Default 100.")
    p.add_argument("-f", "--fraction", type=float, default=0.8, metavar="<float>", help="Filter primers by match fraction. Default: 0.8.")
    p.add_argument("-c", "--coordinate", type=str, default="1,2,-1", metavar="<str>",
                   help="Positions where a mismatch is not tolerated (>0: from the 5' end, <0: from the 3' end). Default: 1,2,-1.")
    p.add_argument("-p", "--proc", type=int, default=20, metavar="<int>",
                   help="Accepted for compatibility (the reference's process pool is inert; the work runs on the GPU).")
    p.add_argument("-a", "--away", type=int, default=4, metavar="<int>", help="Hairpin: minimal distance of paired bases. Default: 4.")
    p.add_argument("-o", "--out", type=str, default=None, metavar="<file>", help="output file")
    p.add_argument("--device", type=int, default=None, help="GPU ordinal (default 0); with several ranks: the ordinal of local rank 0, rank r opens D + r")
    p.add_argument("--grid", default=None, metavar="RxG|auto",
                   help="with several ranks: R row shards x G window groups instead of row shards only (R x G = ranks); auto = as many "
                        "window groups as fit the device memory")
    p.add_argument("--ngpu", type=int, default=1, help="GPUs of this node to use: re-launches this command as that many ranks")
    p.add_argument("--batch", type=str, default=None, metavar="<file>",
                   help="file of `input<TAB>output` lines: all of them in one process per GPU with this command's flags")
    p.add_argument("--batch-workers", type=int, default=0, metavar="<int>",
                   help="--batch: clusters in flight per GPU (worker threads, a context each); default: up to 8")
    p.add_argument("--batch-procs", type=int, default=1, metavar="<int>",
                   help="--batch: processes per GPU that share its clusters (each with its own worker threads); default 1")
    p.add_argument("--batch-part", type=str, default=None, help=argparse.SUPPRESS)       # "i/n": set by --batch-procs for its children
    p.add_argument("--no-json", action="store_true", help="do not write the two *_seq_id_json side files")
    p.add_argument("--bitsets", action="store_true",
                   help="also write <out>.coverage_bitsets.npz: per window, one bit per sequence the primer does not "
                        "reach (the bitset form of the JSON files; scripts/get_multiPrime.py reads it when the JSON is absent)")
    p.add_argument("--stats", action="store_true", help="print per-phase timings to stderr")
    args = p.parse_args(argv)
    if args.batch is None and (args.input is None or args.out is None):
        p.error("the following arguments are required: -i/--input, -o/--out (or --batch)")      # exit code 2, like the reference's parser
    if args.ngpu < 1:
        p.error("--ngpu must be at least 1")
    return args


def _launcher_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def _respawn(n, argv):
    """This same command as n ranks on this node (one process per GPU), rendezvous on 127.0.0.1."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rest, skip = [], False
    for tok in (sys.argv[1:] if argv is None else list(argv)):
        if skip:
            skip = False
        elif tok == "--ngpu":
            skip = True
        elif not tok.startswith("--ngpu="):
            rest.append(tok)
    script = _script_and_args(argv)[0]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + rest
    return subprocess.call(cmd)


def _core(args, inp, out, device, comm, context=None, grid=None):
    from .core import NN_degenerate
    return NN_degenerate(seq_file=inp, primer_length=args.plen, coverage=args.fraction,
                         number_of_dege_bases=args.dnum, score_of_dege_bases=args.degeneracy,
                         raw_entropy_threshold=args.entropy, product_len=args.size, position=args.coordinate,
                         variation=args.variation, distance=args.away, GC=args.gc, nproc=args.proc, outfile=out,
                         device=device, comm=comm, write_json=not args.no_json, write_bitsets=args.bitsets, context=context, grid=grid)


def _closing_line(e1, e2):
    # same closing line as the reference (V20:1193-1198): it lands in the Snakemake rule's log
    print("INFO {} Total times: {}".format(time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(time.time())),
                                           round(float(e2 - e1), 2)), flush=True)


def _script_and_args(argv):
    """(script, arguments) of this command as a child process would be started: the process's own command line when main() was
    called without arguments, else the drop-in script with the arguments main() was given (a host program calling main([...]))."""
    if argv is None:
        return os.path.abspath(sys.argv[0]), list(sys.argv[1:])
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "multiPrime-core.py"), list(argv)


def _run_batch(args, rank, world, device, argv=None):
    pairs = []
    with open(args.batch) as f:
        for line in f:
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            if len(tok) != 2:
                raise SystemExit(f"{args.batch}: expected `input<TAB>output`, got: {line.rstrip()}")
            pairs.append(tok)
    t0 = time.time()
    mine = pairs[rank::world]
    if args.batch_part:                                  # a child of --batch-procs: its share of this rank's clusters
        i, n = (int(x) for x in args.batch_part.split("/"))
        mine = mine[i::n]
    elif args.batch_procs > 1 and len(mine) > 1:
        # Several PROCESSES on this GPU: the per-cluster Python (filters, TSV rows, array plumbing) holds the interpreter lock, so
        # threads stop scaling at ~60 clusters/s; processes do not share it.  Each child pays its own start-up (0.3-0.4 s, in parallel).
        n = min(args.batch_procs, len(mine))
        script, rest = _script_and_args(argv)
        kids = [subprocess.Popen([sys.executable, script] + rest + ["--batch-part", f"{i}/{n}", "--device", str(device)], stdout=subprocess.PIPE, text=True,
                                 env=dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MP_SHARE_DEVICE="1"))
                for i in range(n)]
        # a child's lines pass through as they are, except its own summary (tagged "part"): the parent prints ONE summary for the rank
        sequences, lock = [0], threading.Lock()

        def relay(kid):
            for line in kid.stdout:
                if line.startswith('{"batch"') and '"part"' in line:
                    with lock:
                        sequences[0] += json.loads(line).get("sequences", 0)
                else:
                    sys.stdout.write(line)
                    sys.stdout.flush()

        th = [threading.Thread(target=relay, args=(k,)) for k in kids]
        for t in th:
            t.start()
        for t in th:
            t.join()
        codes = [k.wait() for k in kids]
        dt = time.time() - t0
        print(json.dumps({"batch": args.batch, "rank": rank, "n_ranks": world, "clusters": len(mine), "sequences": sequences[0], "processes": n,
                          "seconds": round(dt, 3), "clusters_per_s": round(len(mine) / dt, 2) if dt > 0 else None}), flush=True)
        if any(codes):
            raise SystemExit(next(c for c in codes if c))
        return
    # Several clusters in flight on this GPU: worker threads, each with a context (stream) of its own.  A 500-sequence
    # cluster is ~1 ms of device work inside ~20 ms of host work (FASTA, planning, the O(windows x sequences) side files), most of it
    # native code that releases the interpreter lock — so the clusters overlap instead of queueing behind one another.
    workers = max(1, min(args.batch_workers if args.batch_workers > 0 else min(8, (os.cpu_count() or 1) // 8 or 1), len(mine) or 1))
    rows, lock, errors = [0], threading.Lock(), []
    todo = iter(mine)

    def work():
        ctx = None                                       # one context per worker, kept across its clusters
        try:
            while True:
                with lock:
                    pair = next(todo, None)
                if pair is None or errors:
                    return
                inp, out = pair
                try:
                    e1 = time.time()
                    app = _core(args, inp, out, device, None, context=ctx)
                    ctx = app.ctx
                    app.run()
                    with lock:
                        rows[0] += app.total_sequence_number
                        _closing_line(e1, time.time())
                except BaseException as e:               # SystemExit of a refused alignment included: reported, the batch stops
                    with lock:
                        errors.append((inp, e))
                    return
        finally:
            if ctx is not None:
                ctx.close()

    if workers == 1:
        work()
    else:
        th = [threading.Thread(target=work) for _ in range(workers)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    if errors:
        inp, e = errors[0]
        if isinstance(e, SystemExit):
            raise SystemExit(e.code)
        raise RuntimeError(f"{inp}: {e}") from e
    dt = time.time() - t0
    summary = {"batch": args.batch, "rank": rank, "n_ranks": world, "clusters": len(mine), "sequences": rows[0], "workers": workers,
               "seconds": round(dt, 3), "clusters_per_s": round(len(mine) / dt, 2) if dt > 0 else None}
    if args.batch_part:
        summary["part"] = args.batch_part                # a child of --batch-procs: the parent folds this line into its own
    print(json.dumps(summary), flush=True)


def main(argv=None):
    from ._abi import prefer_staged_copies
    prefer_staged_copies()                      # a command line owns its process: see _abi.prefer_staged_copies
    args = parse_args(argv)
    rank, world, local = _launcher_env()
    if args.ngpu > 1 and world == 1:
        sys.exit(_respawn(args.ngpu, argv))
    # several ranks must not open the same GPU (RCCL refuses duplicate devices): --device is then the first rank's ordinal
    base = args.device if args.device is not None else 0
    device = base if world == 1 or os.environ.get("MP_SHARE_DEVICE") == "1" else base + local
    if args.batch is not None:
        _run_batch(args, rank, world, device, argv)     # clusters are independent: no process group, no collective
        return
    comm = grid = None
    if world > 1:                                       # one alignment on several GPUs: its rows are sharded
        import torch
        import torch.distributed as dist
        from .dist import RowShards
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
        comm = RowShards()
        # --grid RxG: R row shards x G window groups (dist.ShardGrid); "auto": as many window groups as the device memory allows —
        # a window group of one row shard has no collective at all
        if args.grid:
            from .dist import ShardGrid
            if args.grid == "auto":                            # the file's bytes stand for rows x columns
                shape = ShardGrid.best_shape(world, os.path.getsize(args.input), 1)
            else:
                shape = ShardGrid.parse(args.grid, world)
            if shape[1] > 1:
                grid, comm = ShardGrid(*shape), None
    e1 = time.time()
    try:
        app = _core(args, args.input, args.out, device, comm, grid=grid)
        app.run()
    finally:
        if comm is not None:
            import torch.distributed as dist
            dist.destroy_process_group()
    e2 = time.time()
    if args.stats and rank == 0:
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in app.stats.items()}), file=sys.stderr)
    if rank == 0:
        _closing_line(e1, e2)


if __name__ == "__main__":
    main()
