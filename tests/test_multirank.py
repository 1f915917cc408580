"""The N>1 path on CPU: world_size 2, 3, 4 and 8 over gloo (even and uneven row splits), rows sharded across ranks, the CPU oracle
standing in for the GPU behind the same C ABI.  Results must be identical to the single-process
run, i.e. to the reference's own outputs."""
import json
import os
import sys

import pytest
import torch.multiprocessing as mp

from conftest import GOLDEN, REPO, golden_input, load_gz_json


def _worker(rank, world, port, name, inp, out, lib_path, write_json=True):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from multiprime_amd._abi import Library
    from multiprime_amd.core import NN_degenerate
    from multiprime_amd.dist import RowShards
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gzip
        meta = json.loads(gzip.open(os.path.join(GOLDEN, name + ".trace.json.gz")).read())["meta"]
        fl = meta["flags"]
        app = NN_degenerate(seq_file=inp, primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"],
                            score_of_dege_bases=fl["d"], raw_entropy_threshold=fl["e"], product_len=fl["s"],
                            position=fl["c"], variation=fl["v"], distance=fl["a"], GC=fl["g"], nproc=1,
                            outfile=out, library=Library(lib_path), comm=RowShards(), write_bitsets=True, write_json=write_json)
        app.run()
        assert app._win_split == (not write_json)
        check_traffic(app, world)
    finally:
        dist.destroy_process_group()


def check_traffic(app, world):
    """Window-split planning: the histogram entries and the exception list reach the rank that plans the window by a personalised
    exchange — a rank receives the entries of ITS windows (about 1 / world of the others' tables; the all-gather delivered all of
    them); every table moves in ONE collective and the small ones travel together.  Replicated planning (JSON side files) keeps the
    all-gathers.  (tests/test_comm_ranks.py makes the 1.2 x statement on the evenly spread synthetic alignment, through RCCL's ABI.)"""
    t = app.comm.traffic
    kinds = [what for what, _, _ in t]
    if not app._win_split:
        assert "histogram entries" not in kinds and kinds.count("all-gather") >= 4
        return
    assert kinds.count("histogram entries") == 1 and kinds.count("exception list") == 1, kinds
    assert kinds.count("all-reduce") <= 3, kinds             # the region's two row histograms, the two statistics tables together (+ the counters')
    import numpy as np
    for what in ("histogram entries", "exception list"):
        (sent, received), = [(s, r) for w, s, r in t if w == what]
        own = np.asarray([app.comm.tables[what]], np.int64)
        others = int(app.comm.gather_var(own).sum()) - int(own[0])      # what an all-gather of the table would have delivered here
        # an even spread delivers others / world; real alignments are skewed (variable regions hold more distinct k-mers)
        assert received <= (0.75 if world == 2 else 0.6) * others + 4096, (what, sent, received, others)
        assert sent <= int(own[0])
    # everything else: the row attributes of the region, two small gathers per fused group (candidates, results, bitsets)
    assert kinds.count("all-gather") <= 14, kinds


# 1/2/4/8 shards incl. uneven splits: syn_iupac has 60 rows (8 ranks: 7 or 8 rows each), ivc_v1 166 (4 ranks: 41/42)
@pytest.mark.parametrize("name,world", [("syn_iupac", 2), ("syn_ragged", 3), ("ivc_v1", 2), ("msa1000_k18_d64", 2),
                                        ("syn_iupac", 8), ("ivc_v1", 4), ("syn_edge", 4), ("msa1000_k18_d64", 8),
                                        ("syn_iupac_k33", 3), ("syn_ragged_k40", 2), ("syn_edge_k63", 4)])      # 64-bit window words
def test_sharded_run_matches_reference(name, world, oracle_lib, tmp_path):
    from test_core_golden import check_outputs
    meta = load_gz_json(name + ".trace.json.gz")["meta"]
    inp = tmp_path / (name + ".fa")
    inp.write_bytes(golden_input(meta["input"]))
    out = tmp_path / (name + ".out")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, name, str(inp), str(out), oracle_lib.path), nprocs=world, join=True)
    check_outputs(name, out)
    # the sharded coverage bitsets (gathered bit by bit across ranks) agree with the reference's JSON files
    import numpy as np
    z = np.load(str(out) + ".coverage_bitsets.npz")
    from multiprime_amd.core import bitset_ids
    ids = bitset_ids(z)
    noncov, gap = load_gz_json(name + ".noncov.json.gz"), load_gz_json(name + ".gap.json.gz")
    for i, pos in enumerate(z["positions"].tolist()):
        g = {x for lst in gap[str(pos)].values() for x in lst}
        for side, arr in ((0, z["not_f"]), (1, z["not_r"])):
            want = g | {x for lst in noncov[str(pos)][side].values() for x in lst}
            bits = np.unpackbits(arr[i].view(np.uint8), bitorder="little")[: len(ids)]
            assert {ids[r] for r in np.nonzero(bits)[0]} == want, (pos, side)


def _check_bitsets(name, out):
    import numpy as np
    from multiprime_amd.core import bitset_ids
    z = np.load(str(out) + ".coverage_bitsets.npz")
    ids = bitset_ids(z)
    noncov, gap = load_gz_json(name + ".noncov.json.gz"), load_gz_json(name + ".gap.json.gz")
    assert len(z["positions"]) == len(noncov)
    for i, pos in enumerate(z["positions"].tolist()):
        g = {x for lst in gap[str(pos)].values() for x in lst}
        for side, arr in ((0, z["not_f"]), (1, z["not_r"])):
            want = g | {x for lst in noncov[str(pos)][side].values() for x in lst}
            bits = np.unpackbits(arr[i].view(np.uint8), bitorder="little")[: len(ids)]
            assert {ids[r] for r in np.nonzero(bits)[0]} == want, (pos, side)


# without the JSON side files the planning is split by windows across the ranks (candidates gathered, results concatenated)
@pytest.mark.parametrize("name,world", [("syn_iupac", 2), ("syn_ragged", 3), ("ivc_v1", 4), ("msa1000_k18_d64", 8), ("syn_edge", 5),
                                        ("cluster0_v2", 3), ("syn_v2_k50", 3), ("ivc_k45_v2", 2)])
def test_sharded_run_with_window_split_planning(name, world, oracle_lib, tmp_path):
    meta = load_gz_json(name + ".trace.json.gz")["meta"]
    inp = tmp_path / (name + ".fa")
    inp.write_bytes(golden_input(meta["input"]))
    out = tmp_path / (name + ".out")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, name, str(inp), str(out), oracle_lib.path, False), nprocs=world, join=True)
    with open(os.path.join(GOLDEN, name + ".tsv"), "rb") as f:
        assert out.read_bytes() == f.read(), "TSV differs from the reference's"
    assert not os.path.exists(str(out) + ".gap_seq_id_json")
    _check_bitsets(name, out)


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["library", "torch"])
def test_rccl_path_single_rank(transport, tmp_path, monkeypatch):
    """The device branch of RowShards with a world of one GPU: the multi-GPU code path the driver's 8-GPU run takes, minus the
    other ranks.  "library": the collectives are the C ABI's (mp_comm_*, csrc/comm.hip) — a real one-rank RCCL communicator
    (MP_COMM_FORCE_RCCL), the counters' all-reduce queued behind the evaluation kernel inside the library.  "torch": the same
    exchanges through torch.distributed's RCCL backend (MP_NATIVE_COMM=0)."""
    if transport == "library":
        monkeypatch.setenv("MP_COMM_FORCE_RCCL", "1")
    else:
        monkeypatch.setenv("MP_NATIVE_COMM", "0")
    import torch
    import torch.distributed as dist
    from multiprime_amd._abi import Library
    from multiprime_amd.core import NN_degenerate
    from multiprime_amd.dist import RowShards
    from test_core_golden import check_outputs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29600 + os.getpid() % 1000)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for name in ("syn_iupac", "msa1000_k18_d64"):
            meta = load_gz_json(name + ".trace.json.gz")["meta"]
            fl = meta["flags"]
            inp = tmp_path / (name + ".fa")
            inp.write_bytes(golden_input(meta["input"]))
            out = tmp_path / (name + ".out")
            comm = RowShards()
            assert comm.on_gpu
            app1 = NN_degenerate(seq_file=str(inp), primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"],
                          score_of_dege_bases=fl["d"], raw_entropy_threshold=fl["e"], product_len=fl["s"], position=fl["c"],
                          variation=fl["v"], distance=fl["a"], GC=fl["g"], nproc=1, outfile=str(out), library=Library(),
                          comm=comm)
            assert (comm.native is not None) == (transport == "library")
            app1.run()
            check_outputs(name, out)
            # the window-split planning path (no JSON side files): candidates and results travel through RCCL all_gathers
            out2 = tmp_path / (name + ".nojson.out")
            app = NN_degenerate(seq_file=str(inp), primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"],
                                score_of_dege_bases=fl["d"], raw_entropy_threshold=fl["e"], product_len=fl["s"], position=fl["c"],
                                variation=fl["v"], distance=fl["a"], GC=fl["g"], nproc=1, outfile=str(out2), library=Library(),
                                comm=RowShards(), write_json=False)
            app.run()
            assert app._win_split and out2.read_bytes() == out.read_bytes()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_comm_exports_on_one_gpu(hip_lib, monkeypatch):
    """mprime.h section 9 through a one-rank RCCL communicator: sums, gathers and the fused evaluate + all-reduce."""
    import numpy as np
    from test_hip_parity import fuzz_msa
    monkeypatch.setenv("MP_COMM_FORCE_RCCL", "1")
    ctx = hip_lib.context(0)
    with pytest.raises(Exception):
        ctx.comm_sum(np.arange(4))                                   # no communicator yet
    uid = ctx.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.comm_init(1, 0, uid)
    a = np.arange(1000, dtype=np.int64).reshape(10, 100) - 37
    assert np.array_equal(ctx.comm_sum(a), a)
    blob = np.frombuffer(bytes(range(256)) * 5, np.uint8)
    got, counts = ctx.comm_gather_bytes(blob, 1)
    assert counts.tolist() == [blob.size] and got.tobytes() == blob.tobytes()
    got, counts = ctx.comm_gather_bytes(np.zeros(0, np.uint8), 1)
    assert counts.tolist() == [0] and got.size == 0
    data, off, _ = fuzz_msa(5, 3000, 150, ragged=False, p_gap=0.02, p_iupac=0.0)
    ctx.load_msa(data, off)
    k, W = 18, 100
    ctx.build_windows(3, W, k, 1)
    rng = np.random.default_rng(2)
    cw = np.repeat(np.arange(W, dtype=np.int32), 2)
    codes = rng.integers(1, 16, size=(len(cw), k)).astype(np.uint8)
    assert np.array_equal(ctx.eval_candidates_allreduce(cw, codes, 12, 3 << 14), ctx.eval_candidates(cw, codes, 12, 3 << 14))
    ctx.comm_destroy()
    with pytest.raises(Exception):
        ctx.comm_sum(np.arange(4))


def test_comm_exports_of_the_checker_are_a_world_of_one(oracle_lib):
    import numpy as np
    ctx = oracle_lib.context(0)
    with pytest.raises(Exception):
        ctx.comm_init(2, 0, bytes(128))                              # the ABI checker has no collectives
    ctx.comm_init(1, 0, None)
    assert ctx.comm_sum(np.arange(5)).tolist() == [0, 1, 2, 3, 4]
    got, counts = ctx.comm_gather_bytes(np.arange(7, dtype=np.uint8), 1)
    assert got.tolist() == list(range(7)) and counts.tolist() == [7]


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("syn_iupac", 2), ("msa1000_k18_d64", 2), ("syn_ragged", 3), ("syn_ragged_k40", 2), ("syn_iupac_k33", 3)])
def test_sharded_hip_contexts_match_reference(name, world, hip_lib, tmp_path):
    """Two ranks, each with its own HIP context on the one GPU of the box (gloo carries the collectives): row
    offsets, histogram merging and counter all-reduce on top of the real kernels."""
    from test_core_golden import check_outputs
    meta = load_gz_json(name + ".trace.json.gz")["meta"]
    inp = tmp_path / (name + ".fa")
    inp.write_bytes(golden_input(meta["input"]))
    out = tmp_path / (name + ".out")
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, name, str(inp), str(out), hip_lib.path), nprocs=world, join=True)
    check_outputs(name, out)


def _bucket_worker(rank, world, port, steps, bucket, rotating, q):
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    from multiprime_amd.dist import StepBuckets
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        sb = StepBuckets(5, bucket, torch.device("cpu"), world)
        for round_ in range(3):                               # drain() must leave the object reusable
            for i in range(steps):
                val = torch.full((5, 3), (rank + 1) * 1000 + i + 100 * round_, dtype=torch.int64)
                if rotating:                                  # "the kernel" of mp_eval_launch_rotating: ADDS to its block, clears the next
                    blk, nxt = sb.begin_rotating()
                    ok = ok and bool((blk == 0).all()) and blk.data_ptr() != nxt.data_ptr()
                    blk.add_(val)
                    nxt.zero_()
                else:                                         # "the kernel" of mp_eval_launch: clears its block itself
                    sb.begin_step().copy_(val)
                sb.end_step()
            sb.drain()
            keep = (sb.D - 1) * bucket                        # the ring still holds at least this many of the run's last steps
            for i in range(max(0, steps - keep), steps):
                want = sum((r + 1) * 1000 + i + 100 * round_ for r in range(world))
                ok = ok and bool((sb.block_of(i) == want).all())
        if rank == 0:
            q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rotating", [False, True])
@pytest.mark.parametrize("world,steps,bucket", [(2, 9, 4), (2, 8, 4), (3, 5, 1), (2, 3, 8), (2, 7, 1)])
def test_bucketed_overlapped_allreduce_reduces_every_step(world, steps, bucket, rotating):
    """bench.py's N > 1 exchange (dist.StepBuckets): several steps per collective, a ring of buffers in flight, partial
    last bucket — every step's block must come out as the sum over the ranks; with rotating launches (the launch that
    fills a block clears the next step's) every block must be found zeroed."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_bucket_worker, args=(world, port, steps, bucket, rotating, q), nprocs=world, join=True)
    assert q.get() is True


def _gather_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    import numpy as np
    import torch.distributed as dist
    from multiprime_amd.dist import RowShards
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sh = RowShards()
        sizes = [400000, 0, 7, 123457][:world]
        rng = np.random.default_rng(100 + rank)
        mine = rng.integers(-2 ** 62, 2 ** 62, size=(sizes[rank], 5), dtype=np.int64)
        got = sh.gather_var(mine)
        want = np.concatenate([np.random.default_rng(100 + r).integers(-2 ** 62, 2 ** 62, size=(sizes[r], 5), dtype=np.int64) for r in range(world)])
        ok = got.shape == want.shape and bool((got == want).all())
        cols = sh.gather_columns(np.full((3, sizes[rank] % 1000), rank, np.uint8))          # [m][n_local] -> [m][n_total]
        ok = ok and cols.shape == (3, sum(s % 1000 for s in sizes)) and cols.dtype == np.uint8
        ok = ok and cols[1].tolist() == [r for r in range(world) for _ in range(sizes[r] % 1000)]
        empty = sh.gather_var(np.zeros((0, 2), np.int32))                                    # nobody has anything
        ok = ok and empty.shape == (0, 2)
        if rank == 0:
            q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_gather_var_with_large_and_skewed_payloads(world):
    """The packed all-gather behind the histogram / exception exchange with what a high-entropy alignment produces: megabytes on one
    rank, nothing on another, a handful of rows on a third; empty everywhere; the column-wise form."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 25500 + (os.getpid() % 2000)
    mp.spawn(_gather_worker, args=(world, port, q), nprocs=world, join=True)
    assert q.get() is True


def _entropy_worker(rank, world, port, inp, out, lib_path):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from multiprime_amd._abi import Library
    from multiprime_amd.core import NN_degenerate
    from multiprime_amd.dist import RowShards
    comm = None
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = RowShards()
    try:
        app = NN_degenerate(seq_file=inp, primer_length=18, coverage=0.05, number_of_dege_bases=6, score_of_dege_bases=64,
                            raw_entropy_threshold=9.0, product_len=100, position="2,3,-1", variation=2, distance=4, GC="0.2,0.8",
                            nproc=1, outfile=out, library=Library(lib_path), comm=comm, write_bitsets=True, write_json=True)
        app.run()
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_high_entropy_alignment_sharded_equals_single(oracle_lib, tmp_path):
    """An alignment in which nearly every (row, window) pair is a k-mer of its own — histogram entries ~ rows x windows, the case the
    packed gather was never exercised with — gives the same files on 3 row shards as in one process."""
    import numpy as np
    rng = np.random.default_rng(77)
    n, L = 240, 160
    root = rng.integers(0, 4, L)
    rows = []
    for r in range(n):
        s = root.copy()
        flip = rng.random(L) < 0.22                       # one mutation every ~4.5 columns: almost no 18-mer survives intact
        s[flip] = rng.integers(0, 4, int(flip.sum()))
        txt = "".join("ACGT"[b] for b in s)
        if r % 17 == 0:
            txt = "-" * (r % 9) + txt[r % 9:]
        rows.append(f">s{r}\n{txt}\n")
    inp = tmp_path / "entropy.fa"
    inp.write_text("".join(rows))
    outs = {}
    for world in (1, 3):
        out = tmp_path / f"entropy_w{world}.out"
        port = 27500 + (os.getpid() % 2000) + world
        mp.spawn(_entropy_worker, args=(world, port, str(inp), str(out), oracle_lib.path), nprocs=world, join=True)
        outs[world] = out
    for suffix in ("", ".gap_seq_id_json", ".non_coverage_seq_id_json"):
        a, b = str(outs[1]) + suffix, str(outs[3]) + suffix
        assert os.path.exists(a) == os.path.exists(b), suffix
        if os.path.exists(a):
            assert open(a, "rb").read() == open(b, "rb").read(), suffix
    assert os.path.getsize(outs[1]) > 0
