"""The n_ranks > 1 branch of csrc/comm.hip, executed: two and three PROCESSES on the box's single GPU, each with its own HIP
context, form one communicator through `mp_comm_init` and run every collective of mprime.h section 9 with several ranks —
`mp_comm_allgather_i64`, `mp_comm_allgatherv` with skewed / empty / multi-chunk payloads (the padded-slot gather),
`mp_comm_allreduce_host_i64`, `mp_eval_candidates_allreduce` on row shards.  Results must equal the single-process ones.

RCCL itself refuses two ranks on one device, and the boxes of this pool have one: the transport under the six `nccl*` entry points
is tests/stub_rccl (shared memory + hipMemcpy, test infrastructure), selected through MP_RCCL_LIBRARY.  Everything above those six
calls — scratch sizing, slot padding and unpadding, count checks, stream ordering, the fused kernel -> collective -> copy — is the
product's code, running with n_ranks > 1.  The whole core step on row shards with this transport: test_core_step_* below
(`MP_NATIVE_COMM=force`; torch.distributed/gloo then only carries the communicator's id).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO, golden_input, load_gz_json

STUB = os.path.join(REPO, "tests", "stub_rccl", "librccl_stub.so")


def _eval_case(hip_lib, wd):
    """A seeded alignment with gaps, ragged edges and IUPAC codes; single-process counts of nested and unrelated candidates."""
    sys.path.insert(0, REPO)
    import bench
    from multiprime_amd import host
    from multiprime_amd.synth import synth_block, synth_root
    n, L, k, v, seed = 5003, 400, 18, 1, 4242
    rows = synth_block(0, n, L, seed, p_gap=0.004, edge_frac=0.2, p_iupac=2e-4, block_rows=1024)
    ctx = hip_lib.context(0)
    ctx.load_msa(rows.reshape(-1), np.arange(n + 1, dtype=np.int64) * L)
    p0, W = 8, L - 16 - k
    n_ex = ctx.build_windows(p0, W, k, v)
    if n_ex:
        ex_w, ex_r, ex_codes = ctx.get_exceptions(n_ex)
        sel = (ex_codes == 0).sum(axis=1) <= v
        words, src = host.expand_kmer_words(ex_codes[sel])
        ctx.set_extra_rows(ex_w[sel][src], words)
    root_codes = np.array([1, 2, 4, 8], np.uint8)[synth_root(L, seed)]
    sF, sR = 0b1100, 0b11 << (k - 2)
    out = dict(rows=rows, k=k, v=v, p0=p0, W=W, sF=sF, sR=sR)
    for tag, nested in (("nested", True), ("unrelated", False)):
        cw, codes = bench.make_candidates(root_codes, p0, W, k, 4, seed + (0 if nested else 1), nested=nested)
        out["cw_" + tag], out["codes_" + tag] = cw, codes
        out["want_" + tag] = ctx.eval_candidates(cw, codes, sF, sR)
    out["freq"], out["nn"] = ctx.window_stats()
    ctx.close()
    np.savez(os.path.join(wd, "eval_case.npz"), **out)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_collectives_of_the_abi_with_several_ranks_on_one_gpu(world, hip_lib, tmp_path, monkeypatch):
    assert os.path.exists(STUB), "tests/stub_rccl/librccl_stub.so is built by __graft_entry__.build()"
    wd = str(tmp_path)
    _eval_case(hip_lib, wd)
    big = (8 << 20) + 12345                                           # beyond one 8 MiB transport chunk of the stand-in
    gathers = [[0] * world, [1] + [0] * (world - 1), [17, 4096, 5][:world], [big] + [3] * (world - 1), [0] * (world - 1) + [70001]]
    rng = np.random.default_rng(world)
    exchanges = [[[0] * world for _ in range(world)],
                 [[int(s == 0 and d == world - 1) for d in range(world)] for s in range(world)],               # one byte, one pair
                 [[int(x) for x in rng.integers(0, 5000, size=world)] for _ in range(world)],
                 [[(big if (s, d) == (0, 1 % world) else 7 * (s + 1) + d) for d in range(world)] for s in range(world)]]
    json.dump({"gathers": gathers, "sums": [0, 1, 1000, (1 << 20) + 7], "exchanges": exchanges}, open(os.path.join(wd, "spec.json"), "w"))
    env = dict(os.environ, MP_RCCL_LIBRARY=STUB)
    # the id comes from the library's own export, as a host would draw it on rank 0
    uid = subprocess.check_output([sys.executable, "-c",
                                   "import sys; sys.path.insert(0, %r)\nfrom multiprime_amd._abi import Library\n"
                                   "print(Library().context(0).comm_unique_id().hex())" % REPO], env=env).decode().strip()
    assert len(uid) == 256
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "comm_rank_worker.py"), str(r), str(world), uid, wd], env=env)
             for r in range(world)]
    codes = [p.wait(timeout=600) for p in procs]
    results = [json.load(open(os.path.join(wd, f"rank{r}.json"))) for r in range(world)]
    for r in results:
        assert r["ok"], r.get("error")
    assert codes == [0] * world


def _core_worker(rank, world, port, name, inp, out, write_json):
    sys.path.insert(0, REPO)
    os.environ["MP_RCCL_LIBRARY"] = STUB
    os.environ["MP_NATIVE_COMM"] = "force"
    import gzip
    import torch.distributed as dist
    from multiprime_amd._abi import Library
    from multiprime_amd.core import NN_degenerate
    from multiprime_amd.dist import RowShards
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        meta = json.loads(gzip.open(os.path.join(GOLDEN, name + ".trace.json.gz")).read())["meta"]
        fl = meta["flags"]
        comm = RowShards()
        app = NN_degenerate(seq_file=inp, primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"],
                            score_of_dege_bases=fl["d"], raw_entropy_threshold=fl["e"], product_len=fl["s"],
                            position=fl["c"], variation=fl["v"], distance=fl["a"], GC=fl["g"], nproc=1,
                            outfile=out, library=Library(), comm=comm, write_json=write_json)
        assert comm.native is not None, "the library's own communicator must carry the collectives"
        seen = app.ctx.comm_describe()
        assert seen[0] == world and seen[1] == rank and seen[2].endswith("librccl_stub.so"), seen
        app.run()
        assert app._win_split == (not write_json)
        from test_multirank import check_traffic
        check_traffic(app, world)                     # window-split: the tables reach the window owners by mp_comm_alltoallv
    finally:
        dist.destroy_process_group()


def _synthetic_worker(rank, world, port, inp, out):
    """Row shards of a synthetic alignment (rows and variation spread evenly): what a rank receives of the histogram entries is about
    what it holds itself."""
    sys.path.insert(0, REPO)
    os.environ["MP_RCCL_LIBRARY"] = STUB
    os.environ["MP_NATIVE_COMM"] = "force"
    import torch.distributed as dist
    from multiprime_amd._abi import Library
    from multiprime_amd.core import NN_degenerate
    from multiprime_amd.dist import RowShards
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = RowShards()
        app = NN_degenerate(seq_file=inp, primer_length=18, coverage=0.8, number_of_dege_bases=4, score_of_dege_bases=10, raw_entropy_threshold=3.6,
                            product_len=150, position="2,3,-1", variation=1, distance=4, GC="0.2,0.7", nproc=1, outfile=out, library=Library(),
                            comm=comm, write_json=False)
        app.run()
        assert app._win_split and comm.native is not None
        (sent, received), = [(s, r) for w, s, r in comm.traffic if w == "histogram entries"]
        own = comm.tables["histogram entries"]
        assert received <= 1.2 * own, (sent, received, own)          # the all-gather delivered (world - 1) x own
        assert received >= 0.5 * own * (world - 1) / world
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_window_owners_receive_about_their_own_share(world, tmp_path):
    """The bench's synthetic alignment (16384 rows) on `world` row shards through the library's communicator: TSV == the checker's
    committed hash, and the personalised exchange hands a rank <= 1.2 x the entries it holds itself."""
    import hashlib
    import torch.multiprocessing as mp
    from multiprime_amd.synth import synth_block, to_fasta
    db = json.load(open(os.path.join(REPO, "tests", "golden", "synth_pipeline.json")))
    entry = next(e for e in db["entries"] if e["rows"] == 16384)
    inp = tmp_path / "syn.fa"
    inp.write_bytes(to_fasta(synth_block(0, 16384, entry["cols"], entry["seed"])))
    out = tmp_path / "syn.tsv"
    port = 30700 + (os.getpid() % 2000)
    mp.spawn(_synthetic_worker, args=(world, port, str(inp), str(out)), nprocs=world, join=True)
    assert hashlib.sha256(out.read_bytes()).hexdigest() == entry["tsv_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,write_json", [("syn_iupac", 2, True), ("msa1000_k18_d64", 3, True), ("syn_ragged", 3, False),
                                                   ("cluster0_v2", 2, False)])
def test_core_step_on_row_shards_through_the_library_communicator(name, world, write_json, tmp_path):
    """The whole core step, rows sharded over `world` processes on the one GPU, EVERY collective through mp_comm_* with
    n_ranks = world (histogram entries and exceptions through the padded all-gather, statistics through the host all-reduce,
    the counters through the fused evaluate + all-reduce): files identical to the reference's."""
    import torch.multiprocessing as mp
    from test_core_golden import check_outputs
    meta = load_gz_json(name + ".trace.json.gz")["meta"]
    inp = tmp_path / (name + ".fa")
    inp.write_bytes(golden_input(meta["input"]))
    out = tmp_path / (name + ".out")
    port = 30300 + (os.getpid() % 2000)
    mp.spawn(_core_worker, args=(world, port, name, str(inp), str(out), write_json), nprocs=world, join=True)
    if write_json:
        check_outputs(name, out)
    else:
        with open(os.path.join(GOLDEN, name + ".tsv"), "rb") as f:
            assert out.read_bytes() == f.read(), "TSV differs from the reference's"
