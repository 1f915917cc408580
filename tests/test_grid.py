"""2-D shards on CPU over gloo (dist.ShardGrid): R row shards x G window groups — every window group is an independent row-sharded job
on its contiguous share of the windows, the leaders' rows meet on rank 0.  The files must be the reference's, byte for byte (TSV)
and as parsed dicts (the two JSON side files), for grids 2x2, 1x2, 1x3, 3x2 and with window-split planning inside a row group."""
import json
import os
import sys

import pytest
import torch.multiprocessing as mp

from conftest import GOLDEN, REPO, golden_input, load_gz_json


def _worker(rank, world, port, name, inp, out, lib_path, R, G, write_json):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from multiprime_amd._abi import Library
    from multiprime_amd.core import NN_degenerate
    from multiprime_amd.dist import ShardGrid
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gzip
        fl = json.loads(gzip.open(os.path.join(GOLDEN, name + ".trace.json.gz")).read())["meta"]["flags"]
        grid = ShardGrid(R, G)
        assert (grid.ri, grid.gi) == divmod(rank, G) and (grid.comm is None) == (R == 1)
        app = NN_degenerate(seq_file=inp, primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"],
                            score_of_dege_bases=fl["d"], raw_entropy_threshold=fl["e"], product_len=fl["s"],
                            position=fl["c"], variation=fl["v"], distance=fl["a"], GC=fl["g"], nproc=1,
                            outfile=out, library=Library(lib_path), grid=grid, write_json=write_json)
        app.run()
        if R > 1:
            assert app.comm.world == R and app._win_split == (not write_json)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,R,G,write_json", [("ivc_v1", 2, 2, True), ("syn_iupac", 1, 2, True), ("msa1000_k18_d64", 1, 3, True),
                                                 ("syn_ragged", 3, 2, True), ("msa1000_k18_d64", 2, 2, False), ("syn_edge", 1, 4, False),
                                                 ("syn_ragged_k40", 2, 2, True)])
def test_grid_run_matches_reference(name, R, G, write_json, oracle_lib, tmp_path):
    from test_core_golden import check_outputs
    meta = load_gz_json(name + ".trace.json.gz")["meta"]
    inp = tmp_path / (name + ".fa")
    inp.write_bytes(golden_input(meta["input"]))
    out = tmp_path / (name + ".out")
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(R * G, port, name, str(inp), str(out), oracle_lib.path, R, G, write_json), nprocs=R * G, join=True)
    if write_json:
        check_outputs(name, out)
    else:
        with open(os.path.join(GOLDEN, name + ".tsv"), "rb") as f:
            assert out.read_bytes() == f.read(), "TSV differs from the reference's"
        assert not os.path.exists(str(out) + ".gap_seq_id_json")


def test_shape_choice():
    from multiprime_amd.dist import ShardGrid
    assert ShardGrid.best_shape(8, 1048576, 1000) == (1, 8)                       # 6 GB a replica: every GPU holds every row
    assert ShardGrid.best_shape(8, 1048576, 1000, replica_budget_bytes=4 << 30) == (2, 4)
    assert ShardGrid.best_shape(8, 1 << 26, 4000, replica_budget_bytes=64 << 30) == (8, 1)
    assert ShardGrid.best_shape(6, 1000, 500) == (1, 6)
    assert ShardGrid.parse("2x4", 8) == (2, 4)
    with pytest.raises(ValueError):
        ShardGrid.parse("3x3", 8)
    with pytest.raises(ValueError):
        ShardGrid.parse("four", 4)
