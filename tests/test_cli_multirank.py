"""The drop-in command itself on several ranks (CPU, gloo): `scripts/multiPrime-core.py` under torch.distributed.run shards the
rows of ONE alignment (rank 0 writes, bytes equal the reference's files), `--ngpu N` launches the ranks itself, and `--batch`
spreads CLUSTERS over the ranks without a collective.  The device calls go to the ABI checker (tests/checker_shim), as in
test_multirank.py; the command line, the launcher glue and the collectives are the product's."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO, golden_input, load_gz_json
from test_core_golden import check_outputs

SCRIPT = os.path.join(REPO, "scripts", "multiPrime-core.py")


def _flags(name):
    fl = load_gz_json(name + ".trace.json.gz")["meta"]["flags"]
    return ["-l", str(fl["l"]), "-n", str(fl["n"]), "-d", str(fl["d"]), "-v", str(fl["v"]), "-e", str(fl["e"]), "-g", fl["g"],
            "-s", str(fl["s"]), "-f", str(fl["f"]), "-c", fl["c"], "-a", str(fl["a"]), "-p", "1"]


def _env(oracle_lib):
    # the children's device calls go to the checker: tests/checker_shim/sitecustomize.py patches the loader's default path in the child
    # (the product loader itself has no switch that admits a non-HIP backend)
    shim = os.path.join(REPO, "tests", "checker_shim")
    env = dict(os.environ, MP_TEST_CHECKER_SO=oracle_lib.path, MP_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1",
               PYTHONPATH=os.pathsep.join([shim] + [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep) if p]))
    env.pop("MPRIME_LIBRARY", None)
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(key, None)
    return env


def _input(name, tmp_path):
    meta = load_gz_json(name + ".trace.json.gz")["meta"]
    inp = tmp_path / (name + ".fa")
    inp.write_bytes(golden_input(meta["input"]))
    return inp


@pytest.mark.parametrize("name,world", [("ivc_v1", 2), ("syn_ragged", 3)])
def test_drop_in_under_the_launcher_shards_rows(name, world, oracle_lib, tmp_path):
    inp, out = _input(name, tmp_path), tmp_path / (name + ".out")
    port = 29600 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), SCRIPT, "-i", str(inp), "-o", str(out)] + _flags(name)
    r = subprocess.run(cmd, env=_env(oracle_lib), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("Total times") == 1                      # rank 0 alone reports
    check_outputs(name, out)                                        # TSV and both JSON side files, byte for byte


def test_ngpu_flag_launches_the_ranks_itself(oracle_lib, tmp_path):
    name = "syn_iupac"
    inp, out = _input(name, tmp_path), tmp_path / (name + ".out")
    r = subprocess.run([sys.executable, SCRIPT, "-i", str(inp), "-o", str(out), "--ngpu", "2"] + _flags(name), env=_env(oracle_lib),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    check_outputs(name, out)


def test_batch_mode_spreads_clusters_over_ranks(oracle_lib, tmp_path):
    """Three alignments with one flag set (ivc_v1's): one process per rank, rank r takes lines r, r + 2, ..."""
    name = "ivc_v1"
    inp = _input(name, tmp_path)
    outs = [tmp_path / f"c{i}.out" for i in range(3)]
    batch = tmp_path / "batch.tsv"
    batch.write_text("# input\toutput\n" + "".join(f"{inp}\t{o}\n" for o in outs))
    r = subprocess.run([sys.executable, SCRIPT, "--batch", str(batch), "--ngpu", "2"] + _flags(name), env=_env(oracle_lib),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    for o in outs:
        check_outputs(name, o)
    summaries = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    assert sorted(s["clusters"] for s in summaries) == [1, 2] and r.stdout.count("Total times") == 3


def test_single_process_batch_and_missing_arguments(oracle_lib, tmp_path):
    name = "syn_edge"
    inp, out = _input(name, tmp_path), tmp_path / "one.out"
    batch = tmp_path / "batch.tsv"
    batch.write_text(f"{inp}\t{out}\n")
    r = subprocess.run([sys.executable, SCRIPT, "--batch", str(batch)] + _flags(name), env=_env(oracle_lib), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    check_outputs(name, out)
    r = subprocess.run([sys.executable, SCRIPT, "-i", str(inp)], env=_env(oracle_lib), capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "required" in r.stderr                # argparse's exit code, like the reference's parser


def test_batch_workers_and_processes_on_one_gpu(oracle_lib, tmp_path):
    """`--batch` with several clusters in flight: worker threads (a context each, kept across clusters) and `--batch-procs` child
    processes that share the list — every cluster's files equal the reference's whichever way they were scheduled."""
    name = "syn_iupac"
    inp = _input(name, tmp_path)
    for tag, extra in (("threads", ["--batch-workers", "3"]), ("procs", ["--batch-procs", "2", "--batch-workers", "2"])):
        outs = [tmp_path / f"{tag}{i}.out" for i in range(5)]
        batch = tmp_path / f"{tag}.tsv"
        batch.write_text("".join(f"{inp}\t{o}\n" for o in outs))
        r = subprocess.run([sys.executable, SCRIPT, "--batch", str(batch)] + extra + _flags(name), env=_env(oracle_lib), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        for o in outs:
            check_outputs(name, o)
        assert r.stdout.count("Total times") == 5
        summaries = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
        assert len(summaries) == 1                                      # the children's own summaries are folded into the rank's
        last = summaries[-1]
        assert last["clusters"] == 5 and (last.get("processes") == 2 if tag == "procs" else last["workers"] == 3)
        n_seq = sum(1 for line in open(inp) if line.startswith(">"))
        assert last["sequences"] == 5 * n_seq


@pytest.mark.parametrize("grid,world", [("2x2", 4), ("auto", 3)])
def test_grid_flag_splits_rows_and_windows(grid, world, oracle_lib, tmp_path):
    """--grid RxG: R row shards x G window groups (dist.ShardGrid); auto = window groups only when the alignment fits a device."""
    name = "ivc_v1"
    inp, out = _input(name, tmp_path), tmp_path / (name + ".out")
    r = subprocess.run([sys.executable, SCRIPT, "-i", str(inp), "-o", str(out), "--ngpu", str(world), "--grid", grid] + _flags(name),
                       env=_env(oracle_lib), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("Total times") == 1
    check_outputs(name, out)
