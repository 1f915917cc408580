"""Host logic (multiprime_amd.core) driven through the C ABI, against the reference's own
outputs recorded in tests/golden/ (TSV byte for byte, JSON side files semantically).

On CPU the ABI is served by the oracle library (test infrastructure); the `gpu` variant at the
bottom runs the same comparison through the HIP library on a real MI355X.
"""
import json
import os

import pytest

from conftest import GOLDEN, golden_input, load_gz_json
from multiprime_amd.core import NN_degenerate

FAST = ["syn_iupac", "syn_v2", "syn_ragged", "syn_v3_k27", "syn_edge", "ivc_v0", "ivc_v1", "ivc_v2",
        "msa1000_k18_d64", "msa1000_k20_d64", "msa1000_k22_d64", "msa1000_k18_d10", "msa1000_k30_d64", "msa1000_k31_d64", "syn_v2_k31", "msa1000_c1_f06", "ivc_e30_g"]
# primers of 32..63 bases (64-bit window words; recorded from V20 at -l 32, 33, 36, 40, 45, 50, 63)
WIDE = ["cluster0_k32", "syn_iupac_k33", "msa1000_k36_d64", "syn_ragged_k40", "ivc_k45_v2", "syn_v2_k50", "syn_edge_k63"]
FAST = FAST + [n for n in WIDE if n != "cluster0_k32"]
FULL = FAST + ["cluster0_v1", "cluster0_v2", "cluster0_v0_d64", "testfa", "cluster0_k32"]


def canon_noncov(d):
    return {str(k): [{km: sorted(ids) for km, ids in sorted(side.items())} for side in v] for k, v in d.items()}


def canon_gap(d):
    return {str(k): {km: list(ids) for km, ids in sorted(v.items())} for k, v in d.items()}


def run_fixture(name, lib, tmp_path, **kw):
    meta = load_gz_json(name + ".trace.json.gz")["meta"]
    fl = meta["flags"]
    inp = tmp_path / (name + ".fa")
    inp.write_bytes(golden_input(meta["input"]))
    out = tmp_path / (name + ".out")
    app = NN_degenerate(seq_file=str(inp), primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"],
                        score_of_dege_bases=fl["d"], raw_entropy_threshold=fl["e"], product_len=fl["s"],
                        position=fl["c"], variation=fl["v"], distance=fl["a"], GC=fl["g"], nproc=1,
                        outfile=str(out), library=lib, **kw)
    assert (int(app.start_position), int(app.stop_position)) == (meta["start"], meta["stop"])
    assert app.total_sequence_number == meta["n_seq"]
    app.run()
    return app, out


def check_outputs(name, out):
    with open(os.path.join(GOLDEN, name + ".tsv"), "rb") as f:
        want = f.read()
    assert out.read_bytes() == want, "TSV differs from the reference's"
    got_nc = canon_noncov(json.load(open(str(out) + ".non_coverage_seq_id_json")))
    assert got_nc == load_gz_json(name + ".noncov.json.gz")
    got_gap = canon_gap(json.load(open(str(out) + ".gap_seq_id_json")))
    assert got_gap == load_gz_json(name + ".gap.json.gz")


@pytest.mark.parametrize("name", FULL)
def test_host_logic_matches_reference(name, oracle_lib, tmp_path):
    _, out = run_fixture(name, oracle_lib, tmp_path)
    check_outputs(name, out)


def test_region_too_short_exits_like_reference(oracle_lib, tmp_path, capsys):
    inp = tmp_path / "short.fa"
    inp.write_bytes(b">a\nACGTACGTACGTACGTACGTACGTACGT\n>b\nACGTACGTACGTACGTACGTACGTACGT\n")
    with pytest.raises(SystemExit) as e:
        NN_degenerate(seq_file=str(inp), primer_length=18, coverage=0.8, product_len=100, position="1,2,-1",
                      variation=1, GC="0.2,0.7", outfile=str(tmp_path / "o"), library=oracle_lib)
    assert e.value.code == 1                                   # V20:635-638
    assert "Non candidate primers" in capsys.readouterr().out


@pytest.mark.gpu
@pytest.mark.parametrize("name", FULL)
def test_hip_path_matches_reference(name, hip_lib, tmp_path):
    assert hip_lib.backend == "hip"
    _, out = run_fixture(name, hip_lib, tmp_path)
    check_outputs(name, out)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["syn_edge", "syn_ragged", "msa1000_k18_d64", "testfa"])
def test_streamed_load_equals_the_array_load(name, hip_lib, tmp_path, monkeypatch):
    """mp_load_msa_fasta (residue bytes from the parsed file through the context's registered transfer buffers: the default of a single
    process) against mp_load_msa of the rows array (MP_LOAD_STREAM=0), and without registered buffers (MP_NO_PIN): the reference's files."""
    for mode, env in (("stream", {}), ("array", {"MP_LOAD_STREAM": "0"}), ("unpinned", {"MP_NO_PIN": "1"})):
        for key in ("MP_LOAD_STREAM", "MP_NO_PIN"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        d = tmp_path / mode
        d.mkdir()
        app, out = run_fixture(name, hip_lib, d)
        check_outputs(name, out)
        app.ctx.close()


def test_one_context_for_many_alignments_on_the_checker(oracle_lib, tmp_path):
    """The same control flow (a context kept across alignments of different shapes) without a GPU."""
    ctx = None
    for name in ("msa1000_k18_d64", "syn_iupac", "msa1000_k18_d64", "syn_edge"):
        d = tmp_path / name
        d.mkdir(exist_ok=True)
        app, out = run_fixture(name, oracle_lib, d, context=ctx)
        ctx = app.ctx
        check_outputs(name, out)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pool", ["1", "0"])
def test_one_context_for_many_alignments(pool, hip_lib, tmp_path, monkeypatch):
    """A --batch worker keeps ONE context across alignments: every stage then receives device blocks a stage of the alignment before
    released (the context's pool, csrc/common.hpp — no hipFree in between, so nothing waits for the device either).  Alignments of different
    shapes one after the other, twice round, each with the reference's bytes; MP_DEVICE_POOL=0 beside it."""
    monkeypatch.setenv("MP_DEVICE_POOL", pool)
    ctx = None
    for rnd in range(2):
        for name in ("msa1000_k18_d64", "syn_iupac", "cluster0_v2", "msa1000_k18_d64", "syn_edge", "ivc_v2"):
            d = tmp_path / ("%d_%s" % (rnd, name))
            d.mkdir(exist_ok=True)
            app, out = run_fixture(name, hip_lib, d, context=ctx)
            ctx = app.ctx
            check_outputs(name, out)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["syn_iupac", "syn_ragged", "syn_edge", "ivc_v2", "msa1000_k18_d64", "msa1000_k22_d64", "msa1000_k31_d64", "cluster0_v2", "testfa",
                                  "syn_ragged_k40", "ivc_k45_v2", "syn_edge_k63"])
def test_streamed_planning_equals_blocking_read_back(name, hip_lib, tmp_path, monkeypatch):
    """Without JSON side files a single process plans straight from the device (mp_plan_create_streamed: histogram entries read back in
    bands beside the planning threads; packed keys and the k >= 22 / 64-bit word tables alike).  Same TSV as the reference, and the
    same plan tables as the two blocking calls (MP_PLAN_STREAM=0), window by window."""
    want = open(os.path.join(GOLDEN, name + ".tsv"), "rb").read()
    plans, gated = {}, {}
    for mode in ("1", "0", "1-staged", "0-staged", "1-gate"):
        monkeypatch.setenv("MP_PLAN_STREAM", mode[0])
        # "1-gate": the entropy gate decided on the device where that is certain (the default of the streamed route); the other modes
        # leave every window to the host, so that the two routes can be compared window by window
        monkeypatch.setenv("MP_DEVICE_GATE", "1" if mode == "1-gate" else "0")
        if mode.endswith("staged"):              # neither registered transfers nor parallel page faults: the runtime's staging path
            monkeypatch.setenv("MP_NO_PIN", "1")
            monkeypatch.setenv("MP_NO_PREFAULT", "1")
        d = tmp_path / mode
        d.mkdir()
        app, out = run_fixture(name, hip_lib, d, write_json=False)
        assert out.read_bytes() == want, f"TSV differs from the reference's with MP_PLAN_STREAM={mode}"
        assert (app._dev_entries is None) == (mode[0] == "1")
        st, cn, gn, cb, tb = app.plan.windows()
        plans[mode] = (st.tolist(), cn.tolist(), gn.tolist(), [repr(x) for x in cb.tolist()], [repr(x) for x in tb.tolist()],
                       [a.tolist() for a in app.plan.candidates()])
        gated[mode] = app.stats.get("windows_device_gated", 0)
        app.ctx.close()
    assert plans["1"] == plans["0"] == plans["1-staged"] == plans["0-staged"]
    assert gated["1"] == 0
    # with the gate on the device: a window it rejected is one the host rejects at the entropy gate (status 3) or at one of the two gates
    # in front of it (gap fraction, empty cover: V20:713-723 in that order), every other window and every candidate is the same
    g, h = plans["1-gate"], plans["1"]
    n_dev = 0
    for w, (sg, sh) in enumerate(zip(g[0], h[0])):
        if sg == 6:                              # MP_WIN_ENTROPY_DEVICE
            assert sh in (1, 2, 3), (w, sh)
            n_dev += 1
        else:
            assert (sg, g[1][w], g[2][w], g[3][w], g[4][w]) == (sh, h[1][w], h[2][w], h[3][w], h[4][w]), w
    assert g[5] == h[5] and n_dev == gated["1-gate"]
    if name == "msa1000_k18_d64":            # (1000 sequences: most of the windows the host rejects are beyond the bound; 166 of ivc_v2: none)
        assert n_dev > 50, "the device gate rejected next to nothing on an alignment where the host rejects hundreds of windows"


def _side_bytes(out):
    return (open(str(out) + ".non_coverage_seq_id_json", "rb").read(), open(str(out) + ".gap_seq_id_json", "rb").read())


@pytest.mark.parametrize("name", ["syn_iupac", "syn_ragged", "syn_edge", "ivc_v2", "msa1000_k18_d10", "cluster0_v2"])
def test_native_json_writer_equals_python_writer(name, oracle_lib, tmp_path, monkeypatch):
    """mp_plan_write_side_files (single process) and the Python writer of the row-sharded path give the same bytes."""
    (tmp_path / "n").mkdir()
    (tmp_path / "p").mkdir()
    _, out_n = run_fixture(name, oracle_lib, tmp_path / "n")
    monkeypatch.setenv("MP_JSON_WRITER", "python")
    _, out_p = run_fixture(name, oracle_lib, tmp_path / "p")
    assert _side_bytes(out_n) == _side_bytes(out_p)


@pytest.mark.parametrize("name,run", [("syn_iupac", 1), ("syn_edge", 2), ("ivc_v2", 3), ("cluster0_v2", 7), ("msa1000_k18_d10", 1000)])
def test_native_json_writer_in_runs_of_windows(name, run, oracle_lib, tmp_path, monkeypatch):
    """mp_plan_write_side_files_part: the side files written a few output windows at a time (what a deep alignment needs, where
    the labels of all output windows would be gigabytes) are the same bytes as written in one call."""
    (tmp_path / "whole").mkdir()
    (tmp_path / "runs").mkdir()
    monkeypatch.setenv("MP_JSON_BATCH", "100000")
    _, out_w = run_fixture(name, oracle_lib, tmp_path / "whole")
    monkeypatch.setenv("MP_JSON_BATCH", str(run))
    _, out_r = run_fixture(name, oracle_lib, tmp_path / "runs")
    assert _side_bytes(out_w) == _side_bytes(out_r)


def test_native_json_writer_escapes_ids_like_json_dump(oracle_lib, tmp_path, monkeypatch):
    _special_ids(oracle_lib, tmp_path, monkeypatch)


@pytest.mark.gpu
def test_special_ids_through_the_gpu_path(hip_lib, tmp_path, monkeypatch):
    _special_ids(hip_lib, tmp_path, monkeypatch)


def _special_ids(lib, tmp_path, monkeypatch):
    """ids with quotes, backslashes, control bytes, non-ASCII UTF-8 (BMP and astral) and invalid UTF-8 (surrogateescape): the
    native writer's strings are json.dump's (ensure_ascii) and the file reads back to the same dicts."""
    import random
    rng = random.Random(5)
    base = "".join(rng.choice("ACGT") for _ in range(90))
    ids = [b'plain', b'quo"te', b'back\\slash', b'tab\there', "été".encode(), "中文".encode(),
           "\U0001f9ec x".encode(), b'bad\xff\xfebytes', b'\x7f del', b'trunc\xe4\xb8', b'over\xc0\xaf', b'sur\xed\xa0\x80']
    rec = []
    for i, name in enumerate(ids):
        s = list(base)
        for _ in range(i % 5):
            s[rng.randrange(len(s))] = rng.choice("ACGT")
        if i % 3 == 1:
            s[30 + i] = "-"
        if i % 4 == 2:
            s[50 + i] = "R"
        rec.append(b">" + name + b"\n" + "".join(s).encode() + b"\n")
    outs = []
    for mode in ("native", "python"):
        d = tmp_path / mode
        d.mkdir()
        (d / "in.fa").write_bytes(b"".join(rec))
        monkeypatch.setenv("MP_JSON_WRITER", mode)
        app = NN_degenerate(seq_file=str(d / "in.fa"), primer_length=18, coverage=0.3, number_of_dege_bases=4, score_of_dege_bases=16,
                            product_len=30, position="1,2,-1", variation=1, GC="0.2,0.8", nproc=1, outfile=str(d / "o"),
                            library=lib)
        app.run()
        outs.append(_side_bytes(d / "o"))
        assert len(open(d / "o").read().splitlines()) > 1
    assert outs[0] == outs[1]
    nc = json.loads(outs[0][0])
    seen = {i for sides in nc.values() for side in sides for v in side.values() for i in v}
    seen |= {i for d in json.loads(outs[0][1]).values() for v in d.values() for i in v}
    assert any("\udcff" in i for i in seen) and any("\U0001f9ec" in i for i in seen) and any('"' in i for i in seen)
