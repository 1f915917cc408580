"""Pins the CPU oracle (oracle/mprime_oracle.c) to the reference: every intermediate the C ABI
produces is compared with what multiPrime-core_V20.py computed internally on the same input
(traces recorded by tests/golden/make_golden.py): the per-window `cover` and `gap_sequence`
dictionaries in insertion order (device histograms + IUPAC exceptions, ordered by the native planning stage),
cover_number, the state_matrix / trans_matrix counts (mp_window_stats), the Viterbi seeds and the result of every
mis_primer_check call.
"""
import numpy as np
import pytest

from conftest import golden_input, load_gz_json
from multiprime_amd import iupac
from multiprime_amd.core import NN_degenerate

NAMES = ["syn_iupac", "syn_v2", "syn_ragged", "syn_v3_k27", "syn_edge", "ivc_v1", "msa1000_k18_d64", "msa1000_k22_d64", "msa1000_k30_d64", "msa1000_k31_d64", "msa1000_c1_f06", "ivc_e30_g", "cluster0_v0_d64",
         "cluster0_v2", "cluster0_k32", "syn_iupac_k33", "msa1000_k36_d64", "syn_ragged_k40", "ivc_k45_v2", "syn_v2_k50", "syn_edge_k63"]


def open_fixture(name, lib, tmp_path):
    tr = load_gz_json(name + ".trace.json.gz")
    fl = tr["meta"]["flags"]
    inp = tmp_path / (name + ".fa")
    inp.write_bytes(golden_input(tr["meta"]["input"]))
    app = NN_degenerate(seq_file=str(inp), primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"],
                        score_of_dege_bases=fl["d"], raw_entropy_threshold=fl["e"], product_len=fl["s"],
                        position=fl["c"], variation=fl["v"], distance=fl["a"], GC=fl["g"], nproc=1,
                        outfile=str(tmp_path / "out"), library=lib, write_json=False)
    return app, tr


def check_against_trace(app, tr):
    k = app.primer_length
    plan = app._plan(keep_tables=True)          # device tables -> native planning stage, every window's tables kept
    status, cover_number, gap_number, _, _ = plan.windows()
    p0 = int(app.start_position)
    sym = iupac.SYMBOL_LUT
    cand_w, cand_p, want = [], [], []
    n_tables = 0
    n_stats = [0, 0, 0]
    for pos, rec in sorted(tr["windows"].items(), key=lambda kv: int(kv[0])):
        if "cover" not in rec:
            continue
        w = int(pos) - p0
        for which, name_ in ((0, "cover"), (1, "gap")):
            codes, counts, _ = plan.window_table(w, which)
            got = list(zip(iupac.strings_of(sym[codes]), counts.tolist()))
            assert got == [tuple(x) for x in rec[name_]], f"{name_} dict at {pos}"
        assert cover_number[w] == rec["cover_number"] and gap_number[w] == rec["gap_number"]
        n_tables += 1
        if "freq" in rec:                                   # state_matrix as the reference built it (rows it has, V20:541-554)
            want_f = np.zeros((4, k), np.int64)
            for name_, row in zip(rec["freq_rows"], rec["freq"]):
                want_f["ACGT".index(name_)] = row
            assert np.array_equal(app._freq[w], want_f), f"state_matrix at {pos}"
            n_stats[0] += 1
        if rec.get("NN") is not None:                       # trans_matrix (V20:556-577)
            assert np.array_equal(app._nn[w], np.asarray(rec["NN"], np.int64)), f"trans_matrix at {pos}"
            n_stats[1] += 1
        if "NM" in rec:                                     # get_optimal_primer_by_viterbi (V20:579-593), native host stage
            assert status[w] == 0
            assert plan.seeds(w)[0].tolist() == rec["NM"], f"Viterbi seed at {pos}"
            n_stats[2] += 1
        for primer, F, R, perfect, _ in rec["mis"]:
            cand_w.append(w)
            cand_p.append(primer)
            want.append((perfect, F, R))
    assert n_tables > 0 and min(n_stats) > 0                       # mp_window_stats and the Viterbi seeds were pinned too
    order = np.argsort(np.asarray(cand_w), kind="stable")
    codes = iupac.MASK_LUT[np.frombuffer("".join(cand_p).encode(), np.uint8)].reshape(len(cand_p), k)
    got = app.ctx.eval_candidates(np.asarray(cand_w, np.int32)[order], codes[order], app._sF, app._sR)
    assert got.tolist() == [list(want[i]) for i in order]
    return len(cand_w)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_internals(name, oracle_lib, tmp_path):
    app, tr = open_fixture(name, oracle_lib, tmp_path)
    n = check_against_trace(app, tr)
    assert n == tr["meta"]["n_mis_calls"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_reference_internals(name, hip_lib, tmp_path):
    app, tr = open_fixture(name, hip_lib, tmp_path)
    check_against_trace(app, tr)
