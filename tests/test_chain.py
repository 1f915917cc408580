"""BASELINE config 5 — the multi-cluster run — as a parity test of the WHOLE rule chain (SURVEY §8d input 5):
core step -> pairing -> cat -> get_Maxprimerset -> primerset_format -> finDimer, through this build's drop-ins.

  * against the UNMODIFIED reference chain: tests/golden/chain.json.gz holds every file the reference's own scripts wrote for
    eight small synthetic clusters (tests/golden/make_golden_chain.py ran them as the Snakemake rules would:
    multiPrime.py:200-207, 232-238, 253-256, 277-295, 396-415).  CPU: the chain over the oracle library; `-m gpu`: over the HIP
    library.  Every file must be identical (the two JSON side files as parsed dictionaries with id lists as sets — their key order
    depends on the reference's hash seed);
  * at scale (`-m gpu`): 16 synthetic clusters of 500 .. 5000 sequences, the chain run twice — HIP library and plain-C oracle
    behind the same ABI — with every file identical, including the deep clusters that take the device-resident bitset route.
"""
import gzip
import json
import os
import sys

import numpy as np
import pytest

from conftest import REPO, load_gz_json

sys.path.insert(0, os.path.join(REPO, "tools"))
import multi_cluster as mc  # noqa: E402


# the chain as recorded from the reference: multiPrime.yaml's flags (-l 18, eight clusters) and the same with -l 36 (five of them;
# primers longer than one 32-bit window word — at that length only one cluster keeps a candidate pair through the pairing filters)
GOLDENS = [("chain.json.gz", 18), ("chain_k36.json.gz", 36)]


def _golden_clusters(tmp_path, golden="chain.json.gz"):
    g = load_gz_json(golden)
    fastas = {}
    for c in g["clusters"]:
        fa = tmp_path / (c["name"] + ".tfa")
        fa.write_bytes(gzip.decompress(bytes.fromhex(c["fasta_gz_hex"])))
        fastas[c["name"]] = (str(fa), c["rows"])
    return g, fastas


def _canon_json(obj):
    """Side files with their id lists as sorted lists (V20 fills them in dict order of a set-like walk)."""
    if isinstance(obj, dict):
        return {k: _canon_json(v) for k, v in obj.items()}
    if isinstance(obj, list):
        if obj and all(isinstance(x, str) for x in obj):
            return sorted(obj)
        return [_canon_json(x) for x in obj]
    return obj


def _check_against_reference(g, got):
    want = g["files"]
    assert sorted(got) == sorted(want), (sorted(set(got) ^ set(want)))
    for fn, rec in want.items():
        if "json" in rec:
            assert _canon_json(json.loads(got[fn])) == _canon_json(rec["json"]), fn
        else:
            assert got[fn].decode() == rec["text"], fn
    final = got["final_maxprimers_set.xls"].decode().splitlines()
    if g.get("primer_length", 18) == 18:
        # the run is not trivial: seven clusters reach the final set, one has no candidate pair and goes to .next.xls
        assert len(final) == 8 and got["final_maxprimers_set.next.xls"].count(b"\n") == 1
    else:
        assert len(final) >= 2 and len(got["final_maxprimers_set.fa.findimer"]) > 0


@pytest.mark.parametrize("golden,k", GOLDENS)
def test_chain_on_the_oracle_equals_the_reference_chain(golden, k, oracle_lib, tmp_path):
    g, fastas = _golden_clusters(tmp_path, golden)
    wd = tmp_path / "run"
    _check_against_reference(g, mc.run_chain(str(wd), fastas, library=oracle_lib, primer_length=k))


@pytest.mark.gpu
@pytest.mark.parametrize("golden,k", GOLDENS)
def test_chain_on_the_gpu_equals_the_reference_chain(golden, k, hip_lib, tmp_path):
    g, fastas = _golden_clusters(tmp_path, golden)
    wd = tmp_path / "run"
    _check_against_reference(g, mc.run_chain(str(wd), fastas, library=hip_lib, primer_length=k))
    # the same clusters through the bitset route (no JSON side files, coverage unions on the device-resident masks):
    # every file except the JSON side files must come out the same
    wd2 = tmp_path / "deep"
    got = mc.run_chain(str(wd2), fastas, library=hip_lib, deep_rows=0, primer_length=k)
    for fn, rec in g["files"].items():
        if "text" in rec:
            assert got[fn].decode() == rec["text"], fn


@pytest.mark.gpu
@pytest.mark.parametrize("k,n", [(18, 16), (34, 6)])
def test_config5_scale_chain_hip_equals_oracle(k, n, hip_lib, oracle_lib, tmp_path):
    from multiprime_amd.synth import synth_block, to_fasta
    seed = 20250303
    rng = np.random.default_rng(seed)
    sizes = np.exp(rng.uniform(np.log(500), np.log(5000), size=n)).astype(int)
    cols = rng.integers(600, 1200, size=n)
    fastas = {}
    for i in range(n):
        fa = tmp_path / f"Cluster_{i}.tfa"
        fa.write_bytes(to_fasta(synth_block(0, int(sizes[i]), int(cols[i]), seed + 1000 * (i + 1))))
        fastas[f"Cluster_{i}"] = (str(fa), int(sizes[i]))
    assert (sizes > mc.DEEP_ROWS).any() and (sizes <= mc.DEEP_ROWS).any()          # both routes are taken
    hip = mc.run_chain(str(tmp_path / "hip"), fastas, library=hip_lib, primer_length=k)
    ora = mc.run_chain(str(tmp_path / "oracle"), fastas, library=oracle_lib, primer_length=k)
    assert mc.compare_chains(hip, ora) == []
    if k == 18:
        assert len(hip["final_maxprimers_set.xls"].splitlines()) > 8                 # most clusters reach the final set
        assert len(hip) >= 4 * n + 7
