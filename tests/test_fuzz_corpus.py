"""Fuzz corpus recorded from the reference (tests/golden/make_golden_fuzz.py): 64 small random
alignments x random flag sets (k 10..24, v 0..3, d, n, f, c, e, gc, ragged rows, IUPAC, junk
characters).  Exit code, TSV bytes and both JSON side files (canonical digests) must match."""
import gzip
import hashlib
import json
import os

import pytest

from conftest import GOLDEN
from multiprime_amd.core import NN_degenerate


def _cases():
    return json.loads(gzip.open(os.path.join(GOLDEN, "fuzz.json.gz")).read())


def canon(obj):
    return hashlib.sha256(json.dumps(obj, sort_keys=True, separators=(",", ":")).encode()).hexdigest()


def canon_noncov(d):
    return {str(k): [{km: sorted(ids) for km, ids in sorted(side.items())} for side in v] for k, v in d.items()}


def canon_gap(d):
    return {str(k): {km: list(ids) for km, ids in sorted(v.items())} for k, v in d.items()}


def replay(lib, tmp_path):
    n_rows = 0
    for rec in _cases():
        fl = rec["flags"]
        inp = tmp_path / f"c{rec['seed']}.fa"
        inp.write_bytes(rec["fasta"].encode("latin-1"))
        out = tmp_path / f"c{rec['seed']}.out"
        rc = 0
        try:
            NN_degenerate(seq_file=str(inp), primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"],
                          score_of_dege_bases=fl["d"], raw_entropy_threshold=fl["e"], product_len=fl["s"], position=fl["c"],
                          variation=fl["v"], distance=fl["a"], GC=fl["g"], nproc=1, outfile=str(out), library=lib).run()
        except SystemExit as e:
            rc = e.code
        assert rc == rec["returncode"], (rec["seed"], fl)
        if rc == 0:
            assert out.read_text() == rec["tsv"], (rec["seed"], fl)
            assert canon(canon_noncov(json.load(open(str(out) + ".non_coverage_seq_id_json")))) == rec["noncov_sha"], rec["seed"]
            assert canon(canon_gap(json.load(open(str(out) + ".gap_seq_id_json")))) == rec["gap_sha"], rec["seed"]
            n_rows += rec["tsv"].count("\n") - 1
    assert n_rows > 2000


def test_fuzz_corpus_matches_reference(oracle_lib, tmp_path, capsys):
    replay(oracle_lib, tmp_path)


@pytest.mark.gpu
def test_fuzz_corpus_hip_matches_reference(hip_lib, tmp_path, capsys):
    replay(hip_lib, tmp_path)
