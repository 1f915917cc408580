"""The get_degePrimer.py drop-in (multiprime_amd/degepair.py) against outputs recorded from the unmodified reference
script (tests/golden/make_golden_degepair.py) on the DEGEPRIME table shipped with the reference, four flag sets:
candidate file byte for byte, stdout lines."""
import contextlib
import gzip
import io
import json
import os

import pytest

from conftest import GOLDEN
from multiprime_amd import degepair


@pytest.fixture(scope="module")
def gold():
    return json.loads(gzip.open(os.path.join(GOLDEN, "degepair.json.gz")).read())


@pytest.mark.parametrize("flagset", ["yaml", "default", "loose", "tight"])
def test_degepair_matches_reference(flagset, gold, tmp_path):
    want = gold["results"]["dege10"][flagset]
    inp = tmp_path / "in.dege.out"
    inp.write_bytes(gzip.open(os.path.join(GOLDEN, "inputs", "degeprime_dege10.out.gz")).read())
    ref = tmp_path / "ref.fa"
    ref.write_text("".join(f">s{i}\nACGT\n" for i in range(1000)))
    out = tmp_path / "cand.txt"
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        degepair.main(["-i", str(inp), "-r", str(ref), "-o", str(out)] + gold["flags"][flagset])
    lines = [l for l in buf.getvalue().splitlines() if not l.startswith("INFO ")]
    assert lines == want["stdout"]
    assert out.read_text().replace(str(out), "<OUT>") == want["txt"]
    assert (len(want["txt"].split("\t")) - 2) // 5 >= 9          # the cases are not vacuous


def test_degepair_dimer_check_is_dead_code_in_the_reference():
    """current_end() of the reference never fills its set (`end_seq.union(...)` without assignment, GD:319-325): every
    pair passes its dimer test, which is why the drop-in's dimer_check is a constant."""
    assert degepair.Primers_filter.dimer_check("ACGTACGTACGTACGTAC", "GTACGTACGTACGTACGT") is False
