"""SURVEY §8c end-to-end known answer: this build's core output, fed through the REFERENCE's own
pairing script (get_multiPrime.py, unmodified) and then through this build's get_Maxprimerset,
must select the reference's final primer pair.  Needs /root/reference (authoring container only)."""
import os
import subprocess
import sys
import types

import pytest

from conftest import golden_input, load_gz_json
from multiprime_amd import maxset
from multiprime_amd.core import NN_degenerate

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not present")


def test_core_to_reference_pairing_to_primer_set(oracle_lib, tmp_path):
    meta = load_gz_json("cluster0_v1.trace.json.gz")["meta"]
    fl = meta["flags"]
    inp = tmp_path / "Cluster_0_20727.tmsa"
    inp.write_bytes(golden_input(meta["input"]))
    out = tmp_path / "Cluster_0_20727.top.primer.out"
    NN_degenerate(seq_file=str(inp), primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"],
                  score_of_dege_bases=fl["d"], raw_entropy_threshold=fl["e"], product_len=fl["s"], position=fl["c"],
                  variation=fl["v"], distance=fl["a"], GC=fl["g"], nproc=1, outfile=str(out), library=oracle_lib).run()
    tfa = os.path.join(REF, "test_data", "results", "Clusters_fa", "Cluster_0_20727.tfa")
    cand = tmp_path / "Cluster_0_20727.candidate.primers.txt"
    subprocess.check_call([sys.executable, os.path.join(REF, "scripts", "get_multiPrime.py"), "-i", str(out), "-r", tfa,
                           "-f", "0.7", "-s", "150,1200", "-g", "0.2,0.7", "-e", "4", "-d", "4", "-a",
                           "TCTTTCCCTACACGACGCTCTTCCGATCT,TGGAGTTCAGACGTGTGCTCTTCCGATCT", "-m", "0", "-o", str(cand), "-p", "1"],
                          stdout=subprocess.DEVNULL, cwd=str(tmp_path))
    fields = [x for x in cand.read_text().strip().split("\t") if x]
    assert (len(fields) - 1) // 5 == 2749
    assert fields[1:6] == ["RRTCAGATGCACCYATTG", "CCCAKRTCYTCAGCATTT", "566:51.59:0.968", "484", "888:1453"]
    final = tmp_path / "final_maxprimers_set.xls"
    maxset.run(types.SimpleNamespace(input=str(cand), step=5, method="T", out=str(final), device=0), library=oracle_lib)
    rows = final.read_text().splitlines()
    assert len(rows) == 2 and rows[1].split("\t")[2:4] == ["RRTCAGATGCACCYATTG", "CCCAKRTCYTCAGCATTT"]
