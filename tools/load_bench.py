#!/usr/bin/env python3
"""Times the way an alignment reaches the device: library / context start-up, native FASTA parse (MP_HOST_TRACE=1 prints its
stages), the joined copy, mp_load_msa cold and warm, and the two ways to the primer region (per-row arrays + np.quantile,
device histograms) on a synthetic rows x cols FASTA file.  One JSON line."""
import argparse
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multiprime_amd._abi import prefer_staged_copies  # noqa: E402
prefer_staged_copies()            # a program's own decision, before the HIP runtime starts (multiprime_amd/_abi.py)
from multiprime_amd import host, msa  # noqa: E402
from multiprime_amd._abi import Library  # noqa: E402
from multiprime_amd.synth import synth_block, to_fasta  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1048576)
ap.add_argument("--cols", type=int, default=1000)
a = ap.parse_args()
res = {"rows": a.rows, "cols": a.cols}
with tempfile.TemporaryDirectory() as td:
    fa_path = os.path.join(td, "syn.fa")
    with open(fa_path, "wb") as f:
        f.write(to_fasta(synth_block(0, a.rows, a.cols, 20250303)))

    def timed(name, fn):
        t0 = time.time()
        out = fn()
        res[name] = round(time.time() - t0, 4)
        return out

    lib = timed("library_s", Library)
    ctx = timed("context_s", lambda: lib.context(0))
    fa = timed("parse_s", lambda: host.Fasta(fa_path))
    timed("parse_again_s", lambda: host.Fasta(fa_path))
    data, off = timed("rows_copy_s", fa.rows)
    timed("load_msa_cold_s", lambda: ctx.load_msa(data, off))
    timed("load_msa_warm_s", lambda: ctx.load_msa(data, off))
    lead, rstrip, _ = timed("row_attributes_s", ctx.row_attributes)
    timed("region_quantiles_s", lambda: msa.region(lead, rstrip, 0.8))
    lh, rh = timed("row_histograms_s", lambda: ctx.row_histograms(a.cols + 1))
    timed("region_from_histograms_s", lambda: msa.region_from_histograms(lh, rh, 0.8))
print(json.dumps(res))
