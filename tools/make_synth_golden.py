#!/usr/bin/env python3
"""The core step's TSV on the bench's synthetic alignment as the CHECKER computes it (oracle/core_ref.py: the pure-Python per-window
logic, over oracle/mprime_oracle.c: the plain-C restatement of the reference's O(N) loops) — minutes to hours on one core, so it is
run once in the authoring container and its SHA-256 is committed (tests/golden/synth_pipeline.json); bench.py's `pipeline` block and
tests/test_scale_parity.py compare the product's TSV on the same generated rows with it.

    python tools/make_synth_golden.py --rows 131072          # adds / replaces the entry for that size
The generator is multiprime_amd.synth.synth_block(0, rows, cols, 20250303) (SURVEY 8d input 4), flags as tools/pipeline_scale.py."""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multiprime_amd._abi import Library  # noqa: E402
from multiprime_amd.synth import synth_block, to_fasta  # noqa: E402
from oracle.core_ref import NN_degenerate  # noqa: E402

FLAGS = dict(primer_length=18, coverage=0.8, number_of_dege_bases=4, score_of_dege_bases=10, raw_entropy_threshold=3.6, product_len=150,
             position="2,3,-1", variation=1, distance=4, GC="0.2,0.7", nproc=1)

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, required=True)
ap.add_argument("--cols", type=int, default=1000)
ap.add_argument("--seed", type=int, default=20250303)
a = ap.parse_args()
path = os.path.join(REPO, "tests", "golden", "synth_pipeline.json")
db = json.load(open(path)) if os.path.exists(path) else {"flags": FLAGS, "generator": "multiprime_amd.synth.synth_block(0, rows, cols, seed)", "entries": []}
lib = Library(os.path.join(REPO, "oracle", "_build", "libmprime_oracle.so"))
with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
    rows = synth_block(0, a.rows, a.cols, a.seed)
    fa = os.path.join(td, "syn.fa")
    with open(fa, "wb") as f:
        f.write(to_fasta(rows))
    del rows
    t0 = time.time()
    app = NN_degenerate(seq_file=fa, outfile=os.path.join(td, "out.tsv"), library=lib, write_json=False, **FLAGS)
    app.run()
    wall = time.time() - t0
    tsv = open(os.path.join(td, "out.tsv"), "rb").read()
entry = {"rows": a.rows, "cols": a.cols, "seed": a.seed, "tsv_sha256": hashlib.sha256(tsv).hexdigest(), "tsv_lines": tsv.count(b"\n"), "tsv_bytes": len(tsv),
         "first_rows": tsv.decode().splitlines()[:3], "checker_wall_s": round(wall, 1), "checker": "oracle/core_ref.py over oracle/mprime_oracle.c, one core"}
db["entries"] = [e for e in db["entries"] if (e["rows"], e["cols"], e["seed"]) != (a.rows, a.cols, a.seed)] + [entry]
db["entries"].sort(key=lambda e: (e["rows"], e["cols"]))
with open(path, "w") as f:
    json.dump(db, f, indent=1)
print(json.dumps(entry))
