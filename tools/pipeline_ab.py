#!/usr/bin/env python3
"""bench.py's `pipeline` block alone (NN_degenerate.run() on the synthetic rows, context kept, median of 5, TSV against the checker's hash):
   tools/pipeline_ab.py [rows ...]      one JSON line per size.  For A/B runs of host-side changes (environment switches of the library)."""
import json
import os
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multiprime_amd._abi import Library, prefer_staged_copies  # noqa: E402
prefer_staged_copies()                            # before the HIP runtime starts, as bench.py does
import bench  # noqa: E402
from multiprime_amd.synth import synth_block  # noqa: E402

a = types.SimpleNamespace(seed=20250303)
lib = Library()
for n in [int(x) for x in sys.argv[1:]] or [131072]:
    rows = synth_block(0, n, 1000, a.seed)
    r = bench.pipeline_block(lib, 0, rows, a)
    print(json.dumps({"rows": n, "run_ms": round(r["run_ms"], 3), "min": round(r["run_ms_min"], 3), "max": round(r["run_ms_max"], 3),
                      "construct_ms": round(r["construct_ms"], 2), "tsv_equal_oracle": r["tsv_equal_oracle"], "phases_ms": r["phases_ms"]}), flush=True)
