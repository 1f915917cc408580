#!/usr/bin/env python3
"""The exception records' path under stress (GPU box): many short-lived contexts, each builds its windows and then runs the histograms on the
calling thread WHILE a helper thread collects the exception list (mp_get_exceptions -> ex_fetch: a copy on the library's own stream, a sort) —
what NN_degenerate.run() does once per alignment, a few hundred times per process, torch alive beside it.  Exit status of the child = the finding
(round 6: one `bench.py` run in four died with std::bad_variant_access from inside the runtime while ex_fetch still REGISTERED its landing buffer
on the helper thread)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, threading
sys.path.insert(0, %r)
import numpy as np
import torch
from multiprime_amd._abi import Library
from multiprime_amd.synth import synth_block
lib = Library()
t_dev = torch.empty(16 << 20, dtype=torch.uint8, device="cuda")
n_iter, rows, fresh = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
blk = synth_block(0, rows, 400, 20250303, p_iupac=2e-4)
data = np.ascontiguousarray(blk).reshape(-1)
off = np.arange(rows + 1, dtype=np.int64) * blk.shape[1]
ctx = None
for i in range(n_iter):
    if ctx is None or i %% fresh == 0:
        if ctx is not None:
            ctx.close()
        ctx = lib.context(0)
        ctx.load_msa(data, off)
    n_ex = ctx.build_windows(2, 380 - (i %% 5), 18, 1)
    box = {}
    th = threading.Thread(target=lambda: box.update(r=ctx.get_exceptions(n_ex)))
    th.start()
    ctx.set_entropy_gate(3.6 if i %% 2 else 0.0)
    ctx.window_unique_device()
    th.join()
    assert len(box["r"][0]) == n_ex and n_ex > 0
    if i %% 3 == 0:
        t_dev.add_(1)
ctx.close()
torch.cuda.synchronize()
print("done", n_ex)
''' % REPO

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    for rows, fresh in ((20000, 1), (131072, 1), (20000, 4), (131072, 3)):
        r = subprocess.run([sys.executable, "-c", CHILD, str(n), str(rows), str(fresh)], capture_output=True, text=True, timeout=1500,
                           env=dict(os.environ, MP_DEBUG_TERMINATE="1"))
        print(json.dumps({"rows": rows, "a_fresh_context_every": fresh, "iterations": n, "exit": r.returncode, "finished": "done" in r.stdout,
                          "stderr_tail": r.stderr[-3000:] if r.returncode else ""}), flush=True)
