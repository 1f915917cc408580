#!/usr/bin/env python3
"""Round 4's crash site under stress (GPU box): mp_load_msa with the caller's residue bytes registered for the transfer
(MP_EXPERIMENT_PIN_LOAD=1), 200 alignments of 1 .. 150 MB from arrays the Python allocator owns (fresh, sliced, reused addresses),
torch imported and copying beside it, two contexts alive, one worker thread loading at the same time.  Exit status of the child = the finding."""
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
lib = Library()
rng = np.random.default_rng(int(sys.argv[1]))
t_dev = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
def loads(n_iter, seed):
    r = np.random.default_rng(seed)
    ctx = lib.context(0)
    for i in range(n_iter):
        rows, L = int(r.integers(1000, 150000)), int(r.integers(200, 1000))
        a = np.full(rows * L + 4096, 65, np.uint8)
        off0 = int(r.integers(0, 4096))
        data = a[off0:off0 + rows * L]
        ctx.load_msa(data, np.arange(rows + 1, dtype=np.int64) * L)
        if i %% 7 == 0:
            t_dev[: min(len(data), 64 << 20)].copy_(torch.from_numpy(data[: 64 << 20]), non_blocking=True)
        if i %% 13 == 0:
            ctx.close(); ctx = lib.context(0)
        del a, data
    ctx.close()
th = threading.Thread(target=loads, args=(int(sys.argv[2]) // 2, 7))
th.start()
loads(int(sys.argv[2]), 3)
th.join()
torch.cuda.synchronize()
print("done")
''' % REPO

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for pin in ("1", "0"):
        for seed in (1, 2):
            env = dict(os.environ, MP_EXPERIMENT_PIN_LOAD=pin) if pin == "1" else {k: v for k, v in os.environ.items() if k != "MP_EXPERIMENT_PIN_LOAD"}
            r = subprocess.run([sys.executable, "-c", CHILD, str(seed), str(n)], capture_output=True, text=True, timeout=1200, env=env)
            print(json.dumps({"registered_for_the_transfer": pin == "1", "seed": seed, "loads": n + n // 2, "exit": r.returncode,
                              "finished": "done" in r.stdout, "stderr_tail": r.stderr[-400:] if r.returncode else ""}), flush=True)
