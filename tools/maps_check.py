#!/usr/bin/env python3
"""Which HIP / HSA runtimes end up in one process (GPU box)?  `python tools/maps_check.py torch-first|lib-first`.
torch brings its own libamdhip64 / libhsa-runtime64 under torch/lib; libmprime_hip.so is linked against /opt/rocm's.  Whichever is
loaded first serves both when the sonames agree — or two runtimes share the process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1] if len(sys.argv) > 1 else "torch-first"


def mapped():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if any(s in l for s in ("amdhip64", "hsa-runtime", "librccl", "libmprime"))})


if order == "torch-first":
    import torch
    print("torch sees a GPU:", torch.cuda.is_available())
    from multiprime_amd._abi import Library
    ctx = Library().context(0)
    print("library context: ok")
else:
    from multiprime_amd._abi import Library
    ctx = Library().context(0)
    print("library context: ok")
    import torch
    try:
        print("torch sees a GPU:", torch.cuda.is_available())
        torch.zeros(4, device="cuda")
        print("torch allocates: ok")
    except Exception as e:
        print("torch:", type(e).__name__, str(e)[:120])
print("\n".join(mapped()))
