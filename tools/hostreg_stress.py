#!/usr/bin/env python3
"""What made `hipHostRegister` of caller-owned memory inside mp_load_msa kill one GPU-suite run in four (round 4, commit c373106: SIGABRT,
no message)?  Each mode below runs in a process of its own (an abort must not take the others with it): N iterations of
register -> hipMemcpyAsync host-to-device -> synchronize -> unregister on a buffer of the kind named, through libamdhip64 directly.

    python tools/hostreg_stress.py [--iterations 200]        # on the GPU box; prints one JSON line per mode
modes: numpy (touched anonymous memory, page-aligned or not, sizes 4 .. 160 MB), numpy_fresh (pages never touched), file_ro (a read-only
private mapping of a file: what host.Fasta hands over when it maps the FASTA), file_ro_copy (np.array of it), threads (two threads,
neighbouring unaligned buffers sharing a page), early_unregister (unregister while the copy is still in flight), twice (a range that is
already registered)."""
import argparse
import ctypes as C
import json
import mmap
import os
import subprocess
import sys
import tempfile
import threading

import numpy as np

MODES = ["numpy", "numpy_fresh", "file_ro", "file_ro_copy", "threads", "early_unregister", "twice", "overlap", "after_runtime_pin", "after_runtime_pin_default"]


def run_mode(mode, n_iter):
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    hip.hipHostUnregister.argtypes = [C.c_void_p]
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipSetDevice(0)
    dev = C.c_void_p()
    cap = 192 << 20
    assert hip.hipMalloc(C.byref(dev), cap) == 0
    rng = np.random.default_rng(1)
    stats = {"registered": 0, "refused": 0, "copy_errors": 0}

    def cycle(addr, nbytes, sync_first=True):
        rc = hip.hipHostRegister(addr, nbytes, 0)
        if rc != 0:
            stats["refused"] += 1
            hip.hipGetLastError()
            return
        stats["registered"] += 1
        if hip.hipMemcpyAsync(dev, addr, nbytes, 1, None) != 0:
            stats["copy_errors"] += 1
        if sync_first:
            hip.hipStreamSynchronize(None)
        hip.hipHostUnregister(addr)
        if not sync_first:
            hip.hipStreamSynchronize(None)

    if mode in ("numpy", "numpy_fresh", "early_unregister", "twice"):
        for i in range(n_iter):
            n = int(rng.integers(4 << 20, 160 << 20))
            a = np.empty(n + 4096, np.uint8)
            if mode != "numpy_fresh":
                a[:] = i & 255
            off = int(rng.integers(0, 4096)) if i % 2 else 0                      # unaligned start every other time
            addr = a.ctypes.data + off
            if mode == "twice":
                hip.hipHostRegister(addr, n, 0)
            cycle(addr, n, sync_first=(mode != "early_unregister"))
            if mode == "twice":
                hip.hipHostUnregister(addr)
                hip.hipGetLastError()
            del a
    elif mode == "overlap":
        # a second registration that OVERLAPS a live one without being the same range (a neighbour's buffer sharing pages)
        for i in range(n_iter):
            n = int(rng.integers(8 << 20, 64 << 20))
            a = np.zeros(n + (1 << 20), np.uint8)
            shift = int(rng.integers(1, 1 << 20))
            r1 = hip.hipHostRegister(a.ctypes.data, n, 0)
            cycle(a.ctypes.data + shift, n)
            if r1 == 0:
                hip.hipHostUnregister(a.ctypes.data)
            hip.hipGetLastError()
            del a
    elif mode in ("after_runtime_pin", "after_runtime_pin_default"):
        # what the GPU suite does between two mp_load_msa calls: large UNREGISTERED async copies (the runtime page-locks the caller's range
        # itself for sizes above GPU_PINNED_MIN_XFER_SIZE and may keep that pin cached), the array is freed, the allocator hands the same
        # addresses to the next array — which is then registered explicitly over a range that overlaps the runtime's cached one
        for i in range(n_iter):
            n = int(rng.integers(4 << 20, 150 << 20))
            a = np.zeros(n, np.uint8)
            hip.hipMemcpyAsync(dev, a.ctypes.data, n, 1, None)
            hip.hipStreamSynchronize(None)
            del a
            n2 = int(rng.integers(4 << 20, 150 << 20))
            b = np.zeros(n2 + 4096, np.uint8)
            cycle(b.ctypes.data + int(rng.integers(0, 4096)), n2)
            del b
    elif mode in ("file_ro", "file_ro_copy"):
        with tempfile.NamedTemporaryFile(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as f:
            f.write(os.urandom(64 << 20))
            f.flush()
            for i in range(n_iter):
                m = mmap.mmap(f.fileno(), 0, prot=mmap.PROT_READ, flags=mmap.MAP_PRIVATE)
                view = np.frombuffer(m, np.uint8)
                if mode == "file_ro_copy":
                    view = np.array(view)
                cycle(view.ctypes.data, (48 << 20) + int(rng.integers(0, 1 << 20)))
                del view
                m.close()
    elif mode == "threads":
        big = np.zeros((96 << 20) + 8192, np.uint8)
        half = (48 << 20) + 1234                                                 # the two halves share a page

        def worker(lo, n):
            for _ in range(n_iter):
                cycle(big.ctypes.data + lo, n)

        th = [threading.Thread(target=worker, args=(0, half)), threading.Thread(target=worker, args=(half, (48 << 20) - 1234))]
        for t in th:
            t.start()
        for t in th:
            t.join()
    print(json.dumps({"mode": mode, **stats}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=200)
    ap.add_argument("--mode", default=None)
    a = ap.parse_args()
    if a.mode:
        run_mode(a.mode, a.iterations)
        sys.exit(0)
    for mode in MODES:
        env = dict(os.environ)
        if mode == "after_runtime_pin":
            env["GPU_PINNED_MIN_XFER_SIZE"] = "1"                                # every copy above 1 MB is pinned by the runtime itself
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--mode", mode, "--iterations", str(a.iterations)], capture_output=True, text=True,
                           timeout=900, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        out = json.loads(line[-1]) if line else {"mode": mode}
        out["exit"] = r.returncode                                               # -6 = SIGABRT
        if r.returncode:
            out["stderr_tail"] = r.stderr[-600:]
        print(json.dumps(out), flush=True)
