#!/usr/bin/env python3
"""Where the drop-in command's process time goes on the GPU box (debug aid): interpreter, imports, library load, context creation,
the run itself, interpreter exit.  usage: python tools/cli_startup.py"""
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timed(cmd, env=None):
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
        best = min(best, time.time() - t0)
    return round(best, 3)


py = sys.executable
print("python -c pass                      ", timed([py, "-c", "pass"]))
print("import numpy                        ", timed([py, "-c", "import numpy"]))
print("import multiprime_amd.cli           ", timed([py, "-c", f"import sys; sys.path.insert(0, {REPO!r}); import multiprime_amd.cli"]))
print("  + Library()                       ", timed([py, "-c", f"import sys; sys.path.insert(0, {REPO!r}); from multiprime_amd._abi import Library; Library()"]))
print("  + context(0)                      ", timed([py, "-c", f"import sys; sys.path.insert(0, {REPO!r}); from multiprime_amd._abi import Library; Library().context(0)"]))
print("  + context(0), os._exit            ", timed([py, "-c", f"import sys, os; sys.path.insert(0, {REPO!r}); from multiprime_amd._abi import Library; c = Library().context(0); os._exit(0)"]))
