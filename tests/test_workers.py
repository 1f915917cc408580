"""The host stages' kept threads (csrc/workers.hpp): a parallel region gives the same answer on kept threads, on threads of its own
(MP_HOST_POOL=0, a separate process), from two Python threads at once (the second finds the pool taken) and in a forked child (which has
none of the parent's threads) — the shapes --batch-procs and the core step's helper thread produce."""
import os
import subprocess
import sys
import threading

import numpy as np

from multiprime_amd import host

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(seed, n=40000, k=18, n_primers=50):
    rng = np.random.default_rng(seed)
    xc = rng.integers(0, 16, size=(n, k), dtype=np.uint8)
    primers = rng.integers(1, 16, size=(n_primers, k), dtype=np.uint8)
    primer_of = rng.integers(0, n_primers, size=n).astype(np.int64)
    return xc, primer_of, primers


def _verdicts(case):
    xc, primer_of, primers = case
    return host.exception_verdicts(xc, primer_of, primers, 1, 0b110, 0b011 << 15)


def _numpy_verdicts(case, v=1, sF=0b110, sR=0b011 << 15):
    xc, primer_of, primers = case
    pr = primers[primer_of]
    can = (xc == 0) | ((xc & ~pr & 15) != 0)
    both = ((xc == 0).sum(axis=1) > v) | (can.sum(axis=1) > v)
    k = xc.shape[1]
    pos = np.arange(k)
    f = (can & (((sF >> pos) & 1) == 1)).any(axis=1)
    r = (can & (((sR >> pos) & 1) == 1)).any(axis=1)
    return np.stack([both | f, both | r], axis=1)


def test_region_on_kept_threads_matches_numpy():
    case = _case(1)
    want = _numpy_verdicts(case)
    for _ in range(3):                                   # the first call starts the threads, the others find them
        assert np.array_equal(_verdicts(case), want)


def test_two_python_threads_at_once():
    cases = [_case(s) for s in (2, 3, 4, 5)]
    want = [_numpy_verdicts(c) for c in cases]
    got = [None] * len(cases)

    def work(i):
        for _ in range(5):
            got[i] = _verdicts(cases[i])

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_forked_child_starts_its_own_threads():
    case = _case(6)
    want = _numpy_verdicts(case)
    assert np.array_equal(_verdicts(case), want)         # the parent's threads exist now
    pid = os.fork()
    if pid == 0:                                         # the child has only the forking thread
        ok = False
        try:
            ok = all(np.array_equal(_verdicts(case), want) for _ in range(3))
        finally:
            os._exit(0 if ok else 1)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
    assert np.array_equal(_verdicts(case), want)         # and the parent's are still there


def test_switch_off():
    code = ("import numpy as np, sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_workers as t; c = t._case(7); "
            "assert np.array_equal(t._verdicts(c), t._numpy_verdicts(c)); print('ok')" % (REPO, os.path.join(REPO, "tests")))
    env = dict(os.environ, MP_HOST_POOL="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_beside_hands_the_helpers_exception_to_the_caller():
    """core._Beside (the core step's helper threads: coverage bitsets, self-dimer launch): join() re-raises on the calling thread."""
    import pytest
    from multiprime_amd.core import _Beside

    def boom(x):
        raise ValueError("from the helper: %d" % x)

    for beside in (True,):
        b = _Beside(boom, 7, beside=beside)
        with pytest.raises(ValueError, match="from the helper: 7"):
            b.join()
        b.join()                                           # the error is handed over once
    with pytest.raises(ValueError):
        _Beside(boom, 7, beside=False)                       # not beside: raised where it happens
    seen = []
    ok = _Beside(seen.append, 3)
    ok.join()
    assert seen == [3]
