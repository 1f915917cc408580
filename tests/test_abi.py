"""The C-ABI libraries load and export every symbol include/mprime.h declares (no compute
calls: there is no GPU in the authoring container)."""
import ctypes
import os
import re

import pytest

from conftest import REPO
from multiprime_amd import _abi


def header_symbols(name="mprime.h"):
    src = open(os.path.join(REPO, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", src)))


def test_binding_covers_header():
    assert sorted(n for n, _, _ in _abi.SYMBOLS) == header_symbols()


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return g


def test_hip_library_exports_abi(built):
    dll = ctypes.CDLL(built.HIP_SO)
    for name in header_symbols():
        assert hasattr(dll, name), name
    dll.mp_backend_name.restype = ctypes.c_char_p
    assert dll.mp_backend_name() == b"hip"


def test_host_stage_binding_and_exports(built):
    """include/mprime_host.h (native host stage): the ctypes table covers it and the PRODUCT library exports it; the
    oracle library does not serve it (its checker is oracle/core_ref.py + the golden traces)."""
    from multiprime_amd import host
    want = header_symbols("mprime_host.h")
    assert sorted(n for n, _, _ in host.HOST_SYMBOLS) == want and len(want) >= 19
    dll = ctypes.CDLL(built.HIP_SO)
    for name in want:
        assert hasattr(dll, name), name
    assert host.dll() is not None                       # loads and resolves on a box without a GPU


def test_oracle_library_exports_abi(built):
    dll = ctypes.CDLL(built.ORACLE_SO)
    for name in header_symbols():
        assert hasattr(dll, name), name


def test_product_fails_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _abi.Library()            # loads fine: hipcc cross-compiled it
    with pytest.raises(_abi.MprimeError):
        lib.context(0)              # no device -> error, never a CPU fallback


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(_abi.MprimeError):
        _abi.Library(str(tmp_path / "libmprime_hip.so"))


def _calls_in_wrong_order(lib):
    """Every entry point that needs earlier state must refuse with a negative code and a message, not crash."""
    import numpy as np
    from multiprime_amd._abi import MprimeError
    ctx = lib.context(0)
    for call in (lambda: ctx.build_windows(0, 4, 8, 1), lambda: ctx.row_attributes()):
        with pytest.raises(MprimeError, match="no alignment loaded"):
            call()
    data = np.frombuffer(b"ACGTACGTACGTACGTACGTACGTAC" * 3, np.uint8)
    ctx.load_msa(data, np.array([0, 26, 52, 78], np.int64))
    ctx.n_win, ctx.k = 4, 8                                     # what build_windows would have set on the wrapper
    for call in (lambda: ctx.window_stats(), lambda: ctx.window_unique(),
                 lambda: ctx.eval_candidates(np.zeros(1, np.int32), np.ones((1, 8), np.uint8), 0, 0)):
        with pytest.raises(MprimeError, match="no windows built"):
            call()
    with pytest.raises(MprimeError):
        ctx.build_windows(0, 4, 40, 1)                          # k > MP_MAX_K
    with pytest.raises(MprimeError):
        ctx.build_windows(0, 400, 8, 1)                         # windows past the longest row
    assert ctx.build_windows(0, 4, 8, 1) == 0
    with pytest.raises(MprimeError, match="ascending"):
        ctx.eval_candidates(np.array([2, 1], np.int32), np.ones((2, 8), np.uint8), 0, 0)
    freq, nn = ctx.window_stats()
    assert freq.sum() == 4 * 3 * 8 and nn.sum() == 4 * 3 * 7    # 4 windows x 3 sequences x k (k - 1) symbols


def test_oracle_refuses_calls_in_the_wrong_order(oracle_lib):
    _calls_in_wrong_order(oracle_lib)


@pytest.mark.gpu
def test_hip_refuses_calls_in_the_wrong_order(hip_lib):
    _calls_in_wrong_order(hip_lib)


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py parses, imports and then stops with a message on a box without a GPU (there is no CPU path to time)."""
    import subprocess
    import sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2"], capture_output=True, text=True, timeout=300)
    if r.returncode == 0:
        pytest.skip("a GPU is present")
    assert "needs a GPU" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_tools_and_scripts_compile():
    """The GPU-box tools cannot run here; at least they must be valid Python."""
    import glob
    import py_compile
    from conftest import REPO
    files = glob.glob(os.path.join(REPO, "tools", "*.py")) + glob.glob(os.path.join(REPO, "scripts", "*.py"))
    assert len(files) >= 12
    for f in files:
        py_compile.compile(f, doraise=True)


def test_product_never_reaches_into_the_oracle():
    """The oracle is test infrastructure: no module of the package and no drop-in script imports it or names its library
    (only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may)."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "multiprime_amd", "*.py")) + glob.glob(os.path.join(root, "scripts", "*.py"))
    assert len(files) > 15
    for path in files:
        src = open(path).read()
        tree = ast.parse(src)
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n == "oracle" or n.startswith("oracle.") for n in names), path
            if isinstance(node, ast.Constant) and isinstance(node.value, str) and node is not ast.get_docstring:
                assert "libmprime_oracle" not in node.value, path
    for path in glob.glob(os.path.join(root, "multiprime_amd", "csrc", "*")):
        if os.path.isfile(path) and path.endswith((".hip", ".cpp", ".hpp")):
            assert "oracle/" not in open(path).read().replace("oracle/core_ref.py", ""), path


def test_implicit_loader_refuses_a_backend_that_is_not_hip(oracle_lib, monkeypatch):
    """Library() without a path is the product's loader: no environment variable can put the CPU checker behind a drop-in command.
    Library(path) — tests, tools — loads what it is told to; the CLI tests patch the class from tests/checker_shim/."""
    from multiprime_amd._abi import Library, MprimeError
    monkeypatch.setenv("MPRIME_LIBRARY", oracle_lib.path)
    monkeypatch.setenv("MPRIME_TEST_CHECKER_BACKEND", "1")          # the round-5 hook: gone
    monkeypatch.setenv("MP_TEST_CHECKER_SO", oracle_lib.path)       # read by the tests' shim only, never by the package
    with pytest.raises(MprimeError, match="not 'hip'"):
        Library()
    assert Library(oracle_lib.path).backend != "hip"
    import multiprime_amd
    for root, _, files in os.walk(os.path.dirname(multiprime_amd.__file__)):
        for name in files:
            if name.endswith((".py", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(root, name)).read()
                assert "MPRIME_TEST_CHECKER_BACKEND" not in text and "MP_TEST_CHECKER_SO" not in text, name
