"""The C-ABI libraries load and export every symbol include/mprime.h declares (no compute
calls: there is no GPU in the authoring container)."""
import ctypes
import os
import re

import pytest

from conftest import REPO
from multiprime_amd import _abi


def header_symbols():
    src = open(os.path.join(REPO, "include", "mprime.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mp_[a-z_]+)\s*\(", src)))


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


def test_product_never_reaches_into_the_oracle():
    pkg = os.path.join(REPO, "multiprime_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                for needle in ("libmprime_oracle", "import oracle", "from oracle", "oracle/_build"):
                    assert needle not in txt, (f, needle)
