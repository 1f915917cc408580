import gzip
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "slow: minutes on CPU")


def load_gz_json(name):
    with gzip.open(os.path.join(GOLDEN, name), "rb") as f:
        return json.loads(f.read())


def golden_input(name):
    """Decompressed bytes of an input alignment stored under tests/golden/inputs/."""
    with gzip.open(os.path.join(GOLDEN, "inputs", name + ".gz"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle behind the same C ABI (test infrastructure; built on demand)."""
    if os.environ.get("MP_ORACLE_LIB"):                       # tools/sanitize_host.sh: an ASAN/UBSAN build of the oracle
        from multiprime_amd._abi import Library
        return Library(os.environ["MP_ORACLE_LIB"])
    so = os.path.join(REPO, "oracle", "_build", "libmprime_oracle.so")
    src = os.path.join(REPO, "oracle", "mprime_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")], stdout=subprocess.DEVNULL)
    from multiprime_amd._abi import Library
    return Library(so)


@pytest.fixture(scope="session")
def hip_lib():
    """The product library (hand-written HIP).  Rebuilt with hipcc when sources are newer than the .so."""
    import __graft_entry__ as g
    g.build()
    from multiprime_amd._abi import Library
    return Library()
