"""TEST INFRASTRUCTURE — not product code.  The CPU-only CLI tests (tests/test_cli_multirank.py) run the drop-in command lines in
child processes on a box without a GPU; the device calls of those children go to the ABI checker (oracle/).  The product loader
(`multiprime_amd._abi.Library()`) has no branch that can load a non-HIP backend, so the tests put THIS directory on the children's
PYTHONPATH: the interpreter imports `sitecustomize` at start-up and the loader's default path is patched here, in the test tree, the
way `monkeypatch` would do it in-process.  Active only when MP_TEST_CHECKER_SO names the checker."""
import os
import sys

_so = os.environ.get("MP_TEST_CHECKER_SO")
if _so:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from multiprime_amd import _abi

    _product_init = _abi.Library.__init__

    def _checker_init(self, path=None):
        _product_init(self, path or _so)

    _abi.Library.__init__ = _checker_init
