#!/usr/bin/env python3
"""Drop-in for the reference's scripts/multiPrime-core.py: same file name, same flags, same
output files; the work runs on an MI355X through libmprime_hip.so."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiprime_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()
