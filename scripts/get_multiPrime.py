#!/usr/bin/env python3
"""Drop-in for the reference's scripts/get_multiPrime.py (same flags, same three output files); the
dimer searches and coverage unions of all primer-pair combinations run on an MI355X."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiprime_amd.pairing import main  # noqa: E402

if __name__ == "__main__":
    main()
