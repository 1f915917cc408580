#!/usr/bin/env python3
"""Drop-in entry point with the reference script's name and flags (scripts/get_degePrimer.py) — see multiprime_amd/degepair.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiprime_amd.degepair import main  # noqa: E402

if __name__ == "__main__":
    main()
