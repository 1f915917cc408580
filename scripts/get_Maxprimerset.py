#!/usr/bin/env python3
"""Drop-in for the reference's scripts/get_Maxprimerset.py (same flags -i -a -s -m -o, same output
files incl. sort.<input> and <out>.next.xls); the dimer examinations run on an MI355X."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiprime_amd.maxset import main  # noqa: E402

if __name__ == "__main__":
    main()
