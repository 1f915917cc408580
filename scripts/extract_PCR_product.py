#!/usr/bin/env python3
"""Drop-in for the reference's scripts/extract_PCR_product.py (same flags -r -i -f -o -p -s, same files);
the exact search of every primer pair in every sequence runs on an MI355X."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiprime_amd.pcr import main  # noqa: E402

if __name__ == "__main__":
    main()
