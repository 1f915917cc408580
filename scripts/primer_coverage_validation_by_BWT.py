#!/usr/bin/env python3
"""Drop-in entry point with the reference script's name and flags (scripts/primer_coverage_validation_by_BWT.py):
the bowtie2 + samtools mapping step is replaced by one GPU k-mismatch scan — see multiprime_amd/validate.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiprime_amd.validate import main  # noqa: E402

if __name__ == "__main__":
    main()
