#!/usr/bin/env python3
"""Drop-in for the reference's scripts/finDimer.py (same flags -i -n -t -o, same two output
files); the all-pairs 3'-end search runs on an MI355X through mp_dimer_scan."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="For primer dimer check")
    parser.add_argument("-i", "--input", type=str, required=True, help="input fasta primer file", metavar="<file>")
    parser.add_argument("-n", "--num", type=int, default=5, metavar="<int>",
                        help="accepted for compatibility (the reference's process count); the scan runs on the GPU")
    parser.add_argument("-t", "--threshold", type=float, default=3.96, metavar="<int>",
                        help="threshold of loss function. Default: 3.96")
    parser.add_argument("-o", "--output", type=str, required=True, help="output file", metavar="<file>")
    parser.add_argument("--device", type=int, default=0, help="GPU ordinal")
    return parser.parse_args(argv)


def main(argv=None):
    from multiprime_amd.dimer import Dimer
    args = parse_args(argv)
    e1 = time.time()
    Dimer(primer_file=args.input, threshold=args.threshold, outfile=args.output, nproc=args.num, device=args.device).run()
    e2 = time.time()
    print("INFO {} Total times: {}".format(time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(time.time())),
                                           round(float(e2 - e1), 2)))


if __name__ == "__main__":
    main()
