"""Command line of the core step — the exact flag set of scripts/multiPrime-core.py
(parseArg, V20:60-102), so Snakemake rule `multiPrime` (multiPrime.py:200-207) can call this
program instead of the reference script without touching the rule:

    python {scripts_dir}/multiPrime-core.py -i {tmsa} -n 4 -d 10 -v 1 -c 2,3,-1 -g 0.2,0.7 -s 150 \\
           -l 18 -e 3.6 -o {out} -f 0.7 -p 1

Extra, optional flags: --device (GPU ordinal), --no-json (skip the two O(windows x sequences)
JSON side files, which stop being writable at ~10^5 sequences — SURVEY §7 "hard parts").
"""
from __future__ import annotations

import argparse
import time


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="For degenerate primer design (MI355X-native core step)")
    p.add_argument("-i", "--input", type=str, required=True, metavar="<file>",
                   help="Input file: multi-alignment output (muscle or others).")
    p.add_argument("-l", "--plen", type=int, default=18, metavar="<int>", help="Length of primer. Default: 18.")
    p.add_argument("-n", "--dnum", type=int, default=4, metavar="<int>", help="Max number of degenerate. Default: 4.")
    p.add_argument("-d", "--degeneracy", type=int, default=10, metavar="<int>", help="Max degeneracy of primer. Default: 10.")
    p.add_argument("-v", "--variation", type=int, default=1, metavar="<int>", help="Max mismatch number of primer. Default: 1")
    p.add_argument("-e", "--entropy", type=float, default=3.6, metavar="<float>",
                   help="Entropy threshold of a primer-length window. Default: 3.6.")
    p.add_argument("-g", "--gc", type=str, default="0.2,0.7", metavar="<str>", help="Filter primers by GC content. Default [0.2,0.7].")
    p.add_argument("-s", "--size", type=int, default=100, metavar="<int>", help="Filter primers by mini PRODUCT size. Default 100.")
    p.add_argument("-f", "--fraction", type=float, default=0.8, metavar="<float>", help="Filter primers by match fraction. Default: 0.8.")
    p.add_argument("-c", "--coordinate", type=str, default="1,2,-1", metavar="<str>",
                   help="Positions where a mismatch is not tolerated (>0: from the 5' end, <0: from the 3' end). Default: 1,2,-1.")
    p.add_argument("-p", "--proc", type=int, default=20, metavar="<int>",
                   help="Accepted for compatibility (the reference's process pool is inert; the work runs on the GPU).")
    p.add_argument("-a", "--away", type=int, default=4, metavar="<int>", help="Hairpin: minimal distance of paired bases. Default: 4.")
    p.add_argument("-o", "--out", type=str, required=True, metavar="<file>", help="output file")
    p.add_argument("--device", type=int, default=0, help="GPU ordinal (default 0)")
    p.add_argument("--no-json", action="store_true", help="do not write the two *_seq_id_json side files")
    p.add_argument("--bitsets", action="store_true",
                   help="also write <out>.coverage_bitsets.npz: per window, one bit per sequence the primer does not "
                        "reach (the bitset form of the JSON files; scripts/get_multiPrime.py reads it when the JSON is absent)")
    p.add_argument("--stats", action="store_true", help="print per-phase timings to stderr")
    return p.parse_args(argv)


def main(argv=None):
    from .core import NN_degenerate
    args = parse_args(argv)
    e1 = time.time()
    app = NN_degenerate(seq_file=args.input, primer_length=args.plen, coverage=args.fraction,
                        number_of_dege_bases=args.dnum, score_of_dege_bases=args.degeneracy,
                        raw_entropy_threshold=args.entropy, product_len=args.size, position=args.coordinate,
                        variation=args.variation, distance=args.away, GC=args.gc, nproc=args.proc, outfile=args.out,
                        device=args.device, write_json=not args.no_json, write_bitsets=args.bitsets)
    app.run()
    e2 = time.time()
    if args.stats:
        import json
        import sys
        print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in app.stats.items()}), file=sys.stderr)
    # same closing line as the reference (V20:1193-1198): it lands in the Snakemake rule's log
    print("INFO {} Total times: {}".format(time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(time.time())),
                                           round(float(e2 - e1), 2)))


if __name__ == "__main__":
    main()
