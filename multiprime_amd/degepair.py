"""Drop-in for scripts/get_degePrimer.py (get_degePrimer_V6.py, "GD") — the pairing step of the DegePrime workflow
(multi-DegePrime.py rule get_degePrimer), twin of get_multiPrime.py (multiprime_amd/pairing.py): same class
(`Primers_filter`), flags and output line, fed by a DEGEPRIME table (Pos / ... / PrimerSeq / NumberMatching) instead of
this build's core TSV.

Everything here is O(primers) host work — there is no per-sequence loop in this script, so nothing goes to the GPU; the
per-primer filters (hairpin, GC, repeats, 3' degeneracy, GC clamp) are memoised, which is where the reference spends its
time.  Behaviours of the reference that are kept because the output depends on them:
  * `current_end` builds its set with `end_seq.union(...)` and drops the result (GD:319-325): the set is always empty, so
    `dimer_check` (GD:349-375) never rejects a pair — the F-R dimer test of this script is dead code;
  * `main` does not pass -g to the class, so the GC window is the class default 0.4-0.6 whatever the flag says (GD:541-546);
  * `degenerate_seq` is a generator: in `hairpin_check` only the FIRST expansion of the 5-mer stem meets the tail
    expansions (GD:297-316), as in get_multiPrime;
  * pairs are sorted by min(NumberMatching of the two windows), descending, stable (GD:524).
"""
from __future__ import annotations

import os
import sys
import time
from bisect import bisect_left
from optparse import OptionParser

from . import host, iupac
from .pairing import Primers_filter as _GM


class Primers_filter(object):
    def __init__(self, ref_file, primer_file, adaptor, rep_seq_number=500, distance=4, outfile="", size="300,700", position=9,
                 GC="0.4,0.6", nproc=10, fraction=0.6):
        self.nproc = nproc
        self.primer_file = primer_file
        self.adaptor = adaptor
        self.size = size
        self.outfile = os.path.abspath(outfile)
        self.distance = distance
        self.Input_file = ref_file
        self.fraction = fraction
        self.GC = GC
        self.rep_seq_number = rep_seq_number
        self.number = self.get_number()
        self.position = position
        self.primers = self.parse_primers()
        self._memo = {}
        self.pre_filter_primers = self.pre_filter()

    def parse_primers(self):
        """GD:245-257: position -> [primer, fraction of sequences matching, number matching]."""
        primer_dict = {}
        with open(self.primer_file) as f:
            for i in f:
                if i.startswith("Pos"):
                    continue
                i = i.strip().split("\t")
                primer_dict[int(i[0])] = [i[5], round(int(i[6]) / self.number, 2), int(i[6])]
        return primer_dict

    def get_number(self):
        """GD:260-271: number of records of the reference FASTA = newlines / 2, capped at -m."""
        seq_number = int(host.count_newlines(self.Input_file) / 2)
        if seq_number > self.rep_seq_number != 0:
            print(seq_number, self.rep_seq_number)
            return self.rep_seq_number
        return seq_number

    # the per-primer filters are those of get_multiPrime (same code in both reference scripts), memoised per string
    def _cached(self, name, fn, key):
        k = (name, key)
        if k not in self._memo:
            self._memo[k] = fn(self, key)
        return self._memo[k]

    def hairpin_check(self, primer):
        return self._cached("hairpin", _GM.hairpin_check, primer)

    def GC_fraction(self, sequence):
        return self._cached("gc", lambda s, x: _GM.GC_fraction(x), sequence)

    def di_nucleotide(self, primer):
        return self._cached("rep", lambda s, x: _GM.di_nucleotide(x), primer)

    def dege_filter_in_term_N_bp(self, sequence):
        return self._cached("term", _GM.dege_filter_in_term_N_bp, sequence)

    def GC_clamp(self, primer, num=4, length=13):
        return self._cached("clamp", lambda s, x: any(s.GC_fraction(x[-i:]) > 0.6 for i in range(num, num + length)), primer)

    @staticmethod
    def dimer_check(primer_F, primer_R):
        """GD:349-375 iterates over current_end(F) | current_end(R), which is always empty (see the module docstring)."""
        return False

    def pre_filter(self):
        """GD:419-438."""
        lo, hi = (float(x) for x in self.GC.split(","))
        keep = []
        for pos, (primer, coverage, _) in self.primers.items():
            if self.hairpin_check(primer):
                continue
            gc = self.GC_fraction(primer)
            if gc > hi or gc < lo:
                continue
            if self.di_nucleotide(primer):
                continue
            if coverage < self.fraction:
                continue
            keep.append(pos)
        return sorted(keep)

    @staticmethod
    def closest(positions, lo, hi):
        """Index range [first, last] of the sorted `positions` inside [lo, hi) — `hi` past the last position keeps the last index
        (GD:441-448: first > last when the range holds none)."""
        last = len(positions) - 1 if hi > positions[-1] else bisect_left(positions, hi) - 1
        return bisect_left(positions, lo), last

    def primer_pairs(self, primer_pairs):
        """GD:450-500."""
        min_len, max_len = (int(x) for x in self.size.split(","))
        cand = self.pre_filter_primers
        adaptor = self.adaptor.split(",")
        if int(cand[-1]) - int(cand[0]) < min_len:
            return
        for start in range(len(cand)):
            fwd = self.primers[cand[start]][0]
            if self.hairpin_check(adaptor[0] + fwd) or self.dege_filter_in_term_N_bp(fwd) or self.GC_clamp(fwd):
                continue
            start_index, stop_index = self.closest(cand, cand[start] + min_len, cand[start] + max_len)
            if start_index > stop_index:
                break
            for stop in range(start_index, stop_index + 1):
                rev = iupac.revcomp(self.primers[cand[stop]][0])
                if self.hairpin_check(adaptor[1] + rev) or self.dege_filter_in_term_N_bp(rev) or self.GC_clamp(rev):
                    continue
                distance = int(cand[stop]) - int(cand[start]) + 1
                if distance > max_len:
                    break
                if min_len <= distance <= max_len and not self.dimer_check(fwd, rev):
                    primer_pairs.append((fwd, rev, distance, min(self.primers[cand[start]][2], self.primers[cand[stop]][2]),
                                         str(cand[start]) + ":" + str(cand[stop])))

    def run(self):
        primer_pairs = []
        self.primer_pairs(primer_pairs)
        primer_pairs_sort = sorted(primer_pairs, key=lambda k: k[3], reverse=True)
        with open(self.outfile, "w") as fo:
            fo.write(str(self.outfile) + "\t")
            for i in primer_pairs_sort:
                fo.write("\t".join(map(str, i)) + "\t")
            fo.write("\n")


def parse_args(argv=None):
    parser = OptionParser('Usage: %prog -i [input] -r [sequence.fa] -o [output] \n \
                Options: {-f [0.6] -m [500] -n [200] -e [4] -p [9] -s [250,500] -g [0.4,0.6] -d [4] -a ","}.')
    parser.add_option('-i', '--input', dest='input', help='Input file: degeprimer out.')
    parser.add_option('-r', '--ref', dest='ref', help='Reference sequence file: all the sequence in 1 fasta, for example: (Cluster_96_171.fa).')
    parser.add_option('-g', '--gc', dest='gc', default="0.4,0.6", help="Filter primers by GC content. Default [0.4,0.6].")
    parser.add_option('-f', '--fraction', dest='fraction', default="0.6", type="float", help="Filter primers by match fraction. Default: 0.6.")
    parser.add_option('-e', '--end', dest='end', default="4", type="int", help="Filter primers by degenerate base position. Default: 4.")
    parser.add_option('-p', '--proc', dest='proc', default="10", type="int", help="Number of process to launch.  default: 10.")
    parser.add_option('-s', '--size', dest='size', default="250,500", help="Filter primers by PRODUCT size. Default [250,500].")
    parser.add_option('-d', '--dist', dest='dist', default=4, type="int", help='Filter param of hairpin: distance of the minimal paired bases. Default: 4.')
    parser.add_option('-a', '--adaptor', dest='adaptor', default="TCTTTCCCTACACGACGCTCTTCCGATCT,TCTTTCCCTACACGACGCTCTTCCGATCT", type="str",
                      help='Adaptor sequence, which is used for NGS next. If you dont want adaptor, use [","]')
    parser.add_option('-m', '--maxseq', dest='maxseq', default=500, type="int", help='Limit of sequence number. Default: 500.')
    parser.add_option('-o', '--out', dest='out', help='Output file: candidate primers. e.g. [*].candidate.primers.txt.')
    options, args = parser.parse_args(argv)
    if (argv is None and len(sys.argv) == 1) or (argv is not None and not argv):
        parser.print_help()
        sys.exit(1)
    for value, msg in ((options.input, "Input file must be specified !!!"), (options.ref, "Reference file must be specified !!!"),
                       (options.out, "No output file provided !!!")):
        if value is None:
            parser.print_help()
            print(msg)
            sys.exit(1)
    return options, args


def main(argv=None):
    from ._abi import prefer_staged_copies
    prefer_staged_copies()                      # a command line owns its process: see _abi.prefer_staged_copies
    e1 = time.time()
    options, _ = parse_args(argv)
    Primers_filter(ref_file=options.ref, primer_file=options.input, adaptor=options.adaptor, rep_seq_number=options.maxseq,
                   distance=options.dist, outfile=options.out, size=options.size, position=options.end, fraction=options.fraction,
                   nproc=options.proc).run()
    e2 = time.time()
    print("INFO {} Total times: {}".format(time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(time.time())), round(float(e2 - e1), 2)))
