"""Primer pairing — drop-in for scripts/get_multiPrime.py (get_multiPrime_V8.py, "GM"), the consumer of
the core step's three files.

The reference walks every (forward window, reverse window) combination inside the product-size
range and, for each, re-runs per-window string filters, a 3'-end dimer search and a union of
sequence-id lists.  Here
  * the per-window filters (hairpin with adaptor, degenerate 3' end, GC clamp; GM:509-534) are
    evaluated once per window instead of once per combination;
  * the dimer searches of all combinations run in one `mp_dimer_pairs` launch (GM:419-438 is the
    union of the ordered pairs F->F, F->R, R->F, R->R; the two self terms are per window);
  * the id-list unions (GM:560-569) become popcounts of OR-ed per-window sequence bitsets in one
    `mp_pair_coverage` launch — the bitset form also removes the O(windows x sequences) Python sets.
The output files, their order, the retry pass that re-appends pairs (GM:629-637), the stdout lines and
the file-name quirks (`str.strip(".txt")` strips characters) are the reference's.
"""
from __future__ import annotations

import json
import os
import sys
import time
from bisect import bisect_left

import numpy as np

from . import batchfilters, host, iupac, thermo
from ._abi import Library
from .dimer import MAX_LEN, dg_limit, encode_primers

HEADERS = ["Primer_F_seq", "Primer_R_seq", "Product length:Tm:coverage_percentage", "Target number", "Primer_start_end"]


def _loss_table_strict(threshold: float) -> np.ndarray:
    """loss_hit[l][GC][d2] = Penalty_points(l, GC, 0, d2) > threshold  (GM:431-435: strictly greater)."""
    by_sum = np.array([[thermo.penalty_points(sm, 0, 0, d2) > threshold for d2 in range(64)] for sm in range(2 * MAX_LEN + 1)],
                      np.uint8)               # 2**l * 2**GC == 2**(l+GC): only the sum matters
    t = np.zeros((MAX_LEN + 1, MAX_LEN + 1, 64), np.uint8)
    for l in range(1, MAX_LEN + 1):
        for gc in range(0, l + 1):
            t[l, gc] = by_sum[l + gc]
    return t


def _dg_params_one_end() -> np.ndarray:
    """deltaG of GM:398-417: like finDimer's but only the FIRST base's initiation term is added."""
    p = np.zeros(16 + 32 + MAX_LEN + 1 + 1, np.float64)
    for i in range(4):
        for j in range(4):
            p[i * 4 + j] = thermo._DG[i][j]
    for a, ca in enumerate("ACGT"):
        for b in range(4):
            p[16 + (a * 4 + b) * 2 + 0] = thermo._DG_END[ca]
            p[16 + (a * 4 + b) * 2 + 1] = thermo._DG_END[ca] + thermo._DG_TA
    for n in range(MAX_LEN + 1):
        p[48 + n] = thermo._NA_TERM * n
    p[48 + MAX_LEN + 1] = thermo._DG_SYMMETRY
    return p


class Primers_filter(object):
    """Drop-in for the reference class of the same name (GM:303-662)."""

    def __init__(self, ref_file, primer_file, adaptor, rep_seq_number=500, distance=4, outfile="", diff_Tm=5,
                 size="300,700", position=9, GC="0.4,0.6", nproc=10, fraction=0.6, *, library: Library | None = None,
                 device: int = 0, core=None):
        # core: the NN_degenerate object of the core step run in THIS process with keep_bitsets=True — its per-window
        # coverage bitsets are still on the device and the coverage unions are taken there (no JSON, no file, no copy)
        self.core = core
        self.nproc = nproc
        self.primer_file = primer_file
        self.adaptor = adaptor
        self.size = size
        self.outfile = os.path.abspath(outfile)
        self.distance = distance
        self.Input_file = ref_file
        self.fraction = fraction
        self.GC = GC
        self.diff_Tm = diff_Tm
        self.rep_seq_number = rep_seq_number
        self.number = self.get_number()
        self.position = position
        self.primers, self.gap_id, self.non_cover_id = self.parse_primers()
        if core is not None:
            self.lib, self.ctx = core.lib, core.ctx
        else:
            self.lib = library if library is not None else Library()
            self.ctx = self.lib.context(device)
        self.pre_filter_primers = self.pre_filter()
        self.stats = {}

    # ---- input ---------------------------------------------------------------------------------
    def get_number(self):
        """GM:348-357: number of sequences = newlines / 2, capped by --maxseq when that is not 0."""
        newlines = host.count_newlines(self.Input_file)            # the count of the reference's whole-file text read, on several threads
        seq_number = int(newlines / 2)
        if seq_number > self.rep_seq_number != 0:
            return self.rep_seq_number
        return seq_number

    def parse_primers(self):
        """GM:323-345."""
        primer_dict = {}
        with open(self.primer_file) as f:
            for line in f:
                if line.startswith("Pos"):
                    continue
                i = line.strip().split("\t")
                primer_dict[int(i[0])] = [i[3], round(int(i[6]) / self.number, 2), int(i[7]), int(i[8]), round(float(i[9]), 2)]
        # the core step's coverage side data: the two JSON files of the reference format, or — when they were not
        # written (deep alignments) — the bitset file of `multiPrime-core.py --bitsets`
        self.bitset_file = None
        gap_json, non_json = self.primer_file + ".gap_seq_id_json", self.primer_file + ".non_coverage_seq_id_json"
        if self.core is not None and (self.core.mask_index or self.core.keep_bitsets):      # (no primer at all: an empty index)
            gap_dict, non_cover_dict = None, None
        elif os.path.exists(gap_json) and os.path.exists(non_json):
            with open(gap_json) as g:
                gap_dict = json.load(g)
            with open(non_json) as n:
                non_cover_dict = json.load(n)
        elif os.path.exists(self.primer_file + ".coverage_bitsets.npz"):
            self.bitset_file = self.primer_file + ".coverage_bitsets.npz"
            gap_dict, non_cover_dict = None, None
        else:
            raise FileNotFoundError(gap_json)           # what the reference raises
        return primer_dict, gap_dict, non_cover_dict

    # ---- per-window string filters ---------------------------------------------------------------
    def hairpin_check(self, primer):
        """GM:373-384.  `degenerate_seq` is a generator in this script, so the tail expansions are
        consumed while the FIRST expansion of the 5-mer is tested and the other 5-mer expansions
        meet an exhausted iterator: only the first stem expansion is ever compared."""
        for n in range(0, len(primer) - 5 - 5 - self.distance + 1):
            stem = iupac.revcomp(iupac.expand(primer[n:n + 5])[0])
            if iupac.occurs_in_some_expansion(stem, primer[n + 5 + self.distance:]):      # == any(stem in t for t in expand(tail))
                return True
        return False

    def _hairpins(self, primers):
        """hairpin_check of every primer: all at once on the symbol-code matrix when they have one length (they do: adaptor + k-mer)."""
        if primers and len({len(p) for p in primers}) == 1:
            codes = iupac.MASK_LUT[np.frombuffer("".join(primers).encode(), np.uint8)].reshape(len(primers), len(primers[0]))
            if (codes != 0).all():
                return batchfilters.hairpin_first_stem_of_primers(codes, self.distance).tolist()
        return [self.hairpin_check(p) for p in primers]

    @staticmethod
    def GC_fraction(sequence):
        """GM:451-458: mean (not rounded) of the per-expansion GC fractions rounded to 3 decimals."""
        n = len(sequence)
        return iupac.exact_mean([round((s.count("G") + s.count("C")) / n, 3) for s in iupac.expand(sequence)])

    @staticmethod
    def di_nucleotide(primer):
        return any(iupac.REPEATS.search(s) for s in iupac.expand(primer))

    def dege_filter_in_term_N_bp(self, sequence):
        """GM:441-449: a degenerate symbol among the last `position` bases."""
        if self.position == 0:
            return False
        return iupac.degeneracy(sequence[-self.position:]) > 1

    def GC_clamp(self, primer, num=4, length=13):
        """GM:469-475."""
        for i in range(num, num + length):
            if self.GC_fraction(primer[-i:]) > 0.6:
                return True
        return False

    def pre_filter(self):
        """GM:477-497."""
        lo, hi = (float(x) for x in self.GC.split(","))
        keep = []
        hairpin = dict(zip(self.primers, self._hairpins([info[0] for info in self.primers.values()])))
        for pos, info in self.primers.items():
            primer = info[0]
            if hairpin[pos]:
                continue
            gc = self.GC_fraction(primer)
            if gc > hi or gc < lo:
                continue
            if self.di_nucleotide(primer):
                continue
            keep.append(pos)
        return sorted(keep)

    @staticmethod
    def closest(positions, lo, hi):
        """Index range [first, last] of the sorted `positions` inside [lo, hi) — `hi` past the last position keeps the last index
        (GM:499-507: first > last when the range holds none)."""
        last = len(positions) - 1 if hi > positions[-1] else bisect_left(positions, hi) - 1
        return bisect_left(positions, lo), last

    # ---- sequence bitsets --------------------------------------------------------------------------
    def _bitsets(self, cand):
        """Per candidate window: the sequences a forward / reverse primer there does NOT reach —
        gap rows U F (resp. R) non-covered ids (GM:560-567) — as bitsets over the ids seen."""
        if self.bitset_file is not None:
            z = np.load(self.bitset_file)
            where = {int(p): i for i, p in enumerate(z["positions"].tolist())}
            sel = np.asarray([where[int(p)] for p in cand], np.int64)
            return [np.ascontiguousarray(z["not_f"][sel]), np.ascontiguousarray(z["not_r"][sel])]
        index = {}
        rows_f, rows_r = [], []
        for pos in cand:
            key = str(pos)
            gap = [i for ids in self.gap_id[key].values() for i in ids]
            f = gap + [i for ids in self.non_cover_id[key][0].values() for i in ids]
            r = gap + [i for ids in self.non_cover_id[key][1].values() for i in ids]
            for lst, dst in ((f, rows_f), (r, rows_r)):
                dst.append(np.fromiter((index.setdefault(i, len(index)) for i in lst), dtype=np.int64, count=len(lst)))
        n_words = max(1, (len(index) + 63) // 64)
        out = []
        for rows in (rows_f, rows_r):
            m = np.zeros((len(cand), n_words), np.uint64)
            for w, ids in enumerate(rows):
                if len(ids):
                    np.bitwise_or.at(m[w], ids >> 6, np.uint64(1) << (ids & 63).astype(np.uint64))
            out.append(m)
        return out

    # ---- driver ------------------------------------------------------------------------------------
    def run(self):
        t_run = time.time()
        min_len, max_len = (int(x) for x in self.size.split(","))
        cand = self.pre_filter_primers
        adaptor = self.adaptor.split(",")
        threshold = 1 - self.fraction
        print("Candidata degenerate primer number is: {}".format(len(cand)))
        ID = str(self.outfile)
        if not cand:
            # no primer survived the core step's filters: the reference dies here with an IndexError (candidate_primer[-1] of an
            # empty list, GM:611; tests/golden/chain_k36.json.gz records its exit status 1 and that it writes no file).  Same status,
            # a message instead of a traceback.
            print("Error: {} holds no candidate primer; the reference fails on such an input too (IndexError).".format(self.primer_file),
                  file=sys.stderr)
            sys.exit(1)
        if int(cand[-1]) - int(cand[0]) < min_len:                                   # GM:611-618
            print("Max PCR product legnth < min len!")
            with open(self.outfile, "w") as fo:
                fo.write(ID + "\n")
            return
        fwd = [self.primers[p][0] for p in cand]
        rev = [iupac.revcomp(s) for s in fwd]
        hp_f, hp_r = self._hairpins([adaptor[0] + s for s in fwd]), self._hairpins([adaptor[1] + s for s in rev])
        f_ok = [not (h or self.dege_filter_in_term_N_bp(s) or self.GC_clamp(s)) for h, s in zip(hp_f, fwd)]
        r_ok = [not (h or self.dege_filter_in_term_N_bp(s) or self.GC_clamp(s)) for h, s in zip(hp_r, rev)]
        # every (start, stop) combination the reference reaches its dimer check with, in its order
        combos = []                       # (start index, stop index, distance)
        first_of = [0] * (len(cand) + 1)
        for a in range(len(cand)):
            first_of[a] = len(combos)
            if not f_ok[a]:
                continue
            lo, hi = self.closest(cand, cand[a] + min_len, cand[a] + max_len)
            for b in range(lo, hi + 1):
                if not r_ok[b]:
                    continue
                dist = int(cand[b]) - int(cand[a]) + 1
                if dist > max_len:
                    combos.append((a, b, -1))                                            # "Error!" + break, GM:537-539
                    break
                if min_len <= dist <= max_len:
                    combos.append((a, b, dist))
        first_of[len(cand)] = len(combos)
        # dimer flags on the device: primer 2i = forward of window i, 2i+1 = reverse complement of window i
        t0 = time.time()
        codes, off = encode_primers([s for pair in zip(fwd, rev) for s in pair])
        real = [(a, b) for a, b, d in combos if d > 0]
        used_f = sorted({a for a, _ in real})
        used_r = sorted({b for _, b in real})
        plist = ([(2 * a, 2 * a) for a in used_f] + [(2 * b + 1, 2 * b + 1) for b in used_r]
                 + [(2 * a, 2 * b + 1) for a, b in real] + [(2 * b + 1, 2 * a) for a, b in real])
        flags = self.ctx.dimer_pairs(codes, off, np.asarray(plist, np.int32).reshape(-1, 2), _loss_table_strict(3.6),
                                     _dg_params_one_end(), dg_limit()) if plist else np.zeros(0, np.uint8)
        self_f = dict(zip(used_f, flags[: len(used_f)].tolist()))
        self_r = dict(zip(used_r, flags[len(used_f): len(used_f) + len(used_r)].tolist()))
        o = len(used_f) + len(used_r)
        cross = {ab: bool(flags[o + i] or flags[o + len(real) + i]) for i, ab in enumerate(real)}
        dimer = {ab: bool(self_f[ab[0]] or self_r[ab[1]] or cross[ab]) for ab in real}
        self.stats["dimer_s"] = time.time() - t0
        # coverage of every combination that survives the dimer and Tm tests, on the device
        t0 = time.time()
        tm = [self.primers[p][4] for p in cand]
        alive = [ab for ab in real if not dimer[ab] and not abs(tm[ab[0]] - tm[ab[1]]) > self.diff_Tm]
        if not alive:
            counts = []
        elif self.core is not None and self.core.mask_index:
            idx = np.asarray([self.core.mask_index[int(p)] for p in cand], np.int32)          # candidate window -> resident mask
            counts = self.ctx.pair_coverage_resident(idx[np.asarray(alive, np.int32).reshape(-1, 2)])
        else:
            sets_f, sets_r = self._bitsets(cand)
            counts = self.ctx.pair_coverage(sets_f, sets_r, np.asarray(alive, np.int32).reshape(-1, 2))
        non_cover = dict(zip(alive, (int(x) for x in counts)))
        self.stats["coverage_s"] = time.time() - t0
        self.stats["n_combinations"] = len(real)

        primer_pairs = []

        def one_pass(thr, announce):
            for a in range(len(cand)):
                if announce:
                    print(a)                                                            # GM:620
                for a_, b, dist in combos[first_of[a]:first_of[a + 1]]:
                    if dist < 0:
                        print("Error! PCR product greater than max length !")
                        break
                    if dimer[(a, b)]:
                        print("Dimer detection between Primer-F and Primer-R!")
                        continue
                    if (a, b) not in non_cover:                                         # Tm difference too large
                        continue
                    n_non = non_cover[(a, b)]
                    if n_non / self.number > thr:
                        continue
                    all_coverage = self.number - n_non
                    cover_percentage = round(all_coverage / self.number, 4)
                    average_tm = str(round(iupac.exact_mean([tm[a], tm[b]]), 2))      # statistics.mean of two floats
                    primer_pairs.append((fwd[a], rev[b], str(dist) + ":" + average_tm + ":" + str(cover_percentage),
                                         all_coverage, str(cand[a]) + ":" + str(cand[b])))

        one_pass(threshold, True)
        if len(primer_pairs) < 10:                                                      # GM:629-637 (appends again)
            one_pass(threshold + 0.1, False)
        primer_id = str(self.outfile).split("/")[-1].rstrip(".txt")
        stem = self.outfile.strip(".txt")
        with open(self.outfile, "w") as fo, open(stem + ".xls", "w") as fo_xls, open(stem + ".fa", "w") as fa:
            fo_xls.write("\t".join(HEADERS) + "\n")
            fo.write(ID + "\t")
            for i in sorted(primer_pairs, key=lambda k: k[3], reverse=True):
                fo.write("\t".join(map(str, i)) + "\t")
                fo_xls.write("\t".join(map(str, i)) + "\n")
                start_stop = i[4].split(":")
                fa.write(">" + primer_id + "_" + start_stop[0] + "F\n" + i[0] + "\n>" + primer_id + "_" + start_stop[1]
                         + "R\n" + i[1] + "\n")
            fo.write("\n")
        self.stats["run_s"] = time.time() - t_run


def parse_args(argv=None):
    import argparse
    p = argparse.ArgumentParser(description="For degenerate primer design")
    p.add_argument("-i", "--input", type=str, required=True, metavar="<file>", help="Input file: multiPrime out.")
    p.add_argument("-r", "--ref", type=str, required=True, metavar="<str>",
                   help="Reference sequence file: all the sequence in 1 fasta, for example: (Cluster_96_171.tfa).")
    p.add_argument("-g", "--gc", type=str, default="0.2,0.7", metavar="<str>",
                   help="Accepted for compatibility: the reference parses it but never passes it on (GC limits stay 0.4,0.6).")
    p.add_argument("-f", "--fraction", type=float, default=0.6, metavar="<float>", help="Filter primers by match fraction. Default: 0.6.")
    p.add_argument("-e", "--end", type=int, default=4, metavar="<int>",
                   help="No degenerate base among the last N bases. Default: 4.")
    p.add_argument("-p", "--proc", type=int, default=20, metavar="<int>", help="Accepted for compatibility.")
    p.add_argument("-s", "--size", type=str, default="250,500", metavar="<str>", help="Filter primers by PRODUCT size. Default [250,500].")
    p.add_argument("-d", "--dist", type=int, default=4, metavar="<int>", help="Hairpin: distance of the minimal paired bases. Default: 4.")
    p.add_argument("-t", "--Tm", type=int, default=4, metavar="<int>", help="Difference of Tm between primer-F and primer-R. Default: 4.")
    p.add_argument("-a", "--adaptor", type=str, default="TCTTTCCCTACACGACGCTCTTCCGATCT,TCTTTCCCTACACGACGCTCTTCCGATCT",
                   metavar="<str>", help='Adaptor sequences F,R for hairpin detection. If you dont want adaptor, use [","]')
    p.add_argument("-m", "--maxseq", type=int, default=0, metavar="<int>", help="Limit of sequence number. Default: 0 (all).")
    p.add_argument("-o", "--out", type=str, required=True, metavar="<file>", help="Output file: candidate primers.")
    p.add_argument("--device", type=int, default=0, help="GPU ordinal")
    return p.parse_args(argv)


def main(argv=None):
    from ._abi import prefer_staged_copies
    prefer_staged_copies()                      # a command line owns its process: see _abi.prefer_staged_copies
    args = parse_args(argv)
    e1 = time.time()
    Primers_filter(ref_file=args.ref, primer_file=args.input, adaptor=args.adaptor, rep_seq_number=args.maxseq,
                   distance=args.dist, outfile=args.out, size=args.size, position=args.end, fraction=args.fraction,
                   diff_Tm=args.Tm, nproc=args.proc, device=args.device).run()
    e2 = time.time()
    print("INFO {} Total times: {}".format(time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(time.time())),
                                           round(float(e2 - e1), 2)))


if __name__ == "__main__":
    main()
