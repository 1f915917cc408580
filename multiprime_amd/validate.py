"""Drop-in for scripts/primer_coverage_validation_by_BWT_V9.py ("BWT") — SURVEY §8f-3: which primer pairs of a primer
set amplify which sequences of a database when a few mismatches are tolerated, with the bowtie2 + samtools mapping step
(BWT:264-300) replaced by one exhaustive GPU scan (mp_kmm_scan, csrc/scan.hip).

Same class (`off_targets`), flags and output files as the reference script:
  <out>                   Chrom (or Genes) / Start / Stop / Primer_F / Primer_R / Product length
  <out>.pair.num          primer pairs by number of products and of distinct targets
  <out>.total.acc.num     number of covered sequences (and of all targets with -d)
  <out>.unmatched.fa      with -d <pickle>: the records no pair reaches
  <primers>.term.fa       the expanded 3' terms (BWT:205-239) — written as the reference does

PARITY UNPINNED.  bowtie2 and samtools are not installed in the authoring image, so no reference output could be
recorded.  What is restated exactly is everything the script itself does around the mapper (get_term, the MD:Z filter of
build_dict, PCR_product, the writers); the mapper is replaced by its acceptance rule under default end-to-end scoring:
an ungapped alignment with at most floor((0.6 + 0.6 L) / 6) mismatches (bowtie2's minimum score -0.6 - 0.6 L at 6 per
mismatch; `--max-mismatch` overrides), reported on both strands (`-a`).  Differences to expect against a real bowtie2 run:
gapped alignments are not reported; bowtie2's seed heuristics (-N, -L 8) can MISS alignments this scan finds; ties in
`dict(F_dict[gene])` (several primers at one start) resolve by pattern order here, by SAM order there; genes are written in
database order, the reference's order follows a set().
"""
from __future__ import annotations

import os
import pickle
import re
import time
from bisect import bisect_left
from collections import defaultdict
from pathlib import Path

import numpy as np

from . import iupac
from ._abi import Library

_DEGENERATE = set("RYMKSWHBVDN")


def degenerate_seq(primer: str):
    """BWT:193-203: expansions in itertools.product order; symbols outside the IUPAC table stay as they are."""
    parts = [iupac.MEMBERS[ch] if ch in _DEGENERATE else ch for ch in primer]
    out = [""]
    for p in parts:
        out = [a + b for a in out for b in p]
    return out


def closest(my_list, my_number1, my_number2):
    """BWT:160-168."""
    index_left = bisect_left(my_list, my_number1)
    if my_number2 > my_list[-1]:
        index_right = len(my_list) - 1
    else:
        index_right = bisect_left(my_list, my_number2) - 1
    return index_left, index_right


def bowtie2_mismatch_budget(length: int) -> int:
    """Mismatches bowtie2 --end-to-end admits with default scoring: min score -0.6 - 0.6 L, 6 per mismatch (high quality)."""
    return int((0.6 + 0.6 * length) // 6)


class off_targets(object):
    def __init__(self, primer_file, term_length, reference_file, PCR_product_size, mismatch_num, outfile, term_threshold,
                 bowtie="bowtie2", nproc=20, targets="None", *, library: Library | None = None, device: int = 0, max_mismatch=None):
        self.bowtie = bowtie                    # accepted for compatibility: no external mapper is run
        self.term_threshold = term_threshold
        self.nproc = nproc
        self.term_len = term_length
        self.primer_file = primer_file
        self.reference_file = reference_file
        self.outfile = outfile
        self.PCR_size = PCR_product_size
        self.mismatch_num = mismatch_num        # bowtie's -N / -n: seed sensitivity only, the scan is exhaustive
        self.targets = targets
        self.max_mismatch = max_mismatch
        self.lib = library if library is not None else Library()
        self.ctx = self.lib.context(device)
        self.stats = {}

    def get_term(self):
        """BWT:205-239."""
        Output = Path(self.primer_file).parent.joinpath(Path(self.primer_file).stem).with_suffix(".term.fa")
        term_len = self.term_len
        term_list = defaultdict(list)
        seq_ID = defaultdict(list)
        with open(self.primer_file, "r") as f:
            for i in f:
                if i.startswith(">"):
                    value = i.strip().lstrip(">")
                else:
                    key = i.strip() if term_len == 0 else i.strip()[-term_len:]
                    term_list[key].append(value)
        for k in term_list.keys():
            Id = "_".join(dict.fromkeys(term_list[k]))           # the reference joins a set(): first-seen order here
            expand_seq = degenerate_seq(k)
            if len(expand_seq) > 1:
                for j in range(len(expand_seq)):
                    seq_ID[expand_seq[j]].append(Id + "_" + str(j))
            else:
                seq_ID[k].append(Id + "_0")
        with open(Output, "w") as fo:
            for seq in seq_ID.keys():
                fo.write(">" + "_".join(seq_ID[seq]) + "\n" + seq + "\n")
        return seq_ID

    def _reference_records(self):
        from .host import Fasta
        path = str(self.reference_file)
        if not os.path.exists(path):
            raise FileNotFoundError(path + ": the scan needs the reference FASTA itself (a bowtie index prefix is not enough)")
        fa = Fasta(path)
        data, row_off = fa.rows()
        names = [s[1:] if s.startswith(">") else s for s in fa.ids]     # bowtie names a reference by its first token
        return names, data, row_off

    def scan(self, seq_ID):
        """Replaces bowtie_map + build_dict_run (BWT:241-316): {gene: [[start, primer], ...]} for both strands."""
        t0 = time.time()
        names, data, row_off = self._reference_records()
        reads = [(seq, "_".join(ids)) for seq, ids in seq_ID.items()]
        usable = [(i, seq) for i, (seq, _) in enumerate(reads) if seq and not set(seq.upper()) - set("ACGT") and 4 <= len(seq) <= 32]
        forward_dict, reverse_dict = defaultdict(list), defaultdict(list)
        by_budget = defaultdict(list)
        for i, seq in usable:
            budget = self.max_mismatch if self.max_mismatch is not None else bowtie2_mismatch_budget(len(seq))
            by_budget[budget].append((i, seq.upper()))
        all_hits = []
        for budget, group in by_budget.items():
            codes = iupac.MASK_LUT[np.frombuffer("".join(s for _, s in group).encode(), np.uint8)]
            off = np.zeros(len(group) + 1, np.int32)
            np.cumsum([len(s) for _, s in group], out=off[1:])
            h = self.ctx.kmm_scan(data, row_off, codes, off, budget, int(self.term_threshold))
            if len(h):
                h = h.copy()
                h[:, 2] = np.asarray([i for i, _ in group], np.int32)[h[:, 2]]
                all_hits.append(h)
        self.stats["scan_s"] = time.time() - t0
        if all_hits:
            h = np.concatenate(all_hits)
            h = h[np.lexsort((h[:, 1], h[:, 0], h[:, 2]))]           # SAM order: read by read
            for row, pos, pat, strand in h.tolist():
                primer = re.split(r"_\d+$", reads[pat][1])[0]        # BWT:249
                (reverse_dict if strand else forward_dict)[names[row]].append([pos, primer])
        print("Number of genes with candidate primers: forward ==> {}; reverse ==> {}.".format(len(forward_dict), len(reverse_dict)))
        both = set(forward_dict.keys()).intersection(reverse_dict.keys())
        target_gene = [g for g in dict.fromkeys(names) if g in both]
        print("Number of genes with candidate primer pairs: {}.".format(len(both)))
        return target_gene, forward_dict, reverse_dict

    def PCR_product(self, gene, F_dict, R_dict):
        """BWT:318-359; returns the lines instead of queueing them."""
        out = []
        product_len = self.PCR_size.split(",")
        primer_F = dict(F_dict[gene])
        position_start = sorted(primer_F.keys())
        primer_R = dict(R_dict[gene])
        position_stop = sorted(primer_R.keys())
        if int(position_stop[0]) - int(position_start[-1]) > int(product_len[1]):
            pass
        elif int(position_stop[-1]) - int(position_start[0]) < int(product_len[0]):
            pass
        else:
            for start in range(len(position_start)):
                stop_index_start, stop_index_stop = closest(position_stop, position_start[start] + int(product_len[0]),
                                                            position_start[start] + int(product_len[1]))
                if stop_index_start > stop_index_stop:
                    break
                for stop in range(stop_index_start, stop_index_stop + 1):
                    distance = int(position_stop[stop]) - int(position_start[start]) + 1
                    if distance > int(product_len[1]):
                        break
                    elif int(product_len[0]) < distance < int(product_len[1]):
                        out.append((gene, int(position_start[start]), int(position_stop[stop]), primer_F[position_start[start]],
                                    primer_R[position_stop[stop]], distance))
        return out

    def run(self):
        seq_ID = self.get_term()
        target_gene, forward_dict, reverse_dict = self.scan(seq_ID)
        primer_pair_id = defaultdict(int)
        primer_pair_acc = defaultdict(list)
        acc_id = set()
        with open(self.outfile, "w") as fo:
            fo.write("\t".join(["Chrom (or Genes)", "Start", "Stop", "Primer_F", "Primer_R", "Product length"]) + "\n")
            for gene in target_gene:
                for res in self.PCR_product(gene, forward_dict, reverse_dict):
                    primer_pair_id[res[3] + "\t" + res[4]] += 1
                    primer_pair_acc[res[3] + "\t" + res[4]].append(res[0])
                    acc_id.add(res[0])
                    fo.write("\t".join(map(str, res)) + "\n")
        primer_pair_id_sort = sorted(primer_pair_id.items(), key=lambda x: x[1], reverse=True)
        target_seq = set()
        with open(self.outfile + ".pair.num", "w") as fo:
            fo.write("Primer_F\tPrimer_R\tPair_num\ttarget accession number\n")
            for k in primer_pair_id_sort:
                primer_pair_acc_set = set(primer_pair_acc[k[0]])
                target_seq = target_seq.union(primer_pair_acc_set)
                fo.write(k[0] + "\t" + str(k[1]) + "\t" + str(len(primer_pair_acc_set)) + "\n")
        with open(self.outfile + ".total.acc.num", "w") as fo2:
            fo2.write("total coverage of primer set (PS) is: {}\n".format(len(acc_id)))
        if self.targets != "None":
            with open(self.outfile + ".unmatched.fa", "w") as out:
                with open(self.targets, "rb") as raw_total_seq_dict:
                    total_dict = pickle.load(raw_total_seq_dict)
                print(len(set(total_dict.keys())), len(target_seq))
                unmatched_seq_set = set(total_dict.keys()) - target_seq
                with open(self.outfile + ".total.acc.num", "a+") as fo3:
                    fo3.write("total target number is: {}\n".format(len(total_dict.keys())))
                for unmatch in sorted(unmatched_seq_set):
                    out.write(total_dict[unmatch])


def parse_args(argv=None):
    import argparse
    p = argparse.ArgumentParser(description="For mismatch coverage stastic (MI355X-native k-mismatch scan instead of bowtie2).")
    p.add_argument("-i", "--input", type=str, required=True, metavar="<file>", help="input file: primer.fa.")
    p.add_argument("-r", "--ref", type=str, required=True, metavar="<str>", help="Reference sequence file (FASTA).")
    p.add_argument("-l", "--len", type=int, default=0, metavar="<int>", help="Length of primer used for mapping (0: whole primer). Default: 0")
    p.add_argument("-t", "--term", type=int, default=4, metavar="<int>", help="Position of mismatch is not allowed in the 3 term of primer. Default: 4")
    p.add_argument("-s", "--size", type=str, default="100,1500", metavar="<str>", help="Length of PCR product, default: 100,1500.")
    p.add_argument("-p", "--proc", type=int, default=20, metavar="<int>", help="Accepted for compatibility (the scan runs on the GPU).")
    p.add_argument("-b", "--bowtie", type=str, default="bowtie2", metavar="<str>", help="Accepted for compatibility: no external mapper is run.")
    p.add_argument("-m", "--seedmms", type=int, default=1, metavar="<int>", help="bowtie seed mismatches: sensitivity only; the scan is exhaustive.")
    p.add_argument("-d", "--dict", type=str, default="None", metavar="<str>", help="Dictionary of targets sequences, binary format (prepare_fa_pickle.py).")
    p.add_argument("-o", "--out", type=str, required=True, metavar="<file>", help="Output file: Prodcut of PCR product with primers.")
    p.add_argument("--max-mismatch", type=int, default=None, help="mismatches per alignment (default: bowtie2's budget floor((0.6 + 0.6 L) / 6))")
    p.add_argument("--device", type=int, default=0)
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    e1 = time.time()
    off_targets(primer_file=args.input, term_length=args.len, reference_file=args.ref, PCR_product_size=args.size,
                mismatch_num=args.seedmms, outfile=args.out, term_threshold=args.term, bowtie=args.bowtie, nproc=args.proc,
                targets=args.dict, device=args.device, max_mismatch=args.max_mismatch).run()
    e2 = time.time()
    print("INFO {} Total times: {}".format(time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(time.time())), round(float(e2 - e1), 2)))
