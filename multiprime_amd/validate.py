"""Drop-in for scripts/primer_coverage_validation_by_BWT_V9.py ("V9") — SURVEY §8f-3: which primer pairs of a primer set
amplify which sequences of a database when a few mismatches are tolerated.  The reference maps the expanded 3' terms with
bowtie2, splits the SAM by strand with samtools (V9:264-286) and pairs the surviving sites per sequence; here the mapping is
ONE exhaustive GPU scan (mp_kmm_scan, csrc/scan.hip) and the rest is restated in three stages of its own:

  TermTable    the reads: every distinct 3' term of the primer file, expanded (V9:205-239)  ->  <primers>.term.fa
  sites        {strand: {sequence: {start: primer}}} — from the GPU scan, or, exactly as the reference does when it finds them
               (V9:270-271), from existing <primers>.for.sam / .rev.sam files filtered by the MD:Z rule of build_dict (V9:241-262)
  amplicons    per sequence, the (forward start, reverse start) combinations inside the size range (V9:318-345), vectorised
  reports      <out>, <out>.pair.num, <out>.total.acc.num, <out>.unmatched.fa (V9:377-398)

Same class name (`off_targets`), constructor arguments, flags and output files as the reference script.

PARITY: everything around the mapper is pinned to the reference — tests/golden/validate.json.gz holds what the unmodified V9
class writes for hand-written and seeded SAM input (tests/golden/make_golden_validate.py), this module reproduces it from the
same SAM text, and the GPU scan is checked end to end against the same stages fed with the SAM lines its hits stand for.  The
MAPPER: bowtie2 / samtools are not installed in the authoring image, so the mapping step is replaced by its acceptance rule
under default end-to-end scoring — an ungapped alignment with at most floor((0.6 + 0.6 L) / 6) mismatches (minimum score
-0.6 - 0.6 L at 6 per mismatch; `--max-mismatch` overrides), both strands, every site (`-a`).  That rule is pinned to the
reference author's own bowtie2 + samtools run of rule BWT_validation (multiPrime.py:441-457), whose output ships in the reference
tree: all 1158 sequences whose text is available (485 with a product row, 673 without) are decided identically, and the
neighbouring rules (budget 0 / 2, 3'-term threshold 0 / 2) are not (tests/golden/make_golden_bwt.py, tests/test_validate_bwt.py).
Beyond what that run exercises: no gapped alignments here (bowtie2 admits a single 1-base gap at L >= 13: score -8 against a
minimum of -0.6 - 0.6 L); sites bowtie2's seed heuristics (-N, -L 8) miss are found here.

Orders the reference takes from a Python set (sequences in <out>, names inside a shared term id, unmatched records) are
deterministic here: first appearance in the forward sites / the primer file / sorted names.
"""
from __future__ import annotations

import os
import sys
import pickle
import re
import time
from pathlib import Path

import numpy as np

from . import iupac
from ._abi import Library
from .dimer import PATTERN_MAX_LEN

_READ_INDEX = re.compile(r"_\d+$")          # a read name ends in the index of its expansion (V9:249)
_MD_TAG = re.compile(r"MD:Z:(\w+)")
_DIGITS = re.compile(r"\d+")


def degenerate_seq(primer: str):
    """Concrete sequences of a degenerate one in itertools.product order (V9:193-203); other characters stay as they are."""
    out = [""]
    for ch in primer:
        out = [a + b for a in out for b in (iupac.MEMBERS[ch] if ch in "RYMKSWHBVDN" else ch)]
    return out


def bowtie2_mismatch_budget(length: int) -> int:
    """Mismatches bowtie2 --end-to-end admits with default scoring: min score -0.6 - 0.6 L, 6 per mismatch (high quality)."""
    return int((0.6 + 0.6 * length) // 6)


class TermTable:
    """The reads of the mapping step: expanded 3' terms -> read name (V9:205-239).  A line of the primer file that starts with
    '>' names the lines after it; every other line is a primer (the reference does not join wrapped records, nor does this)."""

    def __init__(self, primer_file, term_len):
        owners = {}                                  # term -> primer names, file order
        name = None
        with open(primer_file) as f:
            for line in f:
                text = line.strip()
                if line.startswith(">"):
                    name = text.lstrip(">")
                else:
                    owners.setdefault(text if term_len == 0 else text[-term_len:], []).append(name)
        self.reads = {}                              # concrete sequence -> the ids that expand to it
        for term, names in owners.items():
            stem = "_".join(dict.fromkeys(names))
            for j, seq in enumerate(degenerate_seq(term)):
                self.reads.setdefault(seq, []).append(f"{stem}_{j}")

    def names(self):
        return ["_".join(ids) for ids in self.reads.values()]

    def write(self, path):
        with open(path, "w") as fo:
            fo.writelines(f">{name}\n{seq}\n" for seq, name in zip(self.reads, self.names()))


def sites_of_sam(path, threshold):
    """{sequence: {0-based start: primer}} of one strand's SAM file under build_dict's rule (V9:241-262): an alignment counts
    when the number its MD:Z tag ENDS with — read from the tag's last two characters only, so 12 for "5A12" but 0 for "15C10"
    — is at least `threshold`; lines without the tag (unaligned reads) do not count.  Later lines replace earlier ones at the
    same start (the reference turns its list into a dict, V9:320-322)."""
    sites = {}
    with open(path) as f:
        for line in f:
            col = line.strip().split("\t")
            tag = _MD_TAG.search("\t".join(col[11:]))
            if tag and int(_DIGITS.search(tag.group(1)[-2:]).group()) >= threshold:
                sites.setdefault(col[2], {})[int(col[3]) - 1] = _READ_INDEX.split(col[0])[0]
    return sites


def amplicons(forward, reverse, size_lo, size_hi):
    """(start, stop, forward primer, reverse primer, length) of one sequence, in the reference's order (V9:318-345): starts
    ascending, stops ascending, length = stop - start + 1 strictly inside (size_lo, size_hi).  Two quirks are kept: no product
    at all when the sites cannot be closer than size_hi or farther than size_lo as a whole, and the FIRST start without a
    reverse site in [start + size_lo, start + size_hi) ends the search for the later starts as well."""
    starts = np.fromiter(sorted(forward), np.int64, len(forward))
    stops = np.fromiter(sorted(reverse), np.int64, len(reverse))
    if stops[0] - starts[-1] > size_hi or stops[-1] - starts[0] < size_lo:
        return []
    first = np.searchsorted(stops, starts + size_lo, "left")
    last = np.where(starts + size_hi > stops[-1], len(stops) - 1, np.searchsorted(stops, starts + size_hi, "left") - 1)
    dead = np.nonzero(first > last)[0]
    n_live = int(dead[0]) if len(dead) else len(starts)
    # stop - start + 1 < size_hi; from `first` on stop >= start + size_lo, so the lower bound holds already
    last = np.minimum(last, np.searchsorted(stops, starts + size_hi - 1, "left") - 1)
    out = []
    for i in range(n_live):
        a = int(starts[i])
        for b in stops[first[i]:last[i] + 1].tolist():
            if b - a + 1 > size_lo:
                out.append((a, b, forward[a], reverse[b], b - a + 1))
    return out


class off_targets(object):
    def __init__(self, primer_file, term_length, reference_file, PCR_product_size, mismatch_num, outfile, term_threshold,
                 bowtie="bowtie2", nproc=20, targets="None", *, library: Library | None = None, device: int = 0, max_mismatch=None):
        self.bowtie = bowtie                    # accepted for compatibility: no external mapper is run
        self.term_threshold = int(term_threshold)
        self.nproc = nproc
        self.term_len = term_length
        self.primer_file = primer_file
        self.reference_file = reference_file
        self.outfile = outfile
        self.PCR_size = PCR_product_size
        self.mismatch_num = mismatch_num        # bowtie's -N / -n: seed sensitivity only, the scan is exhaustive
        self.targets = targets
        self.max_mismatch = max_mismatch
        self._library, self._device = library, device
        self.stats = {}

    def _beside_primers(self, suffix):
        return Path(self.primer_file).parent.joinpath(Path(self.primer_file).stem).with_suffix(suffix)

    # -- sites from the GPU ------------------------------------------------------------------------------------------------
    def scan(self, table: TermTable):
        """Both strands' sites from one exhaustive k-mismatch scan of the reference FASTA (replaces V9:264-316)."""
        from .host import Fasta
        path = str(self.reference_file)
        if not os.path.exists(path):
            raise FileNotFoundError(path + ": the scan needs the reference FASTA itself (a bowtie index prefix is not enough)")
        t0 = time.time()
        fa = Fasta(path)
        data, row_off = fa.rows()
        genes = [s[1:] if s.startswith(">") else s for s in fa.ids]          # a mapper names a sequence by its first token
        seqs, names = list(table.reads), table.names()
        # The reference hands every read to bowtie2, which ignores what it cannot align (an empty read from a blank line of the
        # primer file — V9 get_term keeps it as key "" —, a read with U / I / N left after expansion, a read shorter than a seed).
        # This build does the same: such reads find nothing, with a warning naming them; only when NO read is usable is that an
        # error.  A read BEYOND the packed pattern width is different: bowtie2 would map it, the scan cannot — reporting "no
        # off-target" for a read that was never looked for would be wrong, so that is a hard error naming the read
        # ("Limits": use -l to map the 3' term of longer primers).
        too_long = [names[i] for i, seq in enumerate(seqs) if len(seq) > PATTERN_MAX_LEN]
        if too_long:
            raise ValueError("read(s) longer than the scan's {} bases: {} — map the 3' term instead (-l)".format(
                PATTERN_MAX_LEN, ", ".join(repr(n) for n in too_long[:20]) + (" ..." if len(too_long) > 20 else "")))
        usable = [i for i, seq in enumerate(seqs) if len(seq) >= 4 and not (set(seq.upper()) - set("ACGT"))]
        usable_set = set(usable)
        skipped = [names[i] for i in range(len(seqs)) if i not in usable_set]
        if skipped:
            print("Warning: {} read(s) not scanned (empty, shorter than 4 bases, or not ACGT after expansion), as bowtie2 ignores them: {}".format(
                len(skipped), ", ".join(repr(n) for n in skipped[:20]) + (" ..." if len(skipped) > 20 else "")),
                file=sys.stderr)
        if not usable:
            raise ValueError(f"no usable read: the scan takes 4..{PATTERN_MAX_LEN} bases of ACGT after expansion")
        lib = self._library if self._library is not None else Library()
        ctx = lib.context(self._device)
        found = []
        try:
            # the database goes to the device ONCE, packed (mp_seq_load: 4 bits per base beside the characters); every mismatch budget
            # scans the stored words (rounds 2-5 sent the ASCII text with every call)
            ctx.seq_load(data, row_off)
            budgets = {}
            for i in usable:
                budgets.setdefault(self.max_mismatch if self.max_mismatch is not None else bowtie2_mismatch_budget(len(seqs[i])), []).append(i)
            for budget, members in budgets.items():
                codes = iupac.MASK_LUT[np.frombuffer("".join(seqs[i].upper() for i in members).encode(), np.uint8)]
                off = np.zeros(len(members) + 1, np.int32)
                np.cumsum([len(seqs[i]) for i in members], out=off[1:])
                h = ctx.kmm_scan_resident(codes, off, budget, self.term_threshold)
                if len(h):
                    h = h.copy()
                    h[:, 2] = np.asarray(members, np.int32)[h[:, 2]]
                    found.append(h)
        finally:
            ctx.close()
        self.stats["scan_s"] = time.time() - t0
        sites = ({}, {})
        if found:
            h = np.concatenate(found)
            h = h[np.lexsort((h[:, 1], h[:, 0], h[:, 2]))]                  # read by read, as a mapper reports them
            primers = [_READ_INDEX.split(n)[0] for n in names]
            for row, pos, read, strand in h.tolist():
                sites[strand].setdefault(genes[row], {})[pos] = primers[read]
        return sites

    # -- reports -------------------------------------------------------------------------------------------------------------
    def _report(self, forward, reverse):
        print("Number of genes with candidate primers: forward ==> {}; reverse ==> {}.".format(len(forward), len(reverse)))
        both = [g for g in forward if g in reverse]
        print("Number of genes with candidate primer pairs: {}.".format(len(both)))
        lo, hi = (int(x) for x in self.PCR_size.split(",")[:2])
        per_pair = {}                                                       # "F<TAB>R" -> [products, {sequences}]
        covered = set()
        with open(self.outfile, "w") as fo:
            fo.write("Chrom (or Genes)\tStart\tStop\tPrimer_F\tPrimer_R\tProduct length\n")
            for gene in both:
                for start, stop, pf, pr, length in amplicons(forward[gene], reverse[gene], lo, hi):
                    fo.write(f"{gene}\t{start}\t{stop}\t{pf}\t{pr}\t{length}\n")
                    tally = per_pair.setdefault(pf + "\t" + pr, [0, set()])
                    tally[0] += 1
                    tally[1].add(gene)
                    covered.add(gene)
        with open(self.outfile + ".pair.num", "w") as fo:
            fo.write("Primer_F\tPrimer_R\tPair_num\ttarget accession number\n")
            for pair, (n, genes) in sorted(per_pair.items(), key=lambda kv: kv[1][0], reverse=True):
                fo.write(f"{pair}\t{n}\t{len(genes)}\n")
        with open(self.outfile + ".total.acc.num", "w") as fo:
            fo.write("total coverage of primer set (PS) is: {}\n".format(len(covered)))
            if self.targets != "None":
                with open(self.targets, "rb") as f:
                    records = pickle.load(f)                                # {name: FASTA record text} (prepare_fa_pickle.py)
                print(len(records), len(covered))
                fo.write("total target number is: {}\n".format(len(records)))
                with open(self.outfile + ".unmatched.fa", "w") as out:
                    out.writelines(records[name] for name in sorted(set(records) - covered))

    def run(self):
        table = TermTable(self.primer_file, self.term_len)
        table.write(self._beside_primers(".term.fa"))
        sams = [self._beside_primers(".for.sam"), self._beside_primers(".rev.sam")]
        if all(p.exists() for p in sams):                                   # V9:270-271: existing SAM files are used, nothing is mapped
            forward, reverse = (sites_of_sam(p, self.term_threshold) for p in sams)
        else:
            forward, reverse = self.scan(table)
        self._report(forward, reverse)


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
    from ._abi import prefer_staged_copies
    prefer_staged_copies()                      # a command line owns its process: see _abi.prefer_staged_copies
    args = parse_args(argv)
    e1 = time.time()
    off_targets(primer_file=args.input, term_length=args.len, reference_file=args.ref, PCR_product_size=args.size,
                mismatch_num=args.seedmms, outfile=args.out, term_threshold=args.term, bowtie=args.bowtie, nproc=args.proc,
                targets=args.dict, device=args.device, max_mismatch=args.max_mismatch).run()
    e2 = time.time()
    print("INFO {} Total times: {}".format(time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(time.time())), round(float(e2 - e1), 2)))
