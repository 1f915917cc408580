"""Greedy minimal primer-set cover — drop-in for scripts/get_Maxprimerset.py
(get_Maxprimerset_V1.3.py, "MS"): clusters sorted by number of candidate pairs, for each cluster
the first pair that forms no 3'-end dimer with the already selected set is taken.

The control flow below restates MS:218-356 (including what the reference actually does after a
back-track in the maximum mode: the outer `for` resumes at the row after the one that failed,
MS:257-277).  The dimer test itself (`dimer_examination`, MS:193-215, quadratic in the size of
the selected set in the reference) runs on the GPU and only over pairs that involve a new
primer (dimer.DimerExaminer).
"""
from __future__ import annotations

import re
import sys
from optparse import OptionParser

from . import iupac
from ._abi import Library
from .dimer import DimerExaminer

COLUMNS = ["#Primer", "Primer_rank", "Primer_F", "Primer_R", "PCR_product (Length:Tm:Coverage)",
           "Coverage number with error in top N", "Primer position (representative sequence)"]


def _write_clique(path, rows):
    """pandas' DataFrame.to_csv(sep='\\t', index=False) of the reference's `clique` frame: missing
    values print as empty fields."""
    with open(path, "w") as f:
        f.write("\t".join(COLUMNS) + "\n")
        for r in rows:
            f.write("\t".join("" if r.get(c) is None else str(r[c]) for c in COLUMNS) + "\n")


def _row(primers, r, c):
    p = primers[r]
    return {"#Primer": p[0], "Primer_rank": str(c), "Primer_F": p[c], "Primer_R": p[c + 1],
            "PCR_product (Length:Tm:Coverage)": p[c + 2], "Coverage number with error in top N": p[c + 3],
            "Primer position (representative sequence)": p[c + 4]}


class PrimerSetCover:
    def __init__(self, primers, step, examiner: DimerExaminer):
        self.primers = primers
        self.step = step
        self.ex = examiner

    def _dimer(self, f, r, primer_set):
        return self.ex.any_dimer(iupac.expand(f) + iupac.expand(r), sorted(primer_set))

    @staticmethod
    def _grow(primer_set, f, r):
        return primer_set | set(iupac.expand(f) + iupac.expand(r))

    def maximal(self, output, next_candidate):
        """greedy_maximal_primers (MS:291-356): a cluster whose pairs all dimerise is skipped and
        written to the .next file."""
        primers, step = self.primers, self.step
        primer_set = set()
        clique = []
        r, c = 0, 1
        while r < len(primers):
            if len(primers[r]) <= 1:
                print("Non primers: virus {} missing!".format(primers[r][0]))
                next_candidate.write("\t".join(primers[r]) + "\n")
                r, c = r + 1, 1
                continue
            while c <= len(primers[r]) - step:
                if self._dimer(primers[r][c], primers[r][c + 1], primer_set):
                    c += step
                    if c > len(primers[r]) - step:
                        clique.append({"#Primer": primers[r][0]})
                        print("virus {} missing!".format(primers[r][0]))
                        next_candidate.write("\t".join(primers[r]) + "\n")
                        r, c = r + 1, 1
                        break
                else:
                    clique.append(_row(primers, r, c))
                    primer_set = self._grow(primer_set, primers[r][c], primers[r][c + 1])
                    r, c = r + 1, 1
                    break
            else:
                # a row too short to hold one pair never enters the inner loop and never advances in
                # the reference either (it would spin); treat it like an exhausted cluster
                raise ValueError(f"row {r} has {len(primers[r])} fields: not a multiple of step")
        _write_clique(output, clique)

    def maximum(self, output):
        """greedy_primers (MS:218-282): back-track to the previous cluster's next pair when a
        cluster is exhausted; exit(1) when the first cluster is exhausted."""
        primers, step = self.primers, self.step
        primer_set = set()
        saved_set, jdict = {}, {}
        clique = []
        blank_row = 0
        c = 1
        for row in range(len(primers)):
            r = row
            if len(primers[r]) <= 1:
                blank_row += 1
                continue
            while c <= len(primers[r]) - step:
                if self._dimer(primers[r][c], primers[r][c + 1], primer_set):
                    c += step
                    while c > len(primers[r]) - step:                    # backtrack_to_previous_row, MS:246-255
                        r -= 1
                        if r < blank_row:
                            print("Non maximum primer set. Try maximal primer set!")
                            sys.exit(1)
                        if r not in jdict:
                            # The reference reads jdict[row_pointer] of a cluster it never selected from (a blank cluster, or one
                            # the search already left: the for-loop's row pointer and the back-tracking one diverge) and dies
                            # with KeyError, exit status 1, nothing written (tests/golden/maxset_multi.json.gz, seeds 3 and 7).
                            # Same status here, with a message instead of a traceback.
                            print("KeyError: {} (the maximum-set search cannot back-track past this cluster; "
                                  "the reference script fails here too)".format(r), file=sys.stderr)
                            sys.exit(1)
                        c = jdict[r] + step
                        primer_set = saved_set[r]
                        clique.pop()
                else:
                    clique.append(_row(primers, r, c))
                    saved_set[r] = primer_set
                    primer_set = self._grow(primer_set, primers[r][c], primers[r][c + 1])
                    jdict[r] = c
                    c = 1
                    break
        _write_clique(output, clique)


def parse_args(argv=None):
    parser = OptionParser("Usage: %prog -i [input] -o [output] \n Options: {-s [step] -m [T]}", version="%prog 0.0.4")
    parser.add_option("-i", "--input", dest="input", help="Input file: primers")
    parser.add_option("-a", "--adaptor", dest="adaptor", type="str",
                      default="TCTTTCCCTACACGACGCTCTTCCGATCT,TCTTTCCCTACACGACGCTCTTCCGATCT",
                      help="Adaptor sequence (accepted for compatibility; the reference does not use it either)")
    parser.add_option("-s", "--step", dest="step", default=5, type="int",
                      help="distance between primers; column number of primer1_F to primer2_F.")
    parser.add_option("-m", "--method", dest="method", default="T", type="str",
                      help="which method: maximal or maximum. If -m [T] use maximal; else maximum")
    parser.add_option("-o", "--out", dest="out", help="Prefix of out file: candidate primers")
    parser.add_option("--device", dest="device", default=0, type="int", help="GPU ordinal")
    options, _ = parser.parse_args(argv)
    if options.input is None:
        parser.print_help()
        print("Input file must be specified !!!")
        sys.exit(1)
    if options.out is None:
        parser.print_help()
        print("No output file provided !!!")
        sys.exit(1)
    return options


def run(options, library: Library | None = None):
    if re.search("/", options.input):                                        # MS:363-367
        parts = options.input.split("/")
        sort = "/".join(parts[:-1]) + "/sort." + parts[-1]
    else:
        sort = "sort." + options.input
    with open(options.input, "r") as primers_file, open(sort, "w") as f:
        primers = list(sorted([list(filter(None, line.strip().split("\t"))) for line in primers_file], key=len))
        for i in primers:
            f.write("\t".join(i) + "\n")
    lib = library if library is not None else Library()
    cover = PrimerSetCover(primers, options.step, DimerExaminer(lib.context(options.device), threshold=3.0))
    if options.method == "T":
        next_candidate = options.out.rstrip(".xls") + ".next.xls"             # MS:375 (strips characters, not a suffix)
        with open(next_candidate, "w") as nxt:
            cover.maximal(options.out, nxt)
    else:
        cover.maximum(options.out)


def main(argv=None):
    from ._abi import prefer_staged_copies
    prefer_staged_copies()                      # a command line owns its process: see _abi.prefer_staged_copies
    run(parse_args(argv))


if __name__ == "__main__":
    main()
