"""Exact in-silico PCR — drop-in for scripts/extract_PCR_product.py (extract_PCR_product_V1.py, "PCR"), the
step that checks the final primer set against ALL input sequences.

The search over every (primer pair, sequence) — first forward expansion that occurs and whose "Product"
(up to its next occurrence) holds a reverse-complemented reverse expansion — runs in one `mp_pcr_scan`
launch; the host only slices the amplicon strings and writes the reference's files.
"""
from __future__ import annotations

import os
import sys
import time
from pathlib import Path

import numpy as np

from . import iupac
from ._abi import Library
from .dimer import PATTERN_MAX_LEN, encode_primers


class Product(object):
    """Drop-in for the reference class of the same name (PCR:118-259)."""

    def __init__(self, primer_file="", output_file="", ref_file="", file_format="fa", coverage="", nproc=10, *,
                 library: Library | None = None, device: int = 0):
        self.nproc = nproc
        self.primers_file = primer_file
        self.ref_file = ref_file
        self.output_file = Path(output_file)
        self.file_format = file_format
        self.primers = self.parse_primers()
        self.coverage = coverage
        self.lib = library if library is not None else Library()
        self.ctx = self.lib.context(device)
        self.stats = {}

    def parse_primers(self):
        """PCR:135-169: `xls` (final_maxprimers_set.xls), `fa` (F/R records alternating) or `seq` ("F,R")."""
        res = {}
        if self.file_format == "seq":
            primers = self.primers_file.split(",")
            res["PCR_info"] = [primers[0], primers[1]]
            return res
        with open(self.primers_file, "r") as f:
            if self.file_format == "xls":
                for line in f:
                    if line.startswith("#"):
                        continue
                    i = line.strip().split("\t")
                    cluster_id = i[0].split("/")[-1].split(".")[0]
                    start, stop = i[6].split(":")[0], i[6].split(":")[1]
                    res[cluster_id + "_" + str(start) + "_F_" + cluster_id + "_" + str(stop)] = [i[2], i[3]]
            elif self.file_format == "fa":
                rows = [ln.rstrip("\n") for ln in f if ln.strip()]          # pandas.read_table skips blank lines
                for idx, row in enumerate(rows):
                    if idx % 4 == 0:
                        f_info = row.lstrip(">")
                    elif idx % 4 == 1:
                        primer_f = row
                    elif idx % 4 == 2:
                        key = f_info + "_" + row.lstrip(">")
                    else:
                        res[key] = [primer_f, row]
        return res

    def _read_ref(self):
        """Header and sequence lines as the reference walks them (PCR:194-197): every non-'>' line is searched
        on its own and belongs to the last header seen."""
        keys, seqs = [], []
        key = None
        with open(self.ref_file, "r") as r:
            for line in r:
                if line.startswith(">"):
                    key = line.strip()
                else:
                    if key is None:
                        raise NameError("sequence line before the first header")
                    keys.append(key)
                    seqs.append(line)
        return keys, seqs

    def run(self):
        if not self.output_file.exists():
            os.makedirs(self.output_file, exist_ok=True)
        keys, seqs = self._read_ref()
        bodies = [s.rstrip("\n").encode("latin-1") for s in seqs]
        row_off = np.zeros(len(bodies) + 1, np.int64)
        np.cumsum([len(b) for b in bodies], out=row_off[1:])
        data = np.frombuffer(b"".join(bodies), dtype=np.uint8) if bodies else np.zeros(0, np.uint8)
        names = list(self.primers.keys())
        flat = [s for n in names for s in self.primers[n]]
        codes, off = encode_primers(flat, PATTERN_MAX_LEN) if flat else (np.zeros(0, np.uint8), np.zeros(1, np.int32))
        t0 = time.time()
        if names and bodies:
            # the database is uploaded and packed once (mp_seq_load) and stays in the context: a caller that hands its context on (a
            # validation step on the same reference) scans the stored words again instead of sending the text a second time
            self.ctx.seq_load(data, row_off)
            hits = self.ctx.pcr_scan_resident(codes, off)
        else:
            hits = np.full((len(names), len(bodies), 4), -1, np.int32)   # -1 = no amplicon (0 would read "expansion 0 at position 0")
        self.stats["scan_s"] = time.time() - t0
        product_ids = set()
        stripped = [line.strip() for line in seqs]                                     # what a non-target record prints (PCR:211)
        for pi, name in enumerate(names):
            F, R = self.primers[name]
            r_exp = [iupac.revcomp(x) for x in iupac.expand(R)]
            product_dict, non_targets = {}, {}
            h = hits[pi]
            i_f, p1, i_r, q = h[:, 0].tolist(), h[:, 1].tolist(), h[:, 2].tolist(), h[:, 3].tolist()
            for row, key in enumerate(keys):                                           # dicts: a later line of a record replaces an earlier one
                if i_f[row] >= 0:
                    product_dict[key] = seqs[row][p1[row]:q[row]].strip() + r_exp[i_r[row]]   # PCR:204-205
                else:
                    non_targets[key] = stripped[row]
            pcr_product = Path(self.output_file).joinpath(name).with_suffix(".PCR.product.fa")
            pcr_non_product = Path(self.output_file).joinpath(name).with_suffix(".non_PCR.product.fa")
            with open(self.coverage, "a+") as c:                                       # PCR:228-232 (appends)
                c.write("Number of Product/non_Product, primer-F and primer-R: {}\t{}\t{}\t{}\t{}\n".format(
                    name, len(product_dict), len(non_targets), F, R))
            product_ids.update(product_dict)
            with open(pcr_product, "w") as p:
                p.write("".join([k + "\n" + v + "\n" for k, v in product_dict.items()]))
            with open(pcr_non_product, "w") as p:
                p.write("".join([k + "\n" + v + "\n" for k, v in non_targets.items()]))
        with open(self.ref_file, encoding="utf-8") as f:
            seq_number = int(f.read().count("\n") / 2)
        with open(self.coverage, "a+") as c:
            c.write("Total number of sequences:\t{}\nCoveraged number of sequence:\t{}\nRate of coverage:\t>= {}\n".format(
                seq_number, len(product_ids), round(float(len(product_ids)) / seq_number, 2)))


def parse_args(argv=None):
    from optparse import OptionParser
    parser = OptionParser("Usage: %prog -r [input] -i [primerF,primerR] -f [format] -o [output]", version="%prog 0.0.2")
    parser.add_option("-r", "--ref", dest="ref", help="reference file: template fasta or reference fasta.")
    parser.add_option("-i", "--input", dest="input", help="Primer file: final_maxprimers_set.xls, primer.fa or primer_F,primer_R.")
    parser.add_option("-f", "--format", dest="format", help="Format of primer file: xls or fa or seq.")
    parser.add_option("-o", "--out", dest="out", default="PCR_product", help="Output_dir. default: PCR_product.")
    parser.add_option("-p", "--process", dest="process", default="10", type="int", help="Accepted for compatibility.")
    parser.add_option("-s", "--stast", dest="stast", default="Coverage.xls", help="Stast information. default: Coverage.xls")
    parser.add_option("--device", dest="device", default=0, type="int", help="GPU ordinal")
    options, _ = parser.parse_args(argv)
    for val, msg in ((options.ref, "Input (reference) file must be specified !!!"), (options.input, "Primer file or sequence must be specified !!!"),
                     (options.format, "Primer file format must be specified !!!")):
        if val is None:
            parser.print_help()
            print(msg)
            sys.exit(1)
    return options


def main(argv=None):
    from ._abi import prefer_staged_copies
    prefer_staged_copies()                      # a command line owns its process: see _abi.prefer_staged_copies
    o = parse_args(argv)
    e1 = time.time()
    Product(primer_file=o.input, output_file=o.out, ref_file=o.ref, file_format=o.format, coverage=o.stast, nproc=o.process,
            device=o.device).run()
    e2 = time.time()
    print("INFO {} Total times: {}".format(time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(time.time())),
                                           round(float(e2 - e1), 2)))


if __name__ == "__main__":
    main()
