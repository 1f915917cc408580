#!/usr/bin/env python3
"""Goldens for the FASTA record parser's white-space handling: small files whose lines start / end with the characters str.strip()
removes in text mode - ASCII blanks, the separators 0x1c-0x1f, and the non-ASCII white space of str.isspace() (U+0085, U+00A0, U+1680,
U+2000-200A, U+2028, U+2029, U+202F, U+205F, U+3000) - run through the UNMODIFIED reference's parse_seq (multiPrime-core_V20.py:441-455;
the method does not touch `self`).  Stored per case: the file bytes (as latin-1 text), the ids in order and the sequences as the
reference returns them (after its per-character mapping).  Characters outside ASCII INSIDE a sequence are left out of the cases: the
reference maps one CHARACTER to one '-', this build one BYTE (limits).
Usage: python tests/golden/make_golden_parser.py   -> tests/golden/parser_ws.json"""
import importlib.util
import json
import os
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
V20 = "/root/reference/scripts/multiPrime-core_V20.py"

# code points of the white space tried at line ends (0 = none); the last two entries are sequences of several
WS_CODES = [[], [0x20], [0xA0], [0x85], [0x1680], [0x2000], [0x2003], [0x200A], [0x2028], [0x2029], [0x202F], [0x205F], [0x3000],
            [0x1C], [0x1D], [0x1E], [0x1F], [0x0B], [0x0C], [0x20, 0xA0, 0x20], [0x09, 0x3000]]
NL, CR = chr(10), chr(13)


def cases():
    out = []
    for i, codes in enumerate(WS_CODES):
        w = "".join(chr(c) for c in codes)
        out.append((">id%d desc%s" % (i, w) + NL + w + "ACGT" + w + NL + "AC-N" + w + w + NL).encode("utf-8"))
        # a line that does not START with '>' is data: with white space in front of the first header the reference has no id yet
        out.append((w + ">x" + NL + ">id%d%s" % (i, w) + NL + "acgt" + w + CR + NL + w + w + "TTGA" + CR + ">id%d" % i + NL + "GG" + w).encode("utf-8"))
    nbsp, ideo, lsep = chr(0xA0), chr(0x3000), chr(0x2028)
    out.append((">a" + nbsp + "b c" + NL + "AC" + nbsp + NL + ">a" + nbsp + "b" + NL + "GT" + NL).encode("utf-8"))      # NBSP inside the id token stays (split(" "))
    out.append((">k" + NL + ideo + ideo + NL + lsep + NL + "AC" + NL).encode("utf-8"))                               # lines of white space only: empty pieces
    out.append(("#c" + nbsp + NL + ">z" + NL + nbsp + "#AC" + NL).encode("utf-8"))                                   # '#' only counts in column 0
    return out


def main():
    spec = importlib.util.spec_from_file_location("v20", V20)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    recs = []
    with tempfile.TemporaryDirectory() as td:
        for raw in cases():
            p = os.path.join(td, "x.fa")
            with open(p, "wb") as f:
                f.write(raw)
            try:
                d, n = mod.NN_degenerate.parse_seq(None, p)
                recs.append({"file_latin1": raw.decode("latin-1"), "ids": list(d.keys()), "seqs": list(d.values())})
            except Exception as e:            # (e.g. data before the first header: NameError / UnboundLocalError in the reference)
                recs.append({"file_latin1": raw.decode("latin-1"), "error": type(e).__name__})
    with open(os.path.join(HERE, "parser_ws.json"), "w") as f:
        json.dump(recs, f, ensure_ascii=True, indent=0)
    print(len(recs), "cases;", sum("error" in r for r in recs), "end in an exception of the reference")


if __name__ == "__main__":
    main()
