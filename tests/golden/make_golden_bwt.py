#!/usr/bin/env python3
"""Pins the MAPPER of SURVEY §8f-3 to the reference author's own bowtie2 + samtools run.

bowtie2 / samtools are absent from the authoring image, so the mapper cannot be run here — but its OUTPUT is in the reference tree:
rule BWT_validation (multiPrime.py:441-457: `primer_coverage_validation_by_BWT.py -i core_final_maxprimers_set.fa -r Total_fa/...
-l 18 -t 1 -s 50,2000`) was run by the author on all 20 727 sequences and left

    test_data/results/Core_primers_set/BWT_coverage/core_final_maxprimers_set.out               20 054 rows (accession, Start, Stop, F, R, length)
    test_data/results/Core_primers_set/BWT_coverage/core_final_maxprimers_set.out.unmatched.fa     673 records (no product)
    test_data/results/Core_primers_set/core_final_maxprimers_set.fa                             the two primers

`bowtie2 -a` reports every alignment of every read, so what it decides for one reference sequence does not depend on the others:
the rows of any subset of the database are the run's decisions for that subset.  Sequences whose text the tree also holds:

  * the 500 records of test_data/results/Clusters_fa/Cluster_0_20727.tfa (already a committed input): 485 have a row, 15 are in
    unmatched.fa — their record text there equals the .tfa's (checked below);
  * all 673 records of unmatched.fa itself (658 more negatives): committed as tests/golden/inputs/bwt_unmatched.fa.gz.

Stored: tests/golden/bwt_cluster0.json.gz = {flags, primers, rows: {accession: [[start, stop, F, R, length], ...]},
unmatched_in_cluster: [...], unmatched_all: [...]}.  tests/test_validate_bwt.py requires the drop-in to reproduce exactly these
decisions and shows that the fixture discriminates (budget 0 / 2 mismatches and 3'-term thresholds 0 / 2 all give other answers).

Run in the authoring container (reads /root/reference, writes only under tests/golden/):  python tests/golden/make_golden_bwt.py
"""
import gzip
import json
import os

REF = "/root/reference/test_data/results"
HERE = os.path.dirname(os.path.abspath(__file__))


def records(path):
    out, name = {}, None
    for line in open(path):
        if line.startswith(">"):
            name = line[1:].split()[0]
            out[name] = ""
        else:
            out[name] += line.strip()
    return out


def main():
    out_file = os.path.join(REF, "Core_primers_set", "BWT_coverage", "core_final_maxprimers_set.out")
    cluster = records(os.path.join(REF, "Clusters_fa", "Cluster_0_20727.tfa"))
    unmatched = records(out_file + ".unmatched.fa")
    lines = open(out_file).read().splitlines()
    assert lines[0].split("\t") == ["Chrom (or Genes)", "Start", "Stop", "Primer_F", "Primer_R", "Product length"]
    rows, seen = {}, set()
    for line in lines[1:]:
        c = line.split("\t")
        seen.add(c[0])
        if c[0] in cluster:
            rows.setdefault(c[0], []).append([int(c[1]), int(c[2]), c[3], c[4], int(c[5])])
    un_cluster = sorted(set(cluster) & set(unmatched))
    # every sequence of the cluster is classified by the run, exactly once; the negatives' text is the same in both files
    assert len(cluster) == 500 and len(rows) + len(un_cluster) == 500 and not (set(rows) & set(un_cluster))
    assert all(cluster[a] == unmatched[a] for a in un_cluster)
    assert not (seen & set(unmatched)) and len(seen) == 20054 and len(unmatched) == 673
    primers = open(os.path.join(REF, "Core_primers_set", "core_final_maxprimers_set.fa")).read()
    golden = {
        "source": "multiPrime.py:441-457 (rule BWT_validation) as run by the reference's author: bowtie2 -N 1 -L 8 -a + samtools + V9",
        "flags": {"l": 18, "t": 1, "s": "50,2000", "m": 1},
        "primers_fa": primers,
        "rows": rows,
        "unmatched_in_cluster": un_cluster,
        "unmatched_all": sorted(unmatched),
        "pair_num": open(out_file + ".pair.num").read(),
        "total_acc_num": open(out_file + ".total.acc.num").read(),
    }
    with gzip.GzipFile(os.path.join(HERE, "bwt_cluster0.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(golden, sort_keys=True).encode())
    with gzip.GzipFile(os.path.join(HERE, "inputs", "bwt_unmatched.fa.gz"), "wb", mtime=0) as f:
        f.write(open(out_file + ".unmatched.fa", "rb").read())
    print("rows", sum(len(v) for v in rows.values()), "sequences with a product", len(rows), "unmatched in cluster", len(un_cluster),
          "unmatched records", len(unmatched))


if __name__ == "__main__":
    main()
