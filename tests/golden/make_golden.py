#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

This script is test infrastructure.  It imports the reference's live core
(`/root/reference/scripts/multiPrime-core_V20.py`, unmodified, by path) in this
container, instruments it by monkey-patching (no reference source is copied),
runs it on the fixtures listed in FIXTURES, and stores

  <name>.tsv                 the reference TSV, byte for byte
  <name>.noncov.json.gz      `.non_coverage_seq_id_json`, canonicalised
  <name>.gap.json.gz         `.gap_seq_id_json`, canonicalised
  <name>.trace.json.gz       per-window intermediates (cover / gap histograms in
                             first-seen order, freq + NN matrices, seeds, every
                             mis_primer_check call, every refine step)
  kat.json                   known-answer vectors of the reference's pure functions

The reference is run with the environment SURVEY.md Appendix A-14 prescribes
(NPY_DISABLE_CPU_FEATURES so numpy's 4-element argsort is the stable one the
author's numpy 1.21 had; PYTHONHASHSEED=0).  /root/reference does not exist on
the GPU box, which is why the vectors (and gz copies of the input alignments)
are committed.

Usage:  python tests/golden/make_golden.py [--only NAME ...] [--jobs 8]
"""
import argparse
import gzip
import importlib.util
import io
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
V20 = os.path.join(REF, "scripts", "multiPrime-core_V20.py")
ENV = {
    "NPY_DISABLE_CPU_FEATURES": "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3",
    "PYTHONHASHSEED": "0",
}

T = os.path.join(REF, "test_data")
YAML = dict(n=4, d=10, c="2,3,-1", g="0.2,0.7", s=150, l=18, e=3.6, f=0.7, a=4)      # multiPrime.yaml:74-96
CFG2 = dict(n=4, c="2,3,-1", g="0.2,0.7", s=150, e=3.6, f=0.8, a=4)                   # BASELINE config 2
DEF = dict(l=18, n=4, d=10, e=3.6, g="0.2,0.7", s=100, f=0.8, c="1,2,-1", a=4)        # V20:60-102 defaults

FIXTURES = {
    "ivc_v0": (f"{T}/variation_effect/IVC/IV_C.msa", dict(DEF, v=0)),
    "ivc_v1": (f"{T}/variation_effect/IVC/IV_C.msa", dict(DEF, v=1)),
    "ivc_v2": (f"{T}/variation_effect/IVC/IV_C.msa", dict(DEF, v=2)),
    "msa1000_k18_d64": (f"{T}/1000_fasta.msa", dict(CFG2, l=18, d=64, v=1)),
    "msa1000_k20_d64": (f"{T}/1000_fasta.msa", dict(CFG2, l=20, d=64, v=1)),
    "msa1000_k22_d64": (f"{T}/1000_fasta.msa", dict(CFG2, l=22, d=64, v=1)),
    "msa1000_k18_d10": (f"{T}/1000_fasta.msa", dict(CFG2, l=18, d=10, v=1)),
    "msa1000_k30_d64": (f"{T}/1000_fasta.msa", dict(CFG2, l=30, d=64, v=2)),
    "msa1000_k31_d64": (f"{T}/1000_fasta.msa", dict(CFG2, l=31, d=64, v=1)),
    # other corners of the flag space: one strict position from the 3' end, low coverage, few degenerate positions, tight entropy / GC
    "msa1000_c1_f06": (f"{T}/1000_fasta.msa", dict(CFG2, l=19, d=16, v=2, c="1,-1", f=0.6, n=2)),
    "ivc_e30_g": (f"{T}/variation_effect/IVC/IV_C.msa", dict(DEF, v=1, e=3.0, g="0.35,0.55", d=48, n=6, c="4")),
    "cluster0_v0_d64": (f"{T}/results/Clusters_msa/Cluster_0_20727.tmsa", dict(YAML, v=0, d=64, n=8, f=0.9, a=2)),
    "cluster0_v1": (f"{T}/results/Clusters_msa/Cluster_0_20727.tmsa", dict(YAML, v=1)),
    "cluster0_v2": (f"{T}/results/Clusters_msa/Cluster_0_20727.tmsa", dict(YAML, v=2)),
    "testfa": (f"{T}/test.fa", dict(DEF, v=1)),
    # synthetic inputs written by this script into inputs/ (see make_synthetic)
    "syn_iupac": ("@syn_iupac", dict(DEF, l=14, v=1, s=60, d=24)),
    "syn_v2": ("@syn_v2", dict(DEF, l=20, v=2, d=64, s=100, c="2,3,-1")),
    "syn_ragged": ("@syn_ragged", dict(DEF, l=16, v=1, s=80)),
    "syn_v3_k27": ("@syn_v2", dict(DEF, l=27, v=3, d=32, s=100, c="1,-2", n=6)),
    "syn_edge": ("@syn_edge", dict(DEF, l=15, v=1, s=50, d=16, f=0.7)),
    "syn_v2_k31": ("@syn_v2", dict(DEF, l=31, v=2, d=64, s=100, c="2,3,-1", n=6)),
    # primers longer than one 32-bit window word (round 4: -l up to 63)
    "msa1000_k36_d64": (f"{T}/1000_fasta.msa", dict(CFG2, l=36, d=64, v=1)),
    "ivc_k45_v2": (f"{T}/variation_effect/IVC/IV_C.msa", dict(DEF, l=45, v=2, d=32, n=6, c="2,3,-1")),
    "cluster0_k32": (f"{T}/results/Clusters_msa/Cluster_0_20727.tmsa", dict(YAML, l=32, v=1)),
    "syn_iupac_k33": ("@syn_iupac", dict(DEF, l=33, v=1, s=60, d=24, e=9.0, f=0.5)),
    "syn_ragged_k40": ("@syn_ragged", dict(DEF, l=40, v=2, s=80)),
    "syn_edge_k63": ("@syn_edge", dict(DEF, l=63, v=3, s=50, d=16, f=0.4, c="1,2,-1,-3", e=12.0)),
    "syn_v2_k50": ("@syn_v2", dict(DEF, l=50, v=2, d=64, s=100, c="2,3,-1", n=6, e=10.0, f=0.5)),
}


def make_synthetic(name):
    """Small seeded inputs that exercise IUPAC expansion, lower case / N, ragged rows."""
    import numpy as np
    sys.path.insert(0, REPO)
    from multiprime_amd.synth import synth_block, to_fasta
    path = os.path.join(HERE, "inputs", name[1:] + ".fa.gz")
    if name == "@syn_iupac":
        rows = synth_block(0, 60, 260, 11, p_iupac=0.01, p_gap=0.01, edge_frac=0.3, block_rows=64)
        rng = np.random.default_rng(5)
        codes = np.frombuffer(b"RYMKSWHBVDNn", dtype=np.uint8)
        hit = rng.random(rows.shape) < 0.004
        rows = np.where(hit, codes[rng.integers(0, len(codes), rows.shape)], rows).astype(np.uint8)
        low = rng.random(rows.shape) < 0.05
        rows = np.where(low & (rows >= 65) & (rows <= 90), rows + 32, rows).astype(np.uint8)
        data = to_fasta(rows)
    elif name == "@syn_v2":
        rows = synth_block(0, 200, 300, 12, p_gap=0.006, edge_frac=0.2, p_iupac=3e-4, block_rows=256)
        data = to_fasta(rows)
    elif name == "@syn_ragged":
        rows = synth_block(0, 14, 520, 13, p_gap=0.0, edge_frac=0.0, p_iupac=2e-3, block_rows=16)
        rng = np.random.default_rng(7)
        lens = [520, 520, 520, 520, 520, 520, 520, 520, 520, 520, 470, 455, 440, 300]
        parts = []
        for i in range(rows.shape[0]):
            s = rows[i, :lens[i]].tobytes()
            parts.append(b">r%02d some description\n" % i)
            for j in range(0, len(s), 70):          # multi-line records
                parts.append(s[j:j + 70] + b"\n")
        parts.insert(6, b"# a comment line the parser must skip\n")
        data = b"".join(parts)
    elif name == "@syn_edge":
        # record-level quirks of parse_seq: a repeated id concatenates (ragged, longer row), CRLF line ends,
        # blank lines, '#' comments, an all-gap row, '*' / '.' / digits, multi-line records of uneven width
        rows = synth_block(0, 40, 180, 14, p_gap=0.02, edge_frac=0.4, p_iupac=3e-3, block_rows=64)
        rows[5, :] = ord("-")
        rows[6, 20:60] = ord("N")
        rows[7, 30:33] = np.frombuffer(b"*.7", dtype=np.uint8)
        parts = []
        for i in range(rows.shape[0]):
            s = rows[i].tobytes()
            parts.append(b">e%02d desc\r\n" % i)
            w = 50 + 7 * (i % 4)
            for j in range(0, len(s), w):
                parts.append(s[j:j + w] + (b"\r\n" if i % 2 else b"\n"))
            if i == 9:
                parts.append(b"\n# comment\n\n")
        parts.append(b">e03 again\n" + rows[3, :25].tobytes() + b"\n")      # same id twice: appended to e03
        data = b"".join(parts)
    else:
        raise KeyError(name)
    with open(path, "wb") as f:
        f.write(gzip.compress(data, 9, mtime=0))
    return path


def load_v20():
    spec = importlib.util.spec_from_file_location("mpcore_v20", V20)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def canon_noncov(d):
    return {str(k): [{km: sorted(ids) for km, ids in sorted(side.items())} for side in v] for k, v in d.items()}


def canon_gap(d):
    return {str(k): {km: list(ids) for km, ids in sorted(v.items())} for k, v in d.items()}


def write_gz_json(path, obj):
    raw = json.dumps(obj, sort_keys=True, separators=(",", ":")).encode()
    with open(path, "wb") as f:
        f.write(gzip.compress(raw, 9, mtime=0))


def run_one(name):
    import numpy as np
    src, fl = FIXTURES[name]
    tmpdir = os.path.join("/tmp", "mp_golden", name)
    os.makedirs(tmpdir, exist_ok=True)
    if src.startswith("@"):
        gz = make_synthetic(src)
        inp = os.path.join(tmpdir, "input.fa")
        with open(inp, "wb") as f:
            f.write(gzip.decompress(open(gz, "rb").read()))
    else:
        inp = src
    out = os.path.join(tmpdir, "out.tsv")
    m = load_v20()
    C = m.NN_degenerate
    trace = {}
    cur = {"pos": None}

    def rec():
        return trace.setdefault(str(cur["pos"]), {"pos": int(cur["pos"]), "mis": [], "refine": []})

    o_get = C.get_primers
    def get_primers(self, sequence_dict, primer_start):
        cur["pos"] = primer_start
        return o_get(self, sequence_dict, primer_start)
    C.get_primers = get_primers

    o_ent = C.entropy
    def entropy(self, cover, cover_number, gap_sequence, gap_sequence_number):
        r = rec()
        r["cover"] = [[k, int(v)] for k, v in cover.items()]
        r["gap"] = [[k, int(v)] for k, v in gap_sequence.items()]
        r["cover_number"] = int(cover_number)
        r["gap_number"] = int(gap_sequence_number)
        cb, tb = o_ent(self, cover, cover_number, gap_sequence, gap_sequence_number)
        r["cBit"], r["tBit"] = cb, tb
        return cb, tb
    C.entropy = entropy

    o_state = C.state_matrix
    def state_matrix(self, primers_db):
        nodes = o_state(self, primers_db)
        r = rec()
        r["freq_rows"] = [str(x) for x in nodes.index.values.tolist()]
        r["freq"] = np.array(nodes).astype(int).tolist()
        return nodes
    C.state_matrix = state_matrix

    o_trans = C.trans_matrix
    def trans_matrix(self, primers):
        t = o_trans(self, primers)
        rec()["NN"] = np.array(t).astype(int).tolist()
        return t
    C.trans_matrix = trans_matrix

    o_vit = C.get_optimal_primer_by_viterbi
    def vit(self, nodes, trans):
        b = o_vit(self, nodes, trans)
        rec()["NM"] = [int(x) for x in b]
        return b
    C.get_optimal_primer_by_viterbi = vit

    o_mm = C.get_optimal_primer_by_MM
    def mm(self, cover_for_MM):
        b = o_mm(self, cover_for_MM)
        rec()["MM"] = [int(x) for x in b]
        return b
    C.get_optimal_primer_by_MM = mm

    o_mis = C.mis_primer_check
    def mis(self, all_primers, optimal_primer, cover, non_gap_seq_id):
        res = o_mis(self, all_primers, optimal_primer, cover, non_gap_seq_id)
        perfect = sum(cover[x] for x in self.degenerate_seq(optimal_primer) if x in cover)
        rec()["mis"].append([optimal_primer, int(res[0]), int(res[2]), int(perfect), int(sum(cover.values()))])
        return res
    C.mis_primer_check = mis

    o_ref = C.refine_by_NN_array
    def refine(self, optimal_primer_list, optimal_coverage_init, cover, optimal_NN_index, optimal_NN_coverage, NN_array):
        before = "".join(optimal_primer_list)
        res = o_ref(self, optimal_primer_list, optimal_coverage_init, cover, optimal_NN_index,
                    optimal_NN_coverage, NN_array)
        rec()["refine"].append([before, "".join(res[0]), int(res[1]), [int(x) for x in res[2]], int(res[4]), int(res[5])])
        return res
    C.refine_by_NN_array = refine

    o_dimer = C.dimer_check
    def dimer(self, primer):
        b = o_dimer(self, primer)
        rec()["self_dimer"] = [primer, bool(b)]
        return b
    C.dimer_check = dimer

    t0 = time.time()
    app = C(seq_file=inp, primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"],
            score_of_dege_bases=fl["d"], raw_entropy_threshold=fl["e"], product_len=fl["s"],
            position=fl["c"], variation=fl["v"], distance=fl["a"], GC=fl["g"], nproc=1, outfile=out)
    app.run()
    wall = time.time() - t0
    meta = {
        "name": name, "flags": fl, "input": os.path.basename(src) if not src.startswith("@") else src[1:] + ".fa",
        "n_seq": int(app.total_sequence_number), "start": int(app.start_position), "stop": int(app.stop_position),
        "entropy_threshold": app.entropy_threshold, "Y_strict": sorted(int(x) for x in app.Y_strict),
        "Y_strict_R": sorted(int(x) for x in app.Y_strict_R), "reference_wall_s": round(wall, 2),
        "n_windows": int(app.stop_position - app.start_position - fl["l"]),
        "n_mis_calls": sum(len(r["mis"]) for r in trace.values()),
        "evals": sum(c[4] for r in trace.values() for c in r["mis"]),
    }
    tsv = open(out, "rb").read()
    with open(os.path.join(HERE, name + ".tsv"), "wb") as f:
        f.write(tsv)
    for line in tsv.decode().splitlines()[1:]:
        cols = line.split("\t")
        trace[cols[0]]["row"] = cols
    write_gz_json(os.path.join(HERE, name + ".noncov.json.gz"),
                  canon_noncov(json.load(open(out + ".non_coverage_seq_id_json"))))
    write_gz_json(os.path.join(HERE, name + ".gap.json.gz"), canon_gap(json.load(open(out + ".gap_seq_id_json"))))
    write_gz_json(os.path.join(HERE, name + ".trace.json.gz"), {"meta": meta, "windows": trace})
    print(json.dumps(meta))


def make_kat():
    """Known-answer vectors of the reference's pure functions."""
    import random
    m = load_v20()
    C = m.NN_degenerate
    rnd = random.Random(20250303)
    app = object.__new__(C)
    app.distance = 4
    app.GC = ["0.2", "0.7"]
    kat = {}
    conc = ["".join(rnd.choice("ACGT") for _ in range(rnd.randint(12, 28))) for _ in range(60)]
    conc += ["ACGTACGTACGTACGTAC", "ACGTACGT", "AATT", "GAATTC", "GGGGCCCC", "ATATATATATATATAT", "CAGCAGCAGCAGCAGCAG"]
    kat["Calc_Tm_v2"] = [[s, m.Calc_Tm_v2(s)] for s in conc]
    kat["Calc_deltaH_deltaS"] = [[s, list(m.Calc_deltaH_deltaS(s))] for s in conc[:20] + conc[-7:]]
    kat["symmetry"] = [[s, bool(m.symmetry(s))] for s in conc]
    iu = "ACGTRYMKSWHBVDN"
    def dege(n, p=0.15):
        return "".join(rnd.choice(iu[4:14]) if rnd.random() < p else rnd.choice("ACGT") for _ in range(n))
    deg = [dege(rnd.randint(12, 24)) for _ in range(80)]
    deg += ["AAAAGCTGCTGCATGCAT", "ACACACACGTTGCAGTCA", "ACGACGACGTTGCAGTCA", "GCGCGCGCATATATATGC", "RRTCAGATGCACCYATTG",
            "CCCAKRTCYTCAGCATTT", "TGCATGCAGTCNACGTTA", "GGGGGGAAAACCCCCCTT"]
    kat["degenerate_seq"] = [[s, C.degenerate_seq(s)] for s in deg[:30] + ["AC-GR", "--RY-", "N"]]
    kat["score_trans"] = [[s, int(m.score_trans(s))] for s in deg]
    kat["dege_number"] = [[s, int(m.dege_number(s))] for s in deg]
    kat["RC"] = [[s, m.RC(s)] for s in deg[:20]]
    kat["GC_fraction"] = [[s, app.GC_fraction(s)] for s in deg]
    kat["di_nucleotide"] = [[s, bool(app.di_nucleotide(s))] for s in deg]
    kat["hairpin_check"] = [[s, bool(app.hairpin_check(s))] for s in deg + conc[:30]]
    kat["deltaG"] = [[s, app.deltaG(s)] for s in [d[-rnd.randint(5, 12):] for d in deg] + ["GCTA", "AATT", "GAATTC", "CCGGTA"]]
    kat["dimer_check"] = [[s, bool(app.dimer_check(s))] for s in deg + conc[:40]]
    kat["primer_pre_filter"] = [[s, app.primer_pre_filter(s)] for s in deg + conc[:20]]
    kat["Penalty_points"] = [[l, g, d1, d2, m.Penalty_points(l, g, d1, d2)]
                             for l in (5, 8, 12, 18) for g in (0, 3, 5) for d1 in (0,) for d2 in (0, 1, 2, 7)]
    yd = []
    for _ in range(200):
        k = rnd.randint(10, 24)
        p = dege(k, 0.3)
        q = "".join(rnd.choice("ACGT-") if rnd.random() < 0.4 else rnd.choice(C.degenerate_seq(ch)) for ch in p)
        yd.append([p, q, [int(x) for x in m.Y_distance(p, q)]])
    kat["Y_distance"] = yd
    gy = []
    for k in (12, 18, 20, 22, 27):
        for c in ("1,2,-1", "2,3,-1", "1,-1", "2,-1", "-2", "0", "18", "3,-3,5"):
            a = object.__new__(C)
            a.position, a.primer_length = c, k
            f, r = a.get_Y()
            gy.append([k, c, sorted(int(x) for x in f), sorted(int(x) for x in r)])
    kat["get_Y"] = gy
    kat["score_table"] = {k: v for k, v in m.score_table.items()}
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=0, sort_keys=True)
    print("kat.json written:", {k: len(v) for k, v in kat.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--run-one")
    ap.add_argument("--kat", action="store_true")
    a = ap.parse_args()
    if any(os.environ.get(k) != v for k, v in ENV.items()):
        os.execve(sys.executable, [sys.executable] + sys.argv, dict(os.environ, **ENV))
    if a.run_one:
        run_one(a.run_one)
        return
    if a.kat:
        make_kat()
        return
    names = a.only or list(FIXTURES)
    procs, pending = [], list(names) + ["--kat"]
    logs = {}
    while pending or procs:
        while pending and len(procs) < a.jobs:
            n = pending.pop(0)
            cmd = [sys.executable, os.path.abspath(__file__)] + (["--kat"] if n == "--kat" else ["--run-one", n])
            procs.append((n, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for n, p in list(procs):
            if p.poll() is not None:
                logs[n] = p.stdout.read()
                print(f"[{n}] exit {p.returncode}\n{logs[n][-2000:]}", flush=True)
                procs.remove((n, p))
        time.sleep(0.5)


if __name__ == "__main__":
    main()
