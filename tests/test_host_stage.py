"""The native host stage (include/mprime_host.h: csrc/fasta.cpp, csrc/hostplan.cpp) against its checkers:
  * the FASTA record parser against the pure-Python restatement of parse_seq's record logic (oracle/core_ref.py),
    on fuzzed files (CRLF, lone CR, comments, repeated ids, blank lines, no final newline), whole and cut into chunks;
  * the insertion-ordered cover / gap dictionaries against a row-by-row replay of V20:689-711;
  * seeds, refinement chains and entropies against the reference's recorded internals (tests/golden traces);
  * the whole drop-in (native stage) against the round-1 pure-Python host logic on random alignments — TSV bytes
    and both JSON files.
All CPU: the C ABI of mprime.h is served by the oracle library here; the host stage itself is the product's code
(pure host C++ inside libmprime_hip.so, it needs no GPU).
"""
import itertools
import json
import os

import numpy as np
import pytest

from conftest import golden_input, load_gz_json
from multiprime_amd import host, iupac
from multiprime_amd.core import NN_degenerate
from multiprime_amd.synth import synth_block, to_fasta
from oracle import core_ref


# ------------------------------------------------------------------------------------------------ FASTA records
def _fuzz_fasta(rng):
    nl = [b"\n", b"\r\n", b"\r"][int(rng.integers(0, 3))] if rng.random() < 0.5 else None
    out = []
    n_ids = int(rng.integers(1, 12))
    ids = [b">s%d" % i for i in range(n_ids)]
    for _ in range(int(rng.integers(1, 60))):
        t = nl or [b"\n", b"\r\n", b"\r"][int(rng.integers(0, 3))]
        r = rng.random()
        if r < 0.25 or not out:
            tail = [b"", b" desc text", b"\tx", b"  ", b" a b"][int(rng.integers(0, 5))]
            out.append(ids[int(rng.integers(0, n_ids))] + tail + t)
        elif r < 0.32:
            out.append(b"#comment >x" + t)
        elif r < 0.38:
            out.append(t)                                             # blank line
        else:
            body = bytes(rng.choice(np.frombuffer(b"ACGTacgtNRY-.*x", np.uint8), size=int(rng.integers(0, 40))))
            pad = [b"", b" ", b"\t", b"  ", b"\x1c", b"\x1f ", b"\x0b"][int(rng.integers(0, 7))]     # str.strip() takes 0x1c-0x1f too
            out.append(pad + body + pad + t)
    raw = b"".join(out)
    if rng.random() < 0.3:
        raw = raw.rstrip(b"\r\n")                                     # no terminator on the last line
    return raw


@pytest.mark.parametrize("threads", [1, 2, 5])
def test_fasta_parser_matches_python_restatement(threads, monkeypatch):
    monkeypatch.setenv("MP_HOST_THREADS", str(threads))               # exact: even tiny inputs are cut into chunks
    rng = np.random.default_rng(threads)
    for _ in range(300):
        raw = _fuzz_fasta(rng)
        want_ids, want_data, want_off = core_ref.parse_records(raw)
        fa = host.Fasta(raw=raw)
        data, off = fa.rows()
        assert fa.ids == want_ids
        assert off.tolist() == want_off.tolist()
        assert data.tobytes() == want_data.tobytes()


def test_fasta_parser_edge_cases(tmp_path):
    with pytest.raises(ValueError):
        host.Fasta(raw=b"ACGT\n>a\nAC\n")                            # data before the first header (V20: NameError)
    with pytest.raises(ValueError):
        host.Fasta(raw=b"\n>a\nAC\n")                                # even a blank line
    fa = host.Fasta(raw=b"")
    assert fa.n_rows == 0 and fa.ids == []
    fa = host.Fasta(raw=b">a x\nAC\n>b\n>a\nGT\n>c\n\n")              # repeated id concatenates; >b never receives a line
    assert fa.ids == [">a", ">c"] and fa.rows()[0].tobytes() == b"ACGT" and fa.rows()[1].tolist() == [0, 4, 4]
    fa = host.Fasta(raw=">séq\nAC\n".encode("utf-8"))           # non-ASCII id: decoded as the reference's text mode does
    assert fa.ids == [">séq"]
    p = tmp_path / "x.fa"
    p.write_bytes(b">a\nACGT\nAC\n#c\n>b\nTTTT")
    fa = host.Fasta(str(p))
    assert fa.ids == [">a", ">b"] and fa.rows()[0].tobytes() == b"ACGTACTTTT"
    with pytest.raises(OSError):
        host.Fasta(str(tmp_path / "missing.fa"))


def test_white_space_at_line_ends_is_stripped_like_the_reference():
    """tests/golden/parser_ws.json: 45 small files run through the UNMODIFIED reference's parse_seq (make_golden_parser.py) — lines that
    start / end with every kind of white space str.strip() removes (ASCII, 0x1c-0x1f, U+0085, U+00A0, U+1680, U+2000-200A, U+2028/9, U+202F,
    U+205F, U+3000), white space inside id tokens, white-space-only lines, '#' behind white space.  The native parser and the checker's
    restatement give the reference's ids and (after V20:453's per-character mapping) its sequences; where the reference dies (data in front
    of the first header) both refuse the file."""
    import json
    import re
    from conftest import GOLDEN
    from oracle.core_ref import parse_records
    recs = json.load(open(os.path.join(GOLDEN, "parser_ws.json")))
    assert len(recs) == 45 and sum("error" in r for r in recs) == 20
    for rec in recs:
        raw = rec["file_latin1"].encode("latin-1")
        if "error" in rec:
            with pytest.raises(ValueError):
                host.Fasta(raw=raw)
            with pytest.raises(ValueError):
                parse_records(raw)
            continue
        fa = host.Fasta(raw=raw)
        data, off = fa.rows()
        got = [re.sub(b"[^ACGTRYMKSWHBVD]", b"-", data[off[i]:off[i + 1]].tobytes().upper()).decode() for i in range(fa.n_rows)]
        assert fa.ids == rec["ids"] and got == rec["seqs"], rec["file_latin1"]
        ids, d2, o2 = parse_records(raw)
        got2 = [re.sub(b"[^ACGTRYMKSWHBVD]", b"-", d2[o2[i]:o2[i + 1]].tobytes().upper()).decode() for i in range(len(ids))]
        assert ids == rec["ids"] and got2 == rec["seqs"], rec["file_latin1"]


def test_lone_carriage_return_files_parse_in_linear_time():
    """Classic-Mac line ends: the terminator search must not run to the end of the file for every line."""
    import time
    raw = b"".join(b">s%d\r" % i + b"ACGTACGTAC" * 6 + b"\r" for i in range(150000))       # 11 MB, no \n anywhere
    t0 = time.time()
    fa = host.Fasta(raw=raw, n_threads=1)
    assert fa.n_rows == 150000 and fa.rows()[1][-1] == 150000 * 60
    assert time.time() - t0 < 5.0


def test_fasta_parser_large_file_threads(tmp_path):
    rows = synth_block(0, 3000, 700, 7)
    p = tmp_path / "big.fa"
    p.write_bytes(to_fasta(rows))                                     # ~2 MB: several reader / scanner threads
    fa = host.Fasta(str(p), n_threads=4)
    want_ids, want_data, want_off = core_ref.parse_records(p.read_bytes())
    data, off = fa.rows()
    assert fa.ids == want_ids and off.tolist() == want_off.tolist() and data.tobytes() == want_data.tobytes()


# ------------------------------------------------------------------------------------------------ ordered tables
def _replay(rows):
    d = {}
    for _, keys in sorted(rows.items()):
        for key in keys:
            d[key] = d.get(key, 0) + 1
    return d


@pytest.mark.parametrize("seed", range(30))
def test_tables_equal_row_by_row_replay(seed):
    """The dictionaries the reference builds by walking the sequences in order (plain rows add their k-mer, IUPAC
    rows add their expansions, V20:689-711) = what the planning stage derives from the device histogram
    (k-mer, count, first row — here in shuffled order and split over two 'shards') plus the exception list."""
    rng = np.random.default_rng(seed)
    k, v = 6, 1
    n_rows = int(rng.integers(1, 300))
    pool = ["".join(rng.choice(list("ACGT-"), p=[.23, .23, .23, .23, .08], size=k)) for _ in range(int(rng.integers(1, 25)))]
    cover_rows, gap_rows, exc, plain = {}, {}, [], {}
    for r in range(n_rows):
        if rng.random() < 0.2:                                         # a row holding IUPAC codes
            s = list(pool[int(rng.integers(0, len(pool)))])
            for _ in range(int(rng.integers(1, 3))):
                s[int(rng.integers(0, k))] = "RYMKSWHBVD"[int(rng.integers(0, 10))]
            s = "".join(s)
            exc.append((r, s))
            if s.count("-") > v:
                gap_rows[r] = [s]
            else:
                cover_rows[r] = iupac.expand(s)
        else:
            s = pool[int(rng.integers(0, len(pool)))]
            plain[r] = s
            (gap_rows if s.count("-") > v else cover_rows)[r] = [s]
    want_cover, want_gap = _replay(cover_rows), _replay(gap_rows)
    # device view: distinct plain k-mers per shard with count and first row, shuffled
    cut = n_rows // 2
    ents = []
    for lo, hi in ((0, cut), (cut, n_rows)):
        cnt, first = {}, {}
        for r in range(lo, hi):
            if r in plain:
                cnt[plain[r]] = cnt.get(plain[r], 0) + 1
                first.setdefault(plain[r], r)
        ents += [(s, cnt[s], first[s]) for s in cnt]
    rng.shuffle(ents)
    chars = np.frombuffer("".join(e[0] for e in ents).encode(), np.uint8).reshape(len(ents), k) if ents else np.zeros((0, k), np.uint8)
    words = iupac.words_of_kmers(chars).T if len(ents) else np.zeros((3, 0), np.uint32)
    xs = [e for e in exc]
    rng.shuffle(xs)
    freq = np.ones((1, 4, k), np.int64)
    nn = np.ones((1, k - 1, 4, 4), np.int64)
    plan = host.Plan(k=k, v=v, n_windows=1, total_sequences=n_rows, coverage=0.0, entropy_threshold=100.0, max_degeneracy=8,
                     max_dege_positions=3, e_window=np.zeros(len(ents), np.int32), e_words=words,
                     e_count=[e[1] for e in ents], e_first=[e[2] for e in ents], x_window=np.zeros(len(xs), np.int32),
                     x_row=[x[0] for x in xs], x_codes=iupac.MASK_LUT[np.frombuffer("".join(x[1] for x in xs).encode(), np.uint8)].reshape(len(xs), k),
                     freq=freq, nn=nn, keep_tables=True)
    for which, want in ((0, want_cover), (1, want_gap)):
        codes, counts, _ = plan.window_table(0, which)
        got = list(zip(iupac.strings_of(iupac.SYMBOL_LUT[codes]), counts.tolist()))
        assert got == list(want.items())
    _, cn, gn, _, _ = plan.windows()
    assert cn[0] == len(cover_rows) and gn[0] == len(gap_rows)


@pytest.mark.parametrize("seed", range(8))
def test_segment_form_of_plan_create_equals_the_general_form(seed):
    """mp_plan_create_segments (one rank's read-back as it stands: window offsets, 32-bit counts, LOCAL first rows + row_base) builds the
    plan mp_plan_create builds from the same entries given with a window per entry, 64-bit counts and global first rows — in window
    order and shuffled (the counting-sort path)."""
    rng = np.random.default_rng(100 + seed)
    k, v, W = 8, 1, int(rng.integers(1, 12))
    row_base = int(rng.integers(0, 5000))
    per = rng.integers(0, 30, size=W)
    off = np.zeros(W + 1, np.int64)
    np.cumsum(per, out=off[1:])
    n = int(off[-1])
    root = rng.integers(0, 4, size=(W, k))
    chars = np.empty((n, k), np.uint8)
    for w in range(W):
        seen = set()
        for i in range(off[w], off[w + 1]):
            while True:                                                    # distinct k-mers inside a window, close to its root
                s = root[w].copy()
                m = rng.random(k) < 0.25
                s[m] = rng.integers(0, 5, size=int(m.sum()))
                if tuple(s) not in seen:
                    seen.add(tuple(s))
                    break
            chars[i] = np.frombuffer(b"ACGT-", np.uint8)[s]
    words = iupac.words_of_kmers(chars).T.copy() if n else np.zeros((3, 0), np.uint32)
    count = rng.integers(1, 40, size=n).astype(np.int32)
    first = rng.permutation(n).astype(np.int32)                      # distinct first rows: ties would leave the insertion order open
    e_window = np.repeat(np.arange(W, dtype=np.int32), per)
    freq = rng.integers(1, 50, size=(W, 4, k)).astype(np.int64)
    nn = rng.integers(1, 50, size=(W, k - 1, 4, 4)).astype(np.int64)
    common = dict(k=k, v=v, n_windows=W, total_sequences=max(n, 1) + row_base, coverage=0.1, entropy_threshold=100.0, max_degeneracy=16,
                  max_dege_positions=3, x_window=np.zeros(0, np.int32), x_row=np.zeros(0, np.int64), x_codes=np.zeros((0, k), np.uint8),
                  freq=freq, nn=nn, keep_tables=True)
    seg = host.Plan(e_off=off, e_words=words, e_count=count, e_first=first, row_base=row_base, **common)
    order = rng.permutation(n)
    plans = [host.Plan(e_window=e_window, e_words=words, e_count=count.astype(np.int64), e_first=first.astype(np.int64) + row_base, **common),
             host.Plan(e_window=e_window[order], e_words=np.ascontiguousarray(words[:, order]), e_count=count[order].astype(np.int64),
                       e_first=first[order].astype(np.int64) + row_base, **common)]
    for gen in plans:
        for a, b in zip(seg.windows(), gen.windows()):
            assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
        assert (seg.n_planned, seg.n_candidates) == (gen.n_planned, gen.n_candidates)
        for a, b in zip(seg.candidates(), gen.candidates()):
            assert np.array_equal(a, b)
        for w in range(W):
            for which in (0, 1):
                for a, b in zip(seg.window_table(w, which), gen.window_table(w, which)):
                    assert np.array_equal(a, b)


def test_expand_kmers_is_itertools_product_order():
    rng = np.random.default_rng(3)
    syms = "ACGTRYMKSWHBVDN-"
    kmers = ["".join(rng.choice(list(syms), size=7)) for _ in range(50)]
    codes = iupac.MASK_LUT[np.frombuffer("".join(kmers).encode(), np.uint8)].reshape(len(kmers), 7)
    exp, src = host.expand_kmers(codes)
    got = iupac.strings_of(iupac.SYMBOL_LUT[exp])
    want = list(itertools.chain.from_iterable(iupac.expand(s) for s in kmers))
    assert got == want
    assert src.tolist() == [i for i, s in enumerate(kmers) for _ in iupac.expand(s)]


def test_plan_refuses_an_exponential_iupac_window():
    k = 18
    codes = np.full((1, k), 15, np.uint8)                              # NNN...: 4^18 expansions
    with pytest.raises(Exception) as e:
        host.Plan(k=k, v=1, n_windows=1, total_sequences=1, coverage=0.8, entropy_threshold=3.6, max_degeneracy=10,
                  max_dege_positions=4, e_window=[], e_words=np.zeros((3, 0), np.uint32), e_count=[], e_first=[],
                  x_window=[0], x_row=[0], x_codes=codes, freq=np.ones((1, 4, k), np.int64), nn=np.ones((1, k - 1, 4, 4), np.int64))
    assert "expansions" in str(e.value)


# ------------------------------------------------------------------------------------------------ reference internals
TRACED = ["syn_iupac", "syn_v2", "syn_ragged", "syn_v3_k27", "syn_edge", "ivc_v1", "msa1000_k18_d64", "msa1000_k22_d64", "cluster0_v2", "msa1000_k30_d64",
          "msa1000_k31_d64", "msa1000_c1_f06", "ivc_e30_g", "cluster0_v0_d64"]


@pytest.mark.parametrize("name", TRACED)
def test_seeds_chains_entropies_match_reference_trace(name, oracle_lib, tmp_path):
    tr = load_gz_json(name + ".trace.json.gz")
    fl = tr["meta"]["flags"]
    inp = tmp_path / (name + ".fa")
    inp.write_bytes(golden_input(tr["meta"]["input"]))
    app = NN_degenerate(seq_file=str(inp), primer_length=fl["l"], coverage=fl["f"], number_of_dege_bases=fl["n"],
                        score_of_dege_bases=fl["d"], raw_entropy_threshold=fl["e"], product_len=fl["s"], position=fl["c"],
                        variation=fl["v"], distance=fl["a"], GC=fl["g"], nproc=1, outfile=str(tmp_path / "o"), library=oracle_lib,
                        write_json=False)
    plan = app._plan()
    status, _, _, cbit, tbit = plan.windows()
    p0 = int(app.start_position)
    sym = iupac.SYMBOL_LUT
    n_refine = n_seed = n_bits = 0
    for pos, rec in tr["windows"].items():
        w = int(pos) - p0
        if "cBit" in rec:                                              # entropy (V20:602-614) as the reference rounded it
            assert (cbit[w], tbit[w]) == (rec["cBit"], rec["tBit"]), f"entropy at {pos}"
            n_bits += 1
        if "NM" not in rec:
            continue
        assert status[w] == 0
        nm, mm, _, _ = plan.seeds(w)
        assert nm.tolist() == rec["NM"]
        if rec.get("MM") is not None and rec["MM"] != rec["NM"]:
            assert mm is not None and mm.tolist() == rec["MM"], f"most-frequent seed at {pos}"
        n_seed += 1
        chains = []
        for s in range(2 if mm is not None else 1):
            codes, cov, _ = plan.chain(w, s)
            chains.append((iupac.strings_of(sym[codes]), cov.tolist()))
        for before, after, cov_after, *_ in rec["refine"]:             # every refine_by_NN_array call the reference made
            ok = any(a == before and b == after and c == cov_after
                     for strs, covs in chains for a, b, c in zip(strs, strs[1:], covs[1:]))
            assert ok, f"refinement step {before} -> {after} ({cov_after}) at {pos} is not on a native chain"
            n_refine += 1
    assert n_seed > 0 and n_bits > 0
    assert n_refine == sum(len(r["refine"]) for r in tr["windows"].values())


# ------------------------------------------------------------------------------------------------ native == python
def _run(cls, lib, inp, out, k, v, d, f):
    app = cls(seq_file=str(inp), primer_length=k, coverage=f, number_of_dege_bases=4, score_of_dege_bases=d,
              raw_entropy_threshold=3.6, product_len=60, position="2,3,-1", variation=v, distance=4, GC="0.2,0.7", nproc=1,
              outfile=str(out), library=lib)
    app.run()
    return app


@pytest.mark.parametrize("seed", range(6))
def test_native_stage_equals_python_restatement_on_random_alignments(seed, oracle_lib, tmp_path, monkeypatch):
    if seed % 2:
        monkeypatch.setenv("MP_JSON_BATCH", str(1 + seed))          # the native side files in runs of 2, 4 and 6 output windows
    rng = np.random.default_rng(100 + seed)
    n, L = int(rng.integers(40, 220)), int(rng.integers(140, 260))
    rows = synth_block(0, n, L, 500 + seed, p_gap=float(rng.choice([0.002, 0.02])), edge_frac=float(rng.choice([0.1, 0.4])),
                       p_iupac=float(rng.choice([0.0, 5e-4, 3e-3])), block_rows=256)
    inp = tmp_path / "in.fa"
    inp.write_bytes(to_fasta(rows))
    k, v = int(rng.choice([12, 18, 22])), int(rng.integers(0, 3))
    d, f = int(rng.choice([4, 10, 64])), float(rng.choice([0.6, 0.8]))
    _run(NN_degenerate, oracle_lib, inp, tmp_path / "native.out", k, v, d, f)
    _run(core_ref.NN_degenerate, oracle_lib, inp, tmp_path / "python.out", k, v, d, f)
    for suffix in ("", ".non_coverage_seq_id_json", ".gap_seq_id_json"):
        a = (tmp_path / ("native.out" + suffix)).read_bytes()
        b = (tmp_path / ("python.out" + suffix)).read_bytes()
        assert a == b, f"native host stage differs from the Python restatement in out{suffix}"
    assert len((tmp_path / "native.out").read_bytes().splitlines()) >= 1


def test_newline_count_equals_text_mode_read(tmp_path, monkeypatch):
    """mp_file_count_newlines against open(path).read().count("\\n") — \\n, \\r\\n, lone \\r, pairs split across thread chunks
    and 4 MB blocks, a trailing \\r, an empty file."""
    import random
    rng = random.Random(3)
    cases = [b"", b"\r", b"\n", b"\r\n", b"a\rb\r\nc\n\r", b"\r\r\n\n\r"]
    cases.append(b"".join(rng.choice([b"ACGT", b"\n", b"\r\n", b"\r", b">id x"]) for _ in range(20000)))
    big = bytearray(b"A" * ((4 << 20) + 10))
    big[(4 << 20) - 1:(4 << 20) + 1] = b"\r\n"                       # \r\n across a block boundary
    big[100:101] = b"\r"
    cases.append(bytes(big))
    for threads in ("1", "3", "7"):
        monkeypatch.setenv("MP_HOST_THREADS", threads)
        for i, raw in enumerate(cases):
            p = tmp_path / f"c{i}.txt"
            p.write_bytes(raw)
            with open(p, encoding="latin-1") as f:
                want = f.read().count("\n")
            assert host.count_newlines(str(p)) == want, (threads, i)


def test_expansions_as_window_words_equal_the_code_path():
    rng = np.random.default_rng(8)
    codes = np.where(rng.random((400, 18)) < 0.85, rng.choice(np.array([1, 2, 4, 8], np.uint8), size=(400, 18)),
                     rng.integers(0, 16, size=(400, 18))).astype(np.uint8)
    exp, src = host.expand_kmers(codes)
    words, src2 = host.expand_kmer_words(codes)
    assert np.array_equal(src, src2) and np.array_equal(iupac.words_of_codes(exp), words)


@pytest.mark.parametrize("k,v", [(18, 1), (18, 0), (25, 3), (40, 2)])
def test_exception_words_equal_selection_plus_expansion(k, v):
    """mp_expand_exception_words (rows with more than v gaps dropped, every expansion with its row's window) against the numpy selection
    around mp_expand_kmer_words that core.py used; a list without a gap; the capacity retry of the wrapper."""
    rng = np.random.default_rng(k + v)
    n = 3000
    codes = np.where(rng.random((n, k)) < 0.9, rng.choice(np.array([1, 2, 4, 8], np.uint8), size=(n, k)), rng.integers(0, 16, size=(n, k))).astype(np.uint8)
    codes[::7, : v + 1] = 0                                     # rows with more than v gaps
    x_win = np.sort(rng.integers(0, 200, size=n)).astype(np.int32)
    for cds in (codes, np.where(codes == 0, 1, codes).astype(np.uint8)):
        sel = (cds == 0).sum(axis=1) <= v
        words, src = host.expand_kmer_words(cds[sel])
        want_win = x_win[sel][src]
        got_win, got_words = host.expand_exception_words(x_win, cds, v)
        assert got_words.dtype == words.dtype and np.array_equal(got_words, words) and np.array_equal(got_win, want_win)
        assert len(got_win) > 0 and (0 < sel.sum() < n or not (cds == 0).any())
    many = np.full((50, k), 15, np.uint8)                       # 4^k expansions each: beyond any capacity
    many[:, 6:] = 1
    win, words = host.expand_exception_words(np.zeros(50, np.int32), many, v)      # 50 x 4096: the first guess (4 n + 1024) is too small
    assert len(win) == 50 * 4096 and len(words) == len(win)
    w0, e0 = host.expand_exception_words(np.zeros(0, np.int32), np.zeros((0, k), np.uint8), v)
    assert len(w0) == 0 and len(e0) == 0


def test_fasta_gather_equals_the_rows_array():
    """mp_fasta_gather (the streamed load's source): any byte range of the rows laid end to end, on any number of threads — multi-line
    records, CRLF, repeated ids, empty records."""
    import ctypes as C
    import numpy as np
    from multiprime_amd import host
    rng = np.random.default_rng(5)
    recs = []
    for i in range(300):
        L = int(rng.integers(0, 400))
        body = bytes(rng.choice(np.frombuffer(b"ACGT-N", np.uint8), size=L))
        width = int(rng.integers(20, 80))
        lines = [body[a:a + width] for a in range(0, L, width)] or [b""]
        recs.append(b">s%03d\n" % (i % 280) + (b"\r\n" if i % 3 == 0 else b"\n").join(lines) + b"\n")
    fa = host.Fasta(raw=b"".join(recs))
    data, off = fa.rows()
    assert np.array_equal(fa.row_offsets(), off)
    total = int(off[-1])
    for a, b, T in [(0, total, 0), (0, total, 1), (5, total - 7, 3), (int(off[17]) + 3, int(off[18]) - 1, 2), (total // 3, total // 3 + 1, 1),
                    (int(off[40]), int(off[95]), 5), (total, total, 1)]:
        out = np.full(max(b - a, 1), 255, np.uint8)
        rc = host.dll().mp_fasta_gather(fa.h, a, b, out.ctypes.data_as(C.c_void_p), T)
        assert rc == 0 and np.array_equal(out[: b - a], data[a:b]), (a, b, T)


def test_parallel_join_equals_serial_join(monkeypatch):
    """The parser's join on all threads (no id repeated: the common file) against the serial join (MP_HOST_SERIAL_JOIN), and its
    fall-backs: a repeated id (first-appearance order, the record continues), headers without lines, blank lines, comments, CRLF / lone
    CR, a record longer than a chunk, sequence data before the first header (the serial join's error)."""
    import numpy as np
    from multiprime_amd import host
    rng = np.random.default_rng(11)

    def make(n, repeat, long_record=False):
        out = []
        for i in range(n):
            ident = b"s%04d" % (i if not (repeat and i % 17 == 5) else i - 3)
            lines = [bytes(rng.choice(np.frombuffer(b"ACGT-", np.uint8), size=int(rng.integers(0, 90)))) for _ in range(int(rng.integers(0, 4)))]
            if long_record and i == n // 2:
                lines = [b"ACGT" * 20] * 400
            if i % 23 == 0:
                lines.insert(0, b"# a comment")
            if i % 29 == 0:
                lines.append(b"")
            nl = (b"\n", b"\r\n", b"\r")[i % 3]
            out.append(b">" + ident + b" description" + nl + b"".join(x + nl for x in lines))
        return b"".join(out)

    def parsed(raw):
        fa = host.Fasta(raw=raw)
        data, off = fa.rows()
        res = (fa.ids, off.tolist(), data.tobytes())
        fa.close()
        return res

    for threads in ("2", "7", "13"):
        for repeat, long_record in ((False, False), (True, False), (False, True)):
            raw = make(500, repeat, long_record)
            monkeypatch.setenv("MP_HOST_THREADS", threads)
            monkeypatch.delenv("MP_HOST_SERIAL_JOIN", raising=False)
            fast = parsed(raw)
            monkeypatch.setenv("MP_HOST_SERIAL_JOIN", "1")
            assert parsed(raw) == fast, (threads, repeat, long_record)
    monkeypatch.delenv("MP_HOST_SERIAL_JOIN", raising=False)
    monkeypatch.setenv("MP_HOST_THREADS", "5")
    import pytest
    with pytest.raises(ValueError, match="before the first"):
        host.Fasta(raw=b"ACGT\n" + make(50, False))
