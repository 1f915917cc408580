"""TEST INFRASTRUCTURE — the pure-Python restatement of the per-window host logic that round 1 shipped as
multiprime_amd/core.py, kept as the checker of the native host stage (multiprime_amd/csrc/hostplan.cpp, fasta.cpp,
include/mprime_host.h).  It is pinned to multiPrime-core_V20.py's recorded internals (tests/golden traces) and to its
output files; tests run it beside the product on the same inputs.  Nothing under multiprime_amd/ imports it.

Original header:
Host side of the MI355X-native core step: same class API, CLI, TSV and JSON files as
`scripts/multiPrime-core.py` (V20 = multiPrime-core_V20.py), with every O(N_sequences) loop
moved behind the C ABI of include/mprime.h (hand-written HIP kernels, csrc/).

What runs where
  device  per-character mapping and packing (V20:453), row attributes (V20:625-627), the
          k-mer of every (window, sequence) with edge-gap repair (V20:666-687), the per-window
          k-mer histograms (V20:689-711) and every candidate x sequence mismatch evaluation
          (V20:1103-1130, 229-233);
  host    the O(1)-per-window control flow: gates, entropy, frequency / nearest-neighbour
          matrices from the histogram, Viterbi and most-frequent seeds, the greedy degeneracy
          refinement, Tm and the string filters.

One design change matters for the GPU: the reference interleaves refinement steps with
coverage evaluations (one mis_primer_check per step, V20:883-906).  The refinement sequence
itself depends only on the nearest-neighbour counts, not on the evaluations — those only
decide where to stop — so the host first derives the whole refinement chain of every window
and seed, evaluates ALL chain members of ALL windows in ONE batched launch, and then replays
the reference's stopping rules on the results.  Outputs are identical; launches per alignment
drop from ~12 dependent rounds to one.
"""
from __future__ import annotations

import json
import math
import sys
import time
from collections import defaultdict
from itertools import compress, islice
from statistics import mean

import numpy as np

from oracle import iupac_ref as iupac, msa_ref as msa, thermo_ref as thermo
from oracle import filters_ref as filters
from multiprime_amd._abi import Library

_B2I = {"A": 0, "C": 1, "G": 2, "T": 3}
_DROP_ACGT = {ord(c): None for c in "ACGT"}
_IDX_LUT = np.full(256, 4, np.int64)
for _c, _i in _B2I.items():
    _IDX_LUT[ord(_c)] = _i

HEADERS = ["Position", "Entropy of cover (bit)", "Entropy of total (bit)", "Optimal_primer",
           "primer_degenerate_number", "nonsense_primer_number", "Optimal_coverage", "Mis-F-coverage",
           "Mis-R-coverage", "Tm", "Information"]


from json.encoder import encode_basestring_ascii as _q


def _dump_side_file(obj, fh, depth_list):
    """json.dump(obj, fh, indent=4) for {pos: {kmer: [ids]}} (depth_list False) or {pos: [{kmer: [ids]}, {...}]}
    (True), byte for byte, without the pure-Python encoder that `indent` forces (it was 25 % of the run)."""
    def ids_block(ids, ind):
        if not ids:
            return "[]"
        pad = " " * (ind + 4)
        return "[\n" + ",\n".join(pad + _q(x) for x in ids) + "\n" + " " * ind + "]"

    def kmer_block(d, ind):
        if not d:
            return "{}"
        pad = " " * (ind + 4)
        return "{\n" + ",\n".join(pad + _q(k) + ": " + ids_block(v, ind + 4) for k, v in d.items()) + "\n" + " " * ind + "}"

    if not obj:
        fh.write("{}")
        return
    parts = []
    for pos, val in obj.items():
        if depth_list:
            body = "[\n" + ",\n".join(" " * 8 + kmer_block(d, 8) for d in val) + "\n" + " " * 4 + "]" if val else "[]"
        else:
            body = kmer_block(val, 4)
        parts.append(" " * 4 + _q(str(pos)) + ": " + body)
    fh.write("{\n" + ",\n".join(parts) + "\n}")


def _desc_stable(values):
    """np.argsort(x)[::-1] with the stable tie order of the author's numpy (SURVEY A-14)."""
    return sorted(range(len(values)), key=lambda i: values[i])[::-1]


def _npos(values):
    """Number of positive entries (of a row / column of the 4 x 4 nearest-neighbour counts)."""
    v = values.tolist() if hasattr(values, "tolist") else values     # plain ints: no numpy scalar per comparison
    if len(v) == 4:
        return (v[0] > 0) + (v[1] > 0) + (v[2] > 0) + (v[3] > 0)
    return sum(1 for x in v if x > 0)


def parse_records(raw: bytes):
    """parse_seq's record semantics (V20:441-455) in plain Python: the checker of csrc/fasta.cpp."""
    pieces = {}
    cur = None

    def strip(line):        # str.strip() as text mode sees the line (UTF-8; bytes that are not stay as they are), back as bytes
        return line.decode("utf-8", errors="surrogateescape").strip().encode("utf-8", errors="surrogateescape")

    for line in raw.splitlines():                   # bytes: \n, \r\n and \r end a line, nothing else (text mode's universal newlines)
        if line.startswith(b"#"):
            continue
        if line.startswith(b">"):
            cur = strip(line).split(b" ")[0]
        else:
            if cur is None:
                raise ValueError("sequence data before the first '>' header")
            pieces.setdefault(cur, []).append(strip(line))
    ids = [k.decode("utf-8", errors="surrogateescape") for k in pieces]
    rows = [b"".join(v) for v in pieces.values()]
    lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
    row_off = np.zeros(len(rows) + 1, np.int64)
    np.cumsum(lens, out=row_off[1:])
    data = np.frombuffer(b"".join(rows), dtype=np.uint8)
    return ids, data, row_off


class _Seed:
    """One seed primer (Viterbi "NM" or most-frequent "MM") and its refinement chain."""
    __slots__ = ("index", "chain", "cov", "stops", "first_cand", "final")

    def __init__(self, index):
        self.index = [int(x) for x in index]
        self.chain = []      # primer strings, chain[0] = the seed
        self.cov = []        # perfect coverage of each chain member (optimal_coverage_init)
        self.stops = []      # True where a structural break rule ends the loop after this member
        self.first_cand = -1
        self.final = None    # (primer, cov, F_mis, R_mis) after replay


def _merge_seen(dev, first, gapfree, items):
    """Insertion-ordered dict of a window after the exception k-mers joined it.

    `dev`: {k-mer: count} of the device entries in first-seen order, `first` their first rows (ascending), `items`:
    (row, j, key) in ascending (row, j), one sequence each.  A key takes the position of its earliest sighting — a
    plain row never shares its row with an exception — and counts add up (V20:689-711 run row by row would give
    exactly this dict).  Returns (dict, counts array, gap-free flags or None)."""
    if not items:
        return dev, np.fromiter(dev.values(), np.int64, len(dev)), gapfree
    first_of = None
    placed = {}                                                  # key -> [row, j, count]: new keys and keys that move up
    for row, j, e in items:
        if e in placed:
            placed[e][2] += 1
        elif e in dev:
            if first_of is None:
                first_of = dict(zip(dev, first.tolist()))
            if first_of[e] > row:                                # seen here before any plain row carried it: it moves up
                placed[e] = [row, j, dev[e] + 1]
            else:
                dev[e] += 1
        else:
            placed[e] = [row, j, 1]
    if not placed:
        return dev, np.fromiter(dev.values(), np.int64, len(dev)), gapfree
    keys = list(dev)
    vals = list(dev.values())
    flags = gapfree.tolist() if gapfree is not None else None
    rows = first
    moved = [e for e in placed if e in dev]
    if moved:
        drop = set(moved)
        keep = [e not in drop for e in keys]
        keys = list(compress(keys, keep))
        vals = list(compress(vals, keep))
        if flags is not None:
            flags = list(compress(flags, keep))
        rows = first[np.asarray(keep, bool)]
    pos = np.searchsorted(rows, [p[0] for p in placed.values()]).tolist()      # placed is in ascending (row, j)
    out_k, out_v, out_f, prev = [], [], [], 0
    for at, (e, (_, _, c)) in zip(pos, placed.items()):
        out_k += keys[prev:at]
        out_v += vals[prev:at]
        out_k.append(e)
        out_v.append(c)
        if flags is not None:
            out_f += flags[prev:at]
            out_f.append("-" not in e)
        prev = at
    out_k += keys[prev:]
    out_v += vals[prev:]
    if flags is not None:
        out_f += flags[prev:]
    return dict(zip(out_k, out_v)), np.asarray(out_v, np.int64), (np.asarray(out_f, bool) if flags is not None else None)


class _Window:
    __slots__ = ("w", "pos", "cover", "cover_number", "gap", "gap_number", "cbit", "tbit", "seeds",
                 "present", "dev_entries", "exc", "cnt", "gapfree")


class NN_degenerate(object):
    """Drop-in for the reference class of the same name (V20:342-345, 1133-1180)."""

    def __init__(self, seq_file, primer_length=18, coverage=0.8, number_of_dege_bases=18, score_of_dege_bases=1000,
                 product_len=250, position="2,-1", variation=2, raw_entropy_threshold=3.6, distance=4, GC="0.4,0.6",
                 nproc=10, outfile="", *, library: Library | None = None, device: int = 0, comm=None,
                 write_json: bool = True, write_bitsets: bool = False):
        self.primer_length = int(primer_length)
        self.coverage = coverage
        self.number_of_dege_bases = number_of_dege_bases
        self.score_of_dege_bases = score_of_dege_bases
        self.product = product_len
        self.position = position
        self.variation = int(variation)
        self.distance = distance
        self.GC = GC.split(",")
        self.nproc = nproc                      # accepted for CLI compatibility; the reference's pool is inert
        self.raw_entropy_threshold = raw_entropy_threshold
        self.outfile = outfile
        self.write_json = write_json
        self.write_bitsets = write_bitsets      # {out}.coverage_bitsets.npz: the bitset form of the two JSON files
        self.comm = comm                        # multiprime_amd.dist.RowShards or None
        k = self.primer_length
        if not 2 <= k <= 28:
            raise ValueError("primer length must be in [2, 28] for the packed window words")
        self.Y_strict, self.Y_strict_R = msa.strict_sets(position, k)
        self._sF = msa.strict_mask(self.Y_strict, k)
        self._sR = msa.strict_mask(self.Y_strict_R, k)
        self.stats = {}

        t0 = time.time()
        self.lib = library if library is not None else Library()      # raises if the HIP library is absent
        self.ctx = self.lib.context(device)
        with open(seq_file, "rb") as fh:
            ids, data, row_off = parse_records(fh.read())
        self.seq_ids = ids
        self.total_sequence_number = len(ids)
        if comm is not None:
            self.ctx.reserve_columns(int(np.diff(row_off).max()))      # windows span the whole alignment, not this shard's rows
            data, row_off = comm.take_shard(data, row_off)
        self.ctx.load_msa(data, row_off)
        lead, rstrip, _ = self.ctx.row_attributes()
        if comm is not None:
            lead, rstrip = comm.gather_rows(lead), comm.gather_rows(rstrip)
        start, stop = msa.region(lead, rstrip, self.coverage)
        if stop - start < int(self.product):     # V20:635-638
            print("Error: max length of PCR product is shorter than the default min Product length with {} "
                  "coverage! Non candidate primers !!!".format(self.coverage))
            sys.exit(1)
        self.start_position, self.stop_position, self.length = start, stop, stop - start
        self.entropy_threshold = self._entropy_threshold(self.length)
        self.stats["load_s"] = time.time() - t0

    def _entropy_threshold(self, length):       # V20:642-649
        if length < 5000:
            return self.raw_entropy_threshold
        return self.raw_entropy_threshold * (0.95 if length < 10000 else 0.9)

    # ------------------------------------------------------------------ device stage
    def _device_tables(self):
        k, v = self.primer_length, self.variation
        p0 = int(self.start_position)
        W = int(self.stop_position - self.start_position - k)
        self.n_windows = W
        if W <= 0:
            return None
        t0 = time.time()
        n_ex = self.ctx.build_windows(p0, W, k, v)
        ex_w, ex_r, ex_codes = self.ctx.get_exceptions(n_ex)
        exc = defaultdict(list)                  # window -> [(row, raw string)]
        self._expansions = {}                    # raw IUPAC k-mer -> its expansions (looked at three times per exception)
        row_base = self.comm.row0 if self.comm is not None else 0
        if n_ex:
            raw = iupac.strings_of(iupac.SYMBOL_LUT[ex_codes])
            extra_w, extra_k = [], []
            for w_, r_, s in zip(ex_w.tolist(), ex_r.tolist(), raw):
                exc[w_].append((r_ + row_base, s))
                if s.count("-") <= v:
                    exps = self._expansions.get(s)
                    if exps is None:
                        exps = self._expansions[s] = iupac.expand(s)
                    extra_w.extend([w_] * len(exps))
                    extra_k.extend(exps)
            if extra_w:
                chars = np.frombuffer("".join(extra_k).encode(), np.uint8).reshape(len(extra_k), k)
                self.ctx.set_extra_rows(np.asarray(extra_w, np.int32), iupac.words_of_kmers(chars))
        self.stats["build_windows_s"] = time.time() - t0
        t0 = time.time()
        # state_matrix / trans_matrix of every window (V20:541-577) straight from the column planes; shards add up
        self._freq, self._nn = self.ctx.window_stats()
        if self.comm is not None:
            self._freq, self._nn = self.comm.sum_int64(self._freq), self.comm.sum_int64(self._nn)
        self._nm_all = self._viterbi_all(self._freq, self._nn)
        self.stats["stats_s"] = time.time() - t0
        t0 = time.time()
        off, words, count, first = self.ctx.window_unique(want_labels=self.write_json)
        self.stats["unique_s"] = time.time() - t0
        first = first.astype(np.int64) + row_base
        if self.comm is not None:
            off, words, count, first, exc = self.comm.merge_tables(off, words, count, first, exc, W)
            if self.write_json:
                self.comm.gather_labels(self.ctx, W)
        t0 = time.time()
        chars = iupac.kmers_of_words(words, k)
        strs = iupac.strings_of(chars)
        gaps = (chars == ord("-")).sum(axis=1)
        self.stats["decode_s"] = time.time() - t0
        return off, strs, count, first, gaps, exc

    # ------------------------------------------------------------------ per-window host logic
    def _tables_of(self, w, off, strs, count, first, gaps, exc):
        """cover / gap_sequence of window w in the reference's dict insertion order (V20:689-711)."""
        v = self.variation
        a, b = int(off[w]), int(off[w + 1])
        win = _Window()
        win.w, win.pos = w, int(self.start_position) + w
        win.exc = exc.get(w)
        win.dev_entries = (a, b)
        # the device entries of a window are distinct and already in first-seen order: two dict(zip()) calls
        is_gap = gaps[a:b] > v
        cnt = count[a:b]
        if is_gap.any():
            keep = ~is_gap
            win.cnt = cnt[keep]
            win.cover = dict(zip(compress(strs[a:b], keep.tolist()), win.cnt.tolist()))
            win.gap = dict(zip(compress(strs[a:b], is_gap.tolist()), cnt[is_gap].tolist()))
            win.gapfree = gaps[a:b][keep] == 0
            first_c, first_g = first[a:b][keep], first[a:b][is_gap]
        else:
            win.cnt = cnt
            win.cover, win.gap = dict(zip(strs[a:b], cnt.tolist())), {}
            win.gapfree = gaps[a:b] == 0
            first_c, first_g = first[a:b], first[a:b][:0]
        if win.exc:
            # IUPAC k-mers (host-expanded, V20:368-380) join at the row they were seen in: (row, expansion index)
            exp_items, gap_items = [], []
            for row, s in win.exc:                               # ascending rows
                if s.count("-") > v:
                    gap_items.append((row, 0, s))                # gap_sequence is keyed by the raw string
                else:
                    exp_items.extend((row, j, e) for j, e in enumerate(self._expand(s)))
            win.cover, win.cnt, win.gapfree = _merge_seen(win.cover, first_c, win.gapfree, exp_items)
            win.gap, _, _ = _merge_seen(win.gap, first_g, None, gap_items)
        win.gap_number = sum(win.gap.values())
        n_exc_cover = sum(1 for _, s in win.exc if s.count("-") <= v) if win.exc else 0
        n_exp = sum(len(self._expand(s)) for _, s in win.exc if s.count("-") <= v) if win.exc else 0
        # cover_number counts sequences (V20:702), cover counts expansions (V20:704)
        win.cover_number = sum(win.cover.values()) - n_exp + n_exc_cover
        return win

    def _expand(self, s):
        """Expansions of an exception k-mer, memoised per run (other ranks' exceptions arrive unexpanded)."""
        exps = self._expansions.get(s)
        if exps is None:
            exps = self._expansions[s] = iupac.expand(s)
        return exps

    def _entropy(self, win):
        """entropy (V20:602-614), same summation order."""
        cn, gn = win.cover_number, win.gap_number
        tot = cn + gn
        if win.cnt is not None and len(win.cnt) > 64:
            # deep windows: one math.log per DISTINCT count (same libm call, same operands as the loop below), then a
            # strictly left-to-right float64 accumulation (ufunc.accumulate) in dict order — the same sums bit for bit
            uniq, inv = np.unique(win.cnt, return_inverse=True)
            tc = np.array([(c / cn) * math.log((c / cn), 2) for c in uniq.tolist()], np.float64)
            tt = np.array([(c / tot) * math.log((c / tot), 2) for c in uniq.tolist()], np.float64)
            cbit = float(np.add.accumulate(tc[inv])[-1])
            tbit = float(np.add.accumulate(tt[inv])[-1])
        else:
            cbit = 0
            tbit = 0
            for c in win.cover.values():
                cbit += (c / cn) * math.log((c / cn), 2)
                tbit += (c / tot) * math.log((c / tot), 2)
        for g in win.gap.values():
            tbit += (g / tot) * math.log((g / tot), 2)
        return round(-cbit, 2), round(-tbit, 2)

    def _plan_window(self, win):
        """Gates (V20:713-740), matrices, seeds and refinement chains of one window.
        Returns False if the window is rejected before any coverage evaluation."""
        k = self.primer_length
        if round(win.gap_number / self.total_sequence_number, 2) >= (1 - self.coverage):
            return False
        if len(win.cover) < 1:
            return False
        win.cbit, win.tbit = self._entropy(win)
        if win.tbit > self.entropy_threshold:
            return False
        # state_matrix (V20:541-554): per-column base counts over all rows of the window ('-' dropped) and
        # trans_matrix (V20:556-577): NN[j][a][b] over ACGT pairs only — counted on the device (mp_window_stats)
        freq = self._freq[win.w]
        if (freq.sum(axis=1) > 0).sum() < 4:         # fewer than 4 distinct bases (V20:736)
            return False
        if (freq.sum(axis=0) == 0).any():            # an all-gap column (V20:738)
            return False
        NN = self._nn[win.w].copy()                  # refinement merges rows / columns in place
        nm = self._nm_all[win.w].tolist()            # get_optimal_primer_by_viterbi, all windows at once (_viterbi_all)
        mm = None
        if win.cnt is not None:                      # get_optimal_primer_by_MM (V20:595-600): first of the most frequent
            if win.gapfree.any():
                mm = next(islice(win.cover, int(np.argmax(np.where(win.gapfree, win.cnt, 0))), None))
        else:
            best = 0
            for s, c in win.cover.items():
                if c > best and "-" not in s:
                    best, mm = c, s
        seeds = [_Seed(nm)]
        if mm is not None:
            mm_idx = [_B2I[c] for c in mm]
            if mm_idx != seeds[0].index:
                seeds.append(_Seed(mm_idx))
        nm_str = "".join(iupac.BASES[i] for i in seeds[0].index)
        win.present = nm_str                          # phantom key inserted at V20:787/800/835 (SURVEY A-13b)
        for s in seeds:
            self._build_chain(s, win.cover, NN)
        win.seeds = seeds
        return True

    @staticmethod
    def _viterbi_all(freq, nn):
        """get_optimal_primer_by_viterbi (V20:579-593) for every window at once: max-sum path over base
        frequencies (freq [W][4][k]) and nearest-neighbour counts (nn [W][k-1][4][4]); ties go to the lowest
        base index (numpy argmax takes the first).  Returns the base indices [W][k]."""
        W, _, k = freq.shape
        score = freq[:, :, 0].copy()                                     # [W][a]
        back = []
        for t in range(1, k):
            M = score[:, :, None] + nn[:, t - 1] + freq[:, None, :, t]   # M[w][a][b]: best score ending in a, then b
            back.append(M.argmax(axis=1))
            score = M.max(axis=1)
        rows = np.arange(W)
        path = [score.argmax(axis=1)]
        for arg in reversed(back):
            path.append(arg[rows, path[-1]])
        return np.stack(path[::-1], axis=1)

    # -- refinement ----------------------------------------------------------------------------
    def _perfect(self, cover, primer_list):
        """Sum of cover[] over the expansions of a primer (V20:954-956)."""
        get = cover.get
        s = "".join(primer_list)
        if not s.translate(_DROP_ACGT):                  # a concrete primer is its own only expansion
            return get(s, 0)
        return sum(get(e, 0) for e in iupac.expand(s))

    def _refine(self, primer, cov, cover, index, nn_cov, NN):
        """refine_by_NN_array (V20:922-1089): add one base next to the weakest nearest-neighbour
        link(s); among the weakest links keep the first one with the largest perfect coverage.

        Link i joins positions i and i+1; `index` is the seed's base index per position and
        NN[i][a][b] the (merged) count of base a at i followed by base b at i+1."""
        last = len(index) - 2
        lo = min(nn_cov)
        best = None
        for i in range(last + 1):
            if nn_cov[i] != lo:
                continue
            nn = NN.copy()
            cv = list(nn_cov)
            P = list(primer)
            c2 = cov
            row, col = index[i], index[i + 1]

            def widen(pos, ranking, skip):
                """Add the best-ranked base other than `skip` at position `pos`; returns it."""
                nonlocal c2
                for x in _desc_stable(ranking):
                    if x != skip:
                        P[pos] = iupac.BASES[x]
                        c2 += self._perfect(cover, P)                 # coverage gained (V20:954-956)
                        P[pos] = iupac.SYMBOL[iupac.MASK[primer[pos]] | (1 << x)]
                        return x
                return None

            if i == 0 and _npos(nn[0, :, col]) > 1:
                # position 0 (V20:941-965): fold predecessor x of `col` into the seed's row
                x = widen(0, nn[0, :, col].tolist(), row)
                if x is not None:
                    nn[0, row, :] += nn[0, x, :]
                    nn[0, x, :] = 0
                    cv[0] = nn[0, row, col]
            elif i == 0 and not _npos(nn[0, row, :]) > 1:
                pass                                                   # V20:1002-1003
            elif i == last and i != 0:
                # last position (V20:1004-1031)
                line = nn[i, row, :]
                if _npos(line) > 1:
                    x = widen(i + 1, line.tolist(), col)
                    if x is not None:
                        nn[i, :, col] += nn[i, :, x]
                        nn[i, :, x] = 0
                        cv[i] = nn[i, row, col]
            else:
                # inner position i+1, shared by link i and link i+1 (V20:967-1001, 1032-1072)
                nrow, ncol = index[i + 1], index[i + 2]
                both = np.minimum(nn[i, row, :], nn[i + 1, :, ncol])
                if _npos(both) > 1:
                    x = widen(i + 1, both.tolist(), col)
                    if x is not None:
                        nn[i, :, col] += nn[i, :, x]
                        nn[i, :, x] = 0
                        nn[i + 1, nrow, :] += nn[i + 1, x, :]
                        nn[i + 1, x, :] = 0
                        cv[i] = nn[i, row, col]
                        cv[i + 1] = nn[i + 1, nrow, ncol]
            if best is None or c2 > best[1]:
                best = (P, c2, cv, nn)
        P, c2, cv, nn = best
        return P, c2, cv, nn, iupac.degeneracy(P), iupac.n_degenerate(P)

    def _build_chain(self, seed, cover, NN):
        """Everything coverage_stast (V20:860-920) does that does not depend on an evaluation."""
        index = seed.index
        P = [iupac.BASES[i] for i in index]
        cov = cover.get("".join(P), 0)
        nn = NN.copy()
        nn_cov = [int(NN[i, index[i], index[i + 1]]) for i in range(len(index) - 1)]
        seed.chain.append("".join(P))
        seed.cov.append(cov)
        seed.stops.append(False)
        d, n = self.score_of_dege_bases, self.number_of_dege_bases
        while True:
            P, cov, cv, nn, deg, ndeg = self._refine(P, cov, cover, index, nn_cov, nn)
            cv = [int(x) for x in cv]
            stop = cv == nn_cov or 2 * deg > d or 3 * deg / 2 > d or ndeg == n     # V20:899-904
            seed.chain.append("".join(P))
            seed.cov.append(cov)
            seed.stops.append(stop)
            if stop:
                break
            nn_cov = cv

    def _self_dimers(self, primers):
        """dimer_check (V20:487-503) for a list of primers, by the checker's own restatement (oracle/filters_ref.py)."""
        hit = {p: filters.self_dimer(p) for p in dict.fromkeys(primers)}
        return [hit[p] for p in primers]

    def _replay(self, seed, ev, cn):
        """The stopping rules of coverage_stast (V20:881-906) on the batched evaluations."""
        i = 0
        base = seed.first_cand
        F, R = int(ev[base, 1]), int(ev[base, 2])
        if seed.cov[0] + F < cn or seed.cov[0] + R < cn:
            while seed.cov[i] + F < cn or seed.cov[i] + R < cn:
                i += 1
                F, R = int(ev[base + i, 1]), int(ev[base + i, 2])
                if max(F, R) == cn or seed.stops[i]:
                    break
        seed.final = (seed.chain[i], seed.cov[i], seed.cov[i] + F, seed.cov[i] + R, i)

    # ------------------------------------------------------------------ JSON side files
    def _rows_by_entry(self, win):
        """Rows (ascending) of every device histogram entry of the window, from the per-row labels."""
        a, b = win.dev_entries
        lab = self.comm.labels(win.w) if self.comm is not None else self.ctx.get_labels(win.w)
        order = np.argsort(lab, kind="stable")
        cnt = np.bincount(lab[lab >= 0], minlength=b - a)
        edge = np.concatenate(([0], np.cumsum(cnt))) + int((lab < 0).sum())
        return [order[edge[e]:edge[e + 1]] for e in range(b - a)]

    def _ids_by_kmer(self, win, strs, rows_by_entry, want, gap_rows):
        """{k-mer: [ids in file order]} for the k-mers in `want` — non_gap_seq_id (V20:707) when
        gap_rows is False, gap_seq_id (V20:698) when True — from the device labels plus the
        host-expanded exception rows."""
        ids = self.seq_ids
        a, b = win.dev_entries
        out = {s: [] for s in want}
        for e in range(b - a):
            s = strs[a + e]
            if s in out:
                out[s] = [int(r) for r in rows_by_entry[e]]
        if win.exc:
            touched = set()
            for row, raw in win.exc:
                if (raw.count("-") > self.variation) != gap_rows:
                    continue
                for e in iupac.expand(raw):
                    if e in out:
                        out[e].append(row)
                        touched.add(e)
            for e in touched:
                out[e].sort()
        return {s: [ids[r] for r in rows] for s, rows in out.items()}

    # ------------------------------------------------------------------ driver
    def run(self):
        k, v = self.primer_length, self.variation
        t_run = time.time()
        tables = self._device_tables()
        rows_out, non_cov_out, gap_out = [], {}, {}
        n_cand = 0
        exc = {}
        if tables is not None:
            off, strs, count, first, gaps, exc = tables
            t0 = time.time()
            windows = []
            cand_w, cand_s = [], []
            for w in range(self.n_windows):
                win = self._tables_of(w, off, strs, count, first, gaps, exc)
                if not self._plan_window(win):
                    continue
                for s in win.seeds:
                    s.first_cand = len(cand_w)
                    cand_w.extend([w] * len(s.chain))
                    cand_s.extend(s.chain)
                windows.append(win)
            self.stats["plan_s"] = time.time() - t0
            self.stats["windows_planned"] = len(windows)      # passed the gap / entropy / composition gates (V20:713-740)
            n_cand = len(cand_w)
            t0 = time.time()
            if n_cand:
                codes = iupac.MASK_LUT[np.frombuffer("".join(cand_s).encode(), np.uint8)].reshape(n_cand, k)
                if self.comm is not None:
                    ev = self.comm.eval_allreduce(self.ctx, np.asarray(cand_w, np.int32), codes, self._sF, self._sR)
                else:
                    ev = self.ctx.eval_candidates(np.asarray(cand_w, np.int32), codes, self._sF, self._sR)
            self.stats["eval_s"] = time.time() - t0
            self.stats["n_candidates"] = n_cand
            t0 = time.time()
            chosen_of = {}
            for win in windows:
                for s in win.seeds:
                    self._replay(s, ev, win.cover_number)
                    # the host's running perfect coverage and the device's count are the same quantity
                    base = s.first_cand
                    if any(int(ev[base + j, 0]) != s.cov[j] for j in range(len(s.chain))):
                        raise RuntimeError(f"perfect-coverage mismatch between host chain and device evaluation "
                                           f"at position {win.pos}")
                if len(win.seeds) == 2:
                    nm, mm = win.seeds[0].final, win.seeds[1].final
                    chosen = nm if (nm[2] + nm[3]) > (mm[2] + mm[3]) else mm       # V20:816
                else:
                    chosen = win.seeds[0].final
                chosen_of[win.w] = chosen
            # the 3'-end self-dimer test of every window's primer (dimer_check, V20:487-503) in ONE launch:
            # it is the ordered pair (x -> x) of the dimer scan with Loss >= 3 and the two-term deltaG
            order = [win for win in windows]
            dimer_flag = self._self_dimers([chosen_of[win.w][0] for win in order])
            for win, is_dimer in zip(order, dimer_flag):
                chosen = chosen_of[win.w]
                primer, cov, f_mis, r_mis, _ = chosen
                if is_dimer:                                                       # V20:749
                    continue
                members = iupac.expand(primer)
                nonsense = sum(1 for e in members if e not in win.cover and e != win.present)   # V20:846
                tms = [thermo.tm(e) for e in members]
                # statistics.mean is the correctly rounded exact mean; for one or two values plain float arithmetic is too
                tm_avg = round(tms[0] if len(tms) == 1 else ((tms[0] + tms[1]) / 2 if len(tms) == 2 else mean(tms)), 2)   # V20:849-852
                info = filters.pre_filter(primer, self.GC, self.distance)          # V20:911
                rows_out.append([win.pos, win.cbit, win.tbit, primer, iupac.n_degenerate(primer), nonsense, cov,
                                 f_mis, r_mis, tm_avg, info])
                if self.write_json:
                    non_cov_out[win.pos], gap_out[win.pos] = self._side_files(win, primer, strs)
            self.stats["finish_s"] = time.time() - t0
            if self.write_bitsets:
                t0 = time.time()
                self._write_bitsets(rows_out, exc)
                self.stats["bitsets_s"] = time.time() - t0
        if self.comm is None or self.comm.rank == 0:
            self._write(rows_out, non_cov_out, gap_out)
        self.stats["run_s"] = time.time() - t_run
        self.stats["n_windows"] = self.n_windows
        self.stats["n_rows"] = len(rows_out)

    def _side_files(self, win, primer, strs):
        """F/R non-coverage dicts of the final primer (V20:1107-1127) and gap_seq_id (V20:698)."""
        v = self.variation
        codes = iupac.codes_of(primer)
        f_keys, r_keys = [], []
        for q in win.cover:
            D = 0
            nd = 0
            for j, ch in enumerate(q):
                if ch == "-" or not (codes[j] >> _B2I[ch]) & 1:
                    D |= 1 << j
                    nd += 1
            if nd == 0:
                continue
            if nd > v or D & self._sF:
                f_keys.append(q)
            if nd > v or D & self._sR:
                r_keys.append(q)
        rows_by_entry = self._rows_by_entry(win)
        ids = self._ids_by_kmer(win, strs, rows_by_entry, set(f_keys) | set(r_keys), False)
        non_cov = [{q: ids[q] for q in f_keys}, {q: ids[q] for q in r_keys}]
        gap_keys = {}
        for g in win.gap:
            for e in iupac.expand(g):
                gap_keys.setdefault(e)
        gids = self._ids_by_kmer(win, strs, rows_by_entry, set(gap_keys), True)
        return non_cov, {g: gids[g] for g in gap_keys}

    def _write_bitsets(self, rows_out, exc):
        """Per output window, which sequences a forward / reverse primer there does NOT reach — exactly
        the union the pairing stage takes of gap_seq_id and non_coverage_seq_id (get_multiPrime_V8.py:
        560-567) — as bits, one per sequence, from one mp_eval_masks launch.  O(W x N / 8) bytes."""
        k, v = self.primer_length, self.variation
        p0 = int(self.start_position)
        n_out = len(rows_out)
        wins = np.asarray([int(r[0]) - p0 for r in rows_out], np.int32)
        if n_out:
            codes = iupac.MASK_LUT[np.frombuffer("".join(r[3] for r in rows_out).encode(), np.uint8)].reshape(n_out, k)
            nf, nr = self.ctx.eval_masks(wins, codes, self._sF, self._sR)
        else:
            nf = nr = np.zeros((0, 1), np.uint64)
        n_local = self.ctx.n_rows
        if self.comm is not None:
            # row shards are not multiples of 64: concatenate bit by bit across ranks, then re-pack
            bits = [np.unpackbits(m.view(np.uint8), axis=1, bitorder="little")[:, :n_local].astype(bool) for m in (nf, nr)]
            bits = [np.concatenate(self.comm._gather_objects(b), axis=1) for b in bits]
            n_total = bits[0].shape[1] if n_out else self.total_sequence_number
            nw = (n_total + 63) // 64
            packed = []
            for b in bits:
                pad = np.zeros((n_out, nw * 64), bool)
                pad[:, :b.shape[1]] = b
                packed.append(np.packbits(pad, axis=1, bitorder="little").view(np.uint64).reshape(n_out, nw))
        else:
            n_total = n_local
            packed = [np.array(nf, np.uint64), np.array(nr, np.uint64)]

        def put(which, i, row, value):
            word, bit = row >> 6, np.uint64(1) << np.uint64(row & 63)
            if value:
                packed[which][i, word] |= bit
            else:
                packed[which][i, word] &= ~bit

        # rows whose window held an IUPAC code: every expansion must be reached (V20:701-707 puts the id
        # under each expansion's k-mer), gap-type ones are in gap_seq_id
        for i, row in enumerate(rows_out):
            lst = exc.get(int(wins[i]))
            if not lst:
                continue
            pc = iupac.codes_of(row[3])
            for r_glob, raw in lst:
                if raw.count("-") > v:
                    put(0, i, r_glob, True)
                    put(1, i, r_glob, True)
                    continue
                bad_f = bad_r = False
                for e in iupac.expand(raw):
                    D, nd = 0, 0
                    for j, ch in enumerate(e):
                        if ch == "-" or not (pc[j] >> _B2I[ch]) & 1:
                            D |= 1 << j
                            nd += 1
                    if nd == 0:
                        continue
                    bad_f |= nd > v or bool(D & self._sF)
                    bad_r |= nd > v or bool(D & self._sR)
                put(0, i, r_glob, bad_f)
                put(1, i, r_glob, bad_r)
        if self.comm is not None and self.comm.rank != 0:
            return
        np.savez(self.outfile + ".coverage_bitsets.npz", positions=np.asarray([int(r[0]) for r in rows_out], np.int64),
                            not_f=packed[0], not_r=packed[1], n_seq=np.int64(n_total), ids=np.asarray(self.seq_ids))

    def _write(self, rows_out, non_cov_out, gap_out):
        with open(self.outfile, "w") as fo:                                        # V20:1148-1170
            fo.write("\t".join(HEADERS) + "\n")
            for row in rows_out:
                fo.write("\t".join(map(str, row)) + "\n")
        if self.write_json:
            with open(self.outfile + ".non_coverage_seq_id_json", "w") as fj:      # V20:1172-1173 json.dump(.., indent=4)
                _dump_side_file(non_cov_out, fj, True)
            with open(self.outfile + ".gap_seq_id_json", "w") as fg:               # V20:1175-1176
                _dump_side_file(gap_out, fg, False)
