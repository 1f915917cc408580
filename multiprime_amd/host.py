"""ctypes binding of include/mprime_host.h — the native host stage of the core step (C++ inside
`csrc/libmprime_hip.so`: FASTA record parser, per-window planning).  Pure host code: it loads and runs without a
GPU, takes and returns numpy arrays, and is always served by the product library (there is no Python fallback;
the pure-Python restatement lives in oracle/core_ref.py as test infrastructure).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import time

import numpy as np

from ._abi import HIP_LIB, MP_ERR_CAPACITY, MprimeError, _ptr

_p = C.c_void_p


class PlanParams(C.Structure):
    _fields_ = [("k", C.c_int32), ("v", C.c_int32), ("n_windows", C.c_int32), ("n_threads", C.c_int32),
                ("total_sequences", C.c_int64), ("coverage", C.c_double), ("entropy_threshold", C.c_double),
                ("max_degeneracy", C.c_double), ("max_dege_positions", C.c_int32), ("keep_tables", C.c_int32)]


# every symbol include/mprime_host.h declares: (name, restype, argtypes)
HOST_SYMBOLS = [
    ("mp_fasta_parse_file", C.c_int, [C.c_char_p, C.c_int32, C.POINTER(_p)]),
    ("mp_fasta_parse_buffer", C.c_int, [_p, C.c_int64, C.c_int32, C.POINTER(_p)]),
    ("mp_fasta_destroy", None, [_p]),
    ("mp_fasta_error", C.c_char_p, [_p]),
    ("mp_fasta_sizes", C.c_int, [_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("mp_fasta_rows", C.c_int, [_p, _p, _p]),
    ("mp_fasta_gather", C.c_int, [_p, C.c_int64, C.c_int64, _p, C.c_int32]),
    ("mp_load_msa_fasta", C.c_int, [_p, _p]),
    ("mp_fasta_ids", C.c_int, [_p, _p, _p]),
    ("mp_file_count_newlines", C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_int64)]),
    ("mp_plan_create", C.c_int, [C.POINTER(PlanParams), C.c_int64, _p, _p, _p, _p, C.c_int64, _p, _p, _p, _p, _p, C.POINTER(_p)]),
    ("mp_plan_create_segments", C.c_int, [C.POINTER(PlanParams), _p, _p, _p, _p, C.c_int64, C.c_int64, _p, _p, _p, _p, _p, C.POINTER(_p)]),
    ("mp_plan_create_streamed", C.c_int, [_p, C.POINTER(PlanParams), C.c_int64, C.c_int64, _p, _p, _p, _p, _p, C.POINTER(C.c_int64), C.POINTER(_p)]),
    ("mp_plan_destroy", None, [_p]),
    ("mp_plan_error", C.c_char_p, [_p]),
    ("mp_plan_windows", C.c_int, [_p, _p, _p, _p, _p, _p]),
    ("mp_plan_sizes", C.c_int, [_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    ("mp_plan_candidates", C.c_int, [_p, _p, _p]),
    ("mp_plan_seeds", C.c_int, [_p, C.c_int32, _p, _p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("mp_plan_chain", C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, _p, _p, _p, C.POINTER(C.c_int32)]),
    ("mp_plan_finish", C.c_int, [_p, _p]),
    ("mp_plan_results", C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    ("mp_plan_window_table", C.c_int, [_p, C.c_int32, C.c_int32, C.c_int64, _p, _p, _p, C.POINTER(C.c_int64)]),
    ("mp_plan_write_side_files", C.c_int, [_p, C.c_int32, _p, _p, _p, C.c_uint64, C.c_uint64, _p, _p, C.c_int64, _p, C.c_int32, C.c_int64,
                                           _p, _p, _p, _p, _p, C.c_char_p, C.c_char_p]),
    ("mp_plan_write_side_files_part", C.c_int, [_p, C.c_int32, _p, _p, _p, C.c_uint64, C.c_uint64, _p, _p, C.c_int64, _p, C.c_int32, C.c_int64,
                                                _p, _p, _p, _p, _p, C.c_char_p, C.c_char_p, C.c_int32]),
    ("mp_expand_kmer_words", C.c_int, [C.c_int32, C.c_int64, _p, C.c_int64, _p, _p, C.POINTER(C.c_int64)]),
    ("mp_expand_kmers", C.c_int, [C.c_int32, C.c_int64, _p, C.c_int64, _p, _p, C.POINTER(C.c_int64)]),
    ("mp_primer_tm", C.c_int, [C.c_int32, C.c_int64, _p, _p, _p]),
    ("mp_primer_filters", C.c_int, [C.c_int32, C.c_int64, _p, _p, C.c_int32, _p, _p, _p]),
    ("mp_exception_verdicts", C.c_int, [C.c_int32, C.c_int32, C.c_int64, _p, _p, C.c_int64, _p, C.c_uint64, C.c_uint64, _p]),
    ("mp_expand_exception_words", C.c_int, [C.c_int32, C.c_int32, C.c_int64, _p, _p, C.c_int64, _p, _p, _p]),
    ("mp_exception_assignments", C.c_int, [C.c_int32, C.c_int32, C.c_int64, _p, _p, _p, C.c_int32, _p, C.c_int64, C.c_int64, C.c_int64, _p, C.c_uint64,
                                           C.c_uint64, _p, _p, _p, _p, _p]),
]

_dll = None


def dll():
    """The product library's host-stage entry points (loaded once; raises if the library is missing)."""
    global _dll
    if _dll is None:
        path = os.environ.get("MP_HOST_LIB", HIP_LIB)       # MP_HOST_LIB: a sanitizer build of hostplan.cpp + fasta.cpp (tools/sanitize_host.sh)
        if not os.path.exists(path):
            raise MprimeError(-2, f"{HIP_LIB} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                                  "the host stage has no Python fallback")
        if os.path.basename(path).startswith("libmprime_hip"):
            from ._abi import one_hip_runtime
            one_hip_runtime()
        d = C.CDLL(path)
        for name, res, args in HOST_SYMBOLS:
            if name in ("mp_plan_create_streamed", "mp_load_msa_fasta") and not hasattr(d, name):
                continue                   # a host-only build (MP_HOST_LIB): the entry point that takes a device context lives in unique.hip
            fn = getattr(d, name)
            fn.restype = res
            fn.argtypes = args
        _dll = d
    return _dll


def serves_device_library(lib) -> bool:
    """True when the host stage loaded here is the SAME build as the device library `lib` and exports the entry point that takes one of
    its contexts (mp_plan_create_streamed).  MP_HOST_LIB may name a host-only / sanitizer build and MPRIME_LIBRARY another device
    build: an mp_ctx of one build must never be handed to code of another, so callers then take the blocking route."""
    d = dll()
    if not hasattr(d, "mp_plan_create_streamed"):
        return False
    mine = os.environ.get("MP_HOST_LIB", HIP_LIB)
    try:
        return os.path.samefile(mine, lib.path)
    except OSError:
        return False


class Fasta:
    """Records of a FASTA / alignment file (parse_seq's record semantics, V20:441-455)."""

    def __init__(self, path: str | None = None, raw: bytes | None = None, n_threads: int = 0):
        self.d = dll()
        h = _p()
        if raw is not None:
            buf = np.frombuffer(raw, np.uint8)
            rc = self.d.mp_fasta_parse_buffer(_ptr(buf) if len(buf) else None, len(buf), n_threads, C.byref(h))
        else:
            rc = self.d.mp_fasta_parse_file(os.fsencode(path), n_threads, C.byref(h))
        self.h = h
        if rc != 0:
            msg = self.d.mp_fasta_error(h).decode(errors="replace") if h else "mp_fasta_parse failed"
            self.close()
            if "before the first" in msg:
                raise ValueError(msg)
            if rc == -1:
                raise OSError(msg)
            raise MprimeError(rc, msg)
        n, nb, ni = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        self.d.mp_fasta_sizes(h, C.byref(n), C.byref(nb), C.byref(ni))
        self.n_rows, self.n_bytes, self.n_id_bytes = n.value, nb.value, ni.value
        self._ids = None

    def rows(self):
        """(data, row_off): the residue bytes of all records back to back — the input of mp_load_msa."""
        data = np.empty(max(self.n_bytes, 1), np.uint8)
        off = np.empty(self.n_rows + 1, np.int64)
        rc = self.d.mp_fasta_rows(self.h, _ptr(data), _ptr(off))
        if rc != 0:
            raise MprimeError(rc, "mp_fasta_rows")
        return data[: self.n_bytes], off

    def row_offsets(self):
        """row_off alone (no residue bytes are copied)."""
        off = np.empty(self.n_rows + 1, np.int64)
        rc = self.d.mp_fasta_rows(self.h, None, _ptr(off))
        if rc != 0:
            raise MprimeError(rc, "mp_fasta_rows")
        return off

    def load_into(self, ctx):
        """mp_load_msa_fasta: the records' residue bytes from the parsed file straight through the context's registered transfer
        buffers to the device (no intermediate array).  The caller checked serves_device_library(ctx.lib)."""
        rc = self.d.mp_load_msa_fasta(ctx.h, self.h)
        if rc != 0:
            raise MprimeError(rc, ctx.d.mp_last_error(ctx.h).decode())
        ctx.n_rows = self.n_rows

    def ids_raw(self):
        """(bytes as a uint8 array, offsets): the ids as they stand in the file, id r = bytes[off[r]:off[r+1]]."""
        buf = np.empty(max(self.n_id_bytes, 1), np.uint8)
        off = np.empty(self.n_rows + 1, np.int64)
        self.d.mp_fasta_ids(self.h, _ptr(buf), _ptr(off))
        return buf[: self.n_id_bytes], off

    @property
    def ids(self):
        """Sequence ids (first-appearance order), decoded like the reference's text-mode read (UTF-8)."""
        if self._ids is None:
            buf = np.empty(max(self.n_id_bytes, 1), np.uint8)
            off = np.empty(self.n_rows + 1, np.int64)
            self.d.mp_fasta_ids(self.h, _ptr(buf), _ptr(off))
            raw = buf[: self.n_id_bytes].tobytes()
            try:
                text = raw.decode("ascii")
                o = off.tolist()
                self._ids = [text[a:b] for a, b in zip(o[:-1], o[1:])]
            except UnicodeDecodeError:
                o = off.tolist()
                self._ids = [raw[a:b].decode("utf-8", errors="surrogateescape") for a, b in zip(o[:-1], o[1:])]
        return self._ids

    def close(self):
        if getattr(self, "h", None):
            self.d.mp_fasta_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def count_newlines(path, n_threads: int = 0) -> int:
    """str.count("\\n") of the file read in text mode (\\n, \\r\\n, lone \\r), counted natively on several threads."""
    n = C.c_int64(0)
    rc = dll().mp_file_count_newlines(os.fsencode(path), n_threads, C.byref(n))
    if rc != 0:
        raise OSError(f"cannot read {path}")
    return n.value


def exception_verdicts(xc: np.ndarray, primer_of: np.ndarray, primers: np.ndarray, v: int, strictF: int, strictR: int) -> np.ndarray:
    """[n][2] bool: the forward / reverse output primer primer_of[i] does NOT reach exception row i (mp_exception_verdicts)."""
    xc = np.ascontiguousarray(xc, np.uint8)
    primers = np.ascontiguousarray(primers, np.uint8)
    primer_of = np.ascontiguousarray(primer_of, np.int64)
    n, k = xc.shape
    bad = np.empty((n, 2), np.uint8)
    rc = dll().mp_exception_verdicts(k, int(v), n, _ptr(xc), _ptr(primer_of), len(primers), _ptr(primers), int(strictF), int(strictR), _ptr(bad))
    if rc != 0:
        raise MprimeError(rc, "mp_exception_verdicts: bad arguments")
    return bad.view(bool)


def exception_assignments(x_window, x_row, xc, slot_of, row0: int, n_rows: int, primers, v: int, strictF: int, strictR: int):
    """(cand, row, which, value) for Context.masks_set_bits: the verdicts of the exception rows whose window is an output window
    (slot_of[window] >= 0) and whose row lies in [row0, row0 + n_rows) — selection, verdicts and layout in one native call
    (mp_exception_assignments)."""
    x_window = np.ascontiguousarray(x_window, np.int32)
    x_row = np.ascontiguousarray(x_row, np.int64)
    xc = np.ascontiguousarray(xc, np.uint8)
    slot_of = np.ascontiguousarray(slot_of, np.int32)
    primers = np.ascontiguousarray(primers, np.uint8)
    n = len(x_window)
    k = xc.shape[1] if xc.ndim == 2 else (primers.shape[1] if primers.ndim == 2 else 1)
    cand, row = np.empty(2 * n, np.int32), np.empty(2 * n, np.int32)
    which, value = np.empty(2 * n, np.uint8), np.empty(2 * n, np.uint8)
    n_out = C.c_int64(0)
    rc = dll().mp_exception_assignments(k, int(v), n, _ptr(x_window), _ptr(x_row), _ptr(xc), len(slot_of), _ptr(slot_of), int(row0), int(n_rows),
                                        len(primers), _ptr(primers), int(strictF), int(strictR), _ptr(cand), _ptr(row), _ptr(which), _ptr(value),
                                        C.byref(n_out))
    if rc != 0:
        raise MprimeError(rc, "mp_exception_assignments: bad arguments")
    m = n_out.value
    return cand[:m], row[:m], which[:m], value[:m]


def expand_kmers(codes: np.ndarray):
    """(expansions [m][k], src [m]): degenerate_seq (V20:368-380) of every k-mer of symbol codes, reference order."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, k = codes.shape
    d = dll()
    need = C.c_int64(0)
    rc = d.mp_expand_kmers(k, n, _ptr(codes), 0, None, None, C.byref(need))
    if rc not in (0, MP_ERR_CAPACITY):
        raise MprimeError(rc, "mp_expand_kmers: bad symbol codes or too many expansions")
    m = need.value
    if m > 1 << 28:
        raise MprimeError(MP_ERR_CAPACITY, f"IUPAC k-mers expand to {m} concrete k-mers")
    out = np.empty((max(m, 1), k), np.uint8)
    src = np.empty(max(m, 1), np.int64)
    rc = d.mp_expand_kmers(k, n, _ptr(codes), m, _ptr(out), _ptr(src), C.byref(need))
    if rc != 0:
        raise MprimeError(rc, "mp_expand_kmers")
    return out[:m], src[:m]


def expand_kmer_words(codes: np.ndarray):
    """(words [m][3] uint32 — uint64 for k > 31 —, src [m]): the expansions of expand_kmers as window words (mp_set_extra_rows)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n, k = codes.shape
    d = dll()
    need = C.c_int64(0)
    wt = np.uint32 if k <= 31 else np.uint64
    # one call when the guess holds (exception k-mers carry one or two IUPAC codes: 2-4 expansions each), else the size, then the call
    m = 4 * n + 1024
    for _ in range(2):
        words = np.empty((max(m, 1), 3), wt)
        src = np.empty(max(m, 1), np.int64)
        rc = d.mp_expand_kmer_words(k, n, _ptr(codes), m, _ptr(words), _ptr(src), C.byref(need))
        if rc != MP_ERR_CAPACITY:
            break
        m = need.value
        if m > 1 << 28:
            raise MprimeError(MP_ERR_CAPACITY, f"IUPAC k-mers expand to {m} concrete k-mers")
    if rc != 0:
        raise MprimeError(rc, "mp_expand_kmer_words: bad symbol codes or too many expansions")
    m = need.value
    return words[:m], src[:m]


def expand_exception_words(x_window: np.ndarray, codes: np.ndarray, v: int):
    """(windows [m] int32, words [m][3]): the expansions of the exception k-mers with at most v gaps, each with its row's window — the
    arguments of Context.set_extra_rows (mp_expand_exception_words)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    x_window = np.ascontiguousarray(x_window, dtype=np.int32)
    n, k = codes.shape
    d = dll()
    need = C.c_int64(0)
    wt = np.uint32 if k <= 31 else np.uint64
    m = 4 * n + 1024
    for _ in range(2):
        words = np.empty((max(m, 1), 3), wt)
        win = np.empty(max(m, 1), np.int32)
        rc = d.mp_expand_exception_words(k, int(v), n, _ptr(x_window), _ptr(codes), m, _ptr(words), _ptr(win), C.byref(need))
        if rc != MP_ERR_CAPACITY:
            break
        m = need.value
        if m > 1 << 28:
            raise MprimeError(MP_ERR_CAPACITY, f"IUPAC k-mers expand to {m} concrete k-mers")
    if rc != 0:
        raise MprimeError(rc, "mp_expand_exception_words: bad symbol codes or too many expansions")
    m = need.value
    return win[:m], words[:m]


class Plan:
    """Per-window planning of one alignment (mp_plan_*)."""

    def __init__(self, *, k, v, n_windows, total_sequences, coverage, entropy_threshold, max_degeneracy, max_dege_positions,
                 e_window=None, e_words=None, e_count=None, e_first=None, x_window, x_row, x_codes, freq, nn, keep_tables=False, n_threads=0,
                 e_off=None, row_base=0, device_context=None):
        """Entries either with a window per entry (`e_window`, any order: several ranks' tables concatenated; counts and GLOBAL first
        rows as int64) or as ONE rank's read-back as it stands: `e_off` [W+1] window segments, int32 counts and LOCAL first rows plus
        `row_base` (mp_plan_create_segments: no per-entry window array, no widening copies) — or, with `device_context` (a HIP context
        whose window_unique has run, ctx.window_unique_device()), not at all: mp_plan_create_streamed reads them back in bands beside the
        planning."""
        self.d = dll()
        self.k, self.W = int(k), int(n_windows)
        P = PlanParams(int(k), int(v), int(n_windows), int(n_threads), int(total_sequences), float(coverage), float(entropy_threshold),
                       float(max_degeneracy), int(max_dege_positions), int(bool(keep_tables)))
        segments = e_off is not None
        streamed = device_context is not None
        if streamed:
            n = 0
        elif segments:
            e_off = np.ascontiguousarray(e_off, dtype=np.int64)
            assert len(e_off) == self.W + 1
            n = int(e_off[-1])
            e_count = np.ascontiguousarray(e_count, dtype=np.int32)
            e_first = np.ascontiguousarray(e_first, dtype=np.int32)
        else:
            e_window = np.ascontiguousarray(e_window, dtype=np.int32)
            n = len(e_window)
            e_count = np.ascontiguousarray(e_count, dtype=np.int64)
            e_first = np.ascontiguousarray(e_first, dtype=np.int64)
        if not streamed:
            e_words = np.ascontiguousarray(e_words, dtype=np.uint32 if self.k <= 31 else np.uint64).reshape(3, n)
            assert len(e_count) == n and len(e_first) == n
        x_window = np.ascontiguousarray(x_window, dtype=np.int32)
        x_row = np.ascontiguousarray(x_row, dtype=np.int64)
        x_codes = np.ascontiguousarray(x_codes, dtype=np.uint8).reshape(len(x_window), self.k)
        freq = np.ascontiguousarray(freq, dtype=np.int64)
        nn = np.ascontiguousarray(nn, dtype=np.int64)
        assert freq.shape == (self.W, 4, self.k) and nn.shape == (self.W, self.k - 1, 4, 4)
        h = _p()
        if streamed:
            ne = C.c_int64(0)
            t_call = time.time()
            rc = self.d.mp_plan_create_streamed(device_context.h, C.byref(P), int(row_base), len(x_window), _ptr(x_window), _ptr(x_row), _ptr(x_codes),
                                                _ptr(freq), _ptr(nn), C.byref(ne), C.byref(h))
            self.n_entries = ne.value
            if os.environ.get("MP_TRACE"):
                print("[mprime] host.Plan: mp_plan_create_streamed call %.3f ms" % ((time.time() - t_call) * 1e3), file=sys.stderr)
        elif segments:
            rc = self.d.mp_plan_create_segments(C.byref(P), _ptr(e_off), _ptr(e_words), _ptr(e_count), _ptr(e_first), int(row_base), len(x_window),
                                                _ptr(x_window), _ptr(x_row), _ptr(x_codes), _ptr(freq), _ptr(nn), C.byref(h))
        else:
            rc = self.d.mp_plan_create(C.byref(P), n, _ptr(e_window), _ptr(e_words), _ptr(e_count), _ptr(e_first), len(x_window),
                                       _ptr(x_window), _ptr(x_row), _ptr(x_codes), _ptr(freq), _ptr(nn), C.byref(h))
        self.h = h
        if rc != 0:
            msg = self.d.mp_plan_error(h).decode() if h else "mp_plan_create failed"
            self.close()
            raise MprimeError(rc, msg)
        np_, nc = C.c_int32(0), C.c_int64(0)
        self.d.mp_plan_sizes(h, C.byref(np_), C.byref(nc))
        self.n_planned, self.n_candidates = np_.value, nc.value

    def _ck(self, rc):
        if rc != 0:
            raise MprimeError(rc, self.d.mp_plan_error(self.h).decode())

    def windows(self):
        """(status, cover_number, gap_number, cbit, tbit) of every window."""
        st = np.empty(self.W, np.int32)
        cn = np.empty(self.W, np.int64)
        gn = np.empty(self.W, np.int64)
        cb = np.empty(self.W, np.float64)
        tb = np.empty(self.W, np.float64)
        self._ck(self.d.mp_plan_windows(self.h, _ptr(st), _ptr(cn), _ptr(gn), _ptr(cb), _ptr(tb)))
        return st, cn, gn, cb, tb

    def candidates(self):
        n = self.n_candidates
        cw = np.empty(max(n, 1), np.int32)
        codes = np.empty((max(n, 1), self.k), np.uint8)
        self._ck(self.d.mp_plan_candidates(self.h, _ptr(cw), _ptr(codes)))
        return cw[:n], codes[:n]

    def seeds(self, w: int):
        nm = np.zeros(self.k, np.uint8)
        mm = np.zeros(self.k, np.uint8)
        has, a, b = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._ck(self.d.mp_plan_seeds(self.h, int(w), _ptr(nm), _ptr(mm), C.byref(has), C.byref(a), C.byref(b)))
        return nm, (mm if has.value else None), a.value, b.value

    def chain(self, w: int, seed: int):
        """(codes [n][k], cov [n], stops [n]) of the refinement chain of seed 0 (NM) / 1 (MM) of window w."""
        cap = 4 * 32 + 16
        codes = np.empty((cap, self.k), np.uint8)
        cov = np.empty(cap, np.int64)
        stops = np.empty(cap, np.uint8)
        n = C.c_int32(0)
        self._ck(self.d.mp_plan_chain(self.h, int(w), int(seed), cap, _ptr(codes), _ptr(cov), _ptr(stops), C.byref(n)))
        return codes[: n.value], cov[: n.value], stops[: n.value]

    def finish(self, ev: np.ndarray):
        ev = np.ascontiguousarray(ev, dtype=np.int64).reshape(-1, 3)
        if len(ev) != self.n_candidates:
            raise ValueError("one evaluation triple per candidate expected")
        self._ck(self.d.mp_plan_finish(self.h, _ptr(ev) if len(ev) else _ptr(np.zeros(3, np.int64))))

    def results(self):
        n = self.n_planned
        m = max(n, 1)
        r = {"window": np.empty(m, np.int32), "cbit": np.empty(m, np.float64), "tbit": np.empty(m, np.float64),
             "codes": np.empty((m, self.k), np.uint8), "cov": np.empty(m, np.int64), "f_mis": np.empty(m, np.int64),
             "r_mis": np.empty(m, np.int64), "nonsense": np.empty(m, np.int32), "n_dege": np.empty(m, np.int32),
             "cover_number": np.empty(m, np.int64)}
        self._ck(self.d.mp_plan_results(self.h, *[_ptr(r[key]) for key in ("window", "cbit", "tbit", "codes", "cov", "f_mis", "r_mis",
                                                                             "nonsense", "n_dege", "cover_number")]))
        return {key: val[:n] for key, val in r.items()}

    def window_table(self, w: int, which: int):
        """(codes [n][k], counts [n], first_row [n]) of the cover (which = 0) or gap_sequence (1) dict of window w, in
        insertion order."""
        n = C.c_int64(0)
        rc = self.d.mp_plan_window_table(self.h, int(w), int(which), 0, None, None, None, C.byref(n))
        if rc not in (0, MP_ERR_CAPACITY):
            self._ck(rc)
        m = n.value
        codes = np.empty((max(m, 1), self.k), np.uint8)
        counts = np.empty(max(m, 1), np.int64)
        first = np.empty(max(m, 1), np.int64)
        self._ck(self.d.mp_plan_window_table(self.h, int(w), int(which), m, _ptr(codes), _ptr(counts), _ptr(first), C.byref(n)))
        return codes[:m], counts[:m], first[:m]

    def write_side_files(self, out_window, out_pos, primer_codes, strictF, strictR, dev_off, dev_words, labels, x_window, x_row,
                         x_codes, ids_bytes, ids_off, noncov_path, gap_path, part=3):
        """The two JSON side files of the core step, written natively (byte-identical to json.dump(..., indent=4)).  `part`: bit 0 =
        first run of output windows (creates the files), bit 1 = last run (closes the objects); 3 = all of them in one call."""
        out_window = np.ascontiguousarray(out_window, dtype=np.int32)
        out_pos = np.ascontiguousarray(out_pos, dtype=np.int64)
        primer_codes = np.ascontiguousarray(primer_codes, dtype=np.uint8).reshape(len(out_window), self.k)
        dev_off = np.ascontiguousarray(dev_off, dtype=np.int64)
        dev_words = np.ascontiguousarray(dev_words, dtype=np.uint32 if self.k <= 31 else np.uint64)
        n_dev = dev_words.shape[1] if dev_words.ndim == 2 else 0
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        n_rows = labels.shape[1] if labels.ndim == 2 else 0
        x_window = np.ascontiguousarray(x_window, dtype=np.int32)
        x_row = np.ascontiguousarray(x_row, dtype=np.int64)
        x_codes = np.ascontiguousarray(x_codes, dtype=np.uint8)
        ids_bytes = np.ascontiguousarray(ids_bytes, dtype=np.uint8)
        ids_off = np.ascontiguousarray(ids_off, dtype=np.int64)
        self._ck(self.d.mp_plan_write_side_files_part(self.h, len(out_window), _ptr(out_window), _ptr(out_pos), _ptr(primer_codes), int(strictF),
                                                      int(strictR), _ptr(dev_off), _ptr(dev_words), n_dev, _ptr(labels), n_rows, len(x_window),
                                                      _ptr(x_window), _ptr(x_row), _ptr(x_codes), _ptr(ids_bytes), _ptr(ids_off),
                                                      os.fsencode(noncov_path), os.fsencode(gap_path), int(part)))

    def close(self):
        if getattr(self, "h", None):
            self.d.mp_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
