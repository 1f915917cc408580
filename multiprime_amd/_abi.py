"""ctypes binding of include/mprime.h.

The product loads `csrc/libmprime_hip.so` (hand-written HIP for gfx950) and fails loudly
when it is missing or cannot create a context on the GPU: there is no CPU fallback.
`Library(path)` accepts any library exporting the same C ABI; the test-suite uses that to
drive the host logic with the CPU oracle — the product never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

MP_MAX_K = 63
MP_NARROW_K = 31          # window words are uint32 up to here, uint64 above (mprime.h MP_WORD_BYTES)


def word_dtype(k: int):
    """numpy dtype of the window words of primer length k."""
    return np.uint32 if k <= MP_NARROW_K else np.uint64
MP_WIN_SKIP = 0x80000000
MP_ERR_CAPACITY = -4
MP_ERR_SHORT_WINDOW = -5

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(HERE, "csrc", "libmprime_hip.so")

# every symbol include/mprime.h declares: (name, restype, argtypes)
_p = C.c_void_p
SYMBOLS = [
    ("mp_create", C.c_int, [C.c_int, C.POINTER(_p)]),
    ("mp_destroy", None, [_p]),
    ("mp_last_error", C.c_char_p, [_p]),
    ("mp_backend_name", C.c_char_p, []),
    ("mp_set_stream", C.c_int, [_p, _p]),
    ("mp_reserve_columns", C.c_int, [_p, C.c_int32]),
    ("mp_load_msa", C.c_int, [_p, _p, _p, C.c_int32]),
    ("mp_row_attributes", C.c_int, [_p, _p, _p, _p]),
    ("mp_row_histograms", C.c_int, [_p, C.c_int32, _p, _p]),
    ("mp_build_windows", C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    ("mp_get_exceptions", C.c_int, [_p, C.c_int32, _p, _p, _p]),
    ("mp_set_extra_rows", C.c_int, [_p, C.c_int32, _p, _p]),
    ("mp_get_window_words", C.c_int, [_p, C.c_int32, C.c_int32, C.c_int32, _p]),
    ("mp_window_stats", C.c_int, [_p, _p, _p]),
    ("mp_window_unique", C.c_int, [_p, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]),
    ("mp_set_entropy_gate", C.c_int, [_p, C.c_double]),
    ("mp_entropy_gate_result", C.c_int, [_p, C.POINTER(C.c_int32), _p]),
    ("mp_get_unique", C.c_int, [_p, _p, _p, _p, _p]),
    ("mp_get_labels", C.c_int, [_p, C.c_int32, _p]),
    ("mp_get_labels_many", C.c_int, [_p, C.c_int32, _p, _p]),
    ("mp_eval_candidates", C.c_int, [_p, C.c_int32, _p, _p, C.c_uint64, C.c_uint64, _p]),
    ("mp_eval_masks", C.c_int, [_p, C.c_int32, _p, _p, C.c_uint64, C.c_uint64, _p, _p]),
    ("mp_eval_masks_resident", C.c_int, [_p, C.c_int32, _p, _p, C.c_uint64, C.c_uint64]),
    ("mp_masks_set_bits", C.c_int, [_p, C.c_int64, _p, _p, _p, _p]),
    ("mp_masks_fetch", C.c_int, [_p, _p, _p]),
    ("mp_pair_coverage_resident", C.c_int, [_p, C.c_int64, _p, _p]),
    ("mp_eval_upload", C.c_int, [_p, C.c_int32, _p, _p, C.c_uint64, C.c_uint64]),
    ("mp_eval_launch", C.c_int, [_p, _p]),
    ("mp_eval_launch_rotating", C.c_int, [_p, _p, _p]),
    ("mp_window_stats_begin", C.c_int, [_p]),
    ("mp_window_stats_end", C.c_int, [_p, _p, _p]),
    ("mp_eval_launch_alt", C.c_int, [_p, _p]),
    ("mp_eval_sync", C.c_int, [_p]),
    ("mp_eval_timing", C.c_int, [_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    ("mp_eval_timing_samples", C.c_int, [_p, C.c_int32, _p, C.POINTER(C.c_int32)]),
    ("mp_eval_plan_info", C.c_int, [_p, _p]),
    ("mp_dimer_scan", C.c_int, [_p, C.c_int32, _p, _p, C.c_int32, C.c_int32, _p, _p, C.c_double, C.c_int64, _p,
                                C.POINTER(C.c_int64)]),
    ("mp_dimer_pairs", C.c_int, [_p, C.c_int32, _p, _p, C.c_int64, _p, _p, _p, C.c_double, _p]),
    ("mp_pair_coverage", C.c_int, [_p, C.c_int32, C.c_int32, _p, _p, C.c_int64, _p, _p]),
    ("mp_pcr_scan", C.c_int, [_p, _p, _p, C.c_int32, C.c_int32, _p, _p, _p]),
    ("mp_kmm_scan", C.c_int, [_p, _p, _p, C.c_int32, C.c_int32, _p, _p, C.c_int32, C.c_int32, C.c_int64, _p, C.POINTER(C.c_int64)]),
    ("mp_seq_load", C.c_int, [_p, _p, _p, C.c_int32]),
    ("mp_seq_free", C.c_int, [_p]),
    ("mp_seq_info", C.c_int, [_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("mp_pcr_scan_resident", C.c_int, [_p, C.c_int32, _p, _p, _p]),
    ("mp_kmm_scan_resident", C.c_int, [_p, C.c_int32, _p, _p, C.c_int32, C.c_int32, C.c_int64, _p, C.POINTER(C.c_int64)]),
    ("mp_comm_unique_id", C.c_int, [_p]),
    ("mp_comm_init", C.c_int, [_p, C.c_int32, C.c_int32, _p]),
    ("mp_comm_destroy", C.c_int, [_p]),
    ("mp_comm_describe", C.c_int, [_p, _p, C.c_char_p, C.c_int32]),
    ("mp_comm_allreduce_i64", C.c_int, [_p, _p, C.c_int64]),
    ("mp_comm_allreduce_host_i64", C.c_int, [_p, _p, C.c_int64]),
    ("mp_comm_allgather_i64", C.c_int, [_p, C.c_int64, _p]),
    ("mp_comm_allgatherv", C.c_int, [_p, _p, C.c_int64, _p, _p]),
    ("mp_comm_alltoall_counts", C.c_int, [_p, _p, _p]),
    ("mp_comm_alltoallv", C.c_int, [_p, _p, _p, _p, _p]),
    ("mp_eval_candidates_allreduce", C.c_int, [_p, C.c_int32, _p, _p, C.c_uint64, C.c_uint64, _p]),
    ("mp_device_bytes", C.c_int, [_p, C.POINTER(C.c_int64)]),
]
COMM_ID_BYTES = 128


def prefer_staged_copies():
    """For the drop-in command lines, called before anything starts the HIP runtime: read-backs into ordinary numpy arrays go
    through the runtime's staging buffers instead of page-locking the array for one use (GPU_PINNED_MIN_XFER_SIZE = 256 MiB: the
    58 MB of histogram entries of a 131072 x 1000 alignment take 6.5 ms instead of 27-32 ms, csrc/api.hip).  A process-wide runtime
    setting, hence a decision of the PROGRAM, not of the library: the class API leaves it alone.  A value the user exported, or
    MP_KEEP_PIN_THRESHOLD, wins."""
    if not os.environ.get("MP_KEEP_PIN_THRESHOLD"):
        os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "256")


def one_hip_runtime():
    """One HIP runtime per process.  torch ships its own libamdhip64 / libhsa-runtime64 (torch/lib); libmprime_hip.so is linked against
    the system's (/opt/rocm).  The file names differ but the SONAMEs agree, so whichever is loaded FIRST serves both — and when this
    library comes first and torch second, the loader finds torch's copy by its run path and the process ends up with two HIP and two
    HSA runtimes on one GPU: torch then reports "No HIP GPUs are available" (tools/maps_check.py), and what else two runtimes do to
    each other's registered memory is anybody's guess (the unexplained SIGABRT of round 4 is a candidate).  So before
    the HIP library is opened in a process where torch is installed but not imported yet, torch's runtime is opened first — exactly the
    state of a process that imported torch first (bench.py, the GPU suite).  MP_KEEP_SYSTEM_HIP=1 leaves the order to the caller."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("MP_KEEP_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    loaded = []
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                loaded.append((name, C.CDLL(path, mode=C.RTLD_GLOBAL)))
            except OSError as e:
                # [r6, advisor] a half-loaded pair (torch's HSA, the system's HIP) is the mixed state this function exists to prevent
                if loaded:
                    raise MprimeError(-2, f"torch's {loaded[0][0]} is loaded but its {name} is not ({e}): refusing to run on a mixed pair of HIP "
                                          "runtimes (MP_KEEP_SYSTEM_HIP=1 skips this step)") from None
                return
    # the runtime that now serves the process against the one libmprime_hip.so was built with (hipcc of /opt/rocm): a different MAJOR is refused
    if len(loaded) == 2:
        ver = C.c_int(0)
        try:
            if loaded[1][1].hipRuntimeGetVersion(C.byref(ver)) == 0:
                major = ver.value // 10000000
                built = int(os.environ.get("MP_BUILT_HIP_MAJOR", "7"))
                if os.environ.get("MP_TRACE"):
                    print(f"[mprime] HIP runtime bound: torch's ({libdir}), version {ver.value}; library built with HIP {built}.x", file=sys.stderr)
                if major != built:
                    raise MprimeError(-2, f"torch's HIP runtime is version {ver.value} (major {major}), libmprime_hip.so was built with HIP {built}.x: "
                                          "set MP_KEEP_SYSTEM_HIP=1 to run on the system's runtime instead")
        except AttributeError:
            pass


class MprimeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mprime error {code}: {msg}")
        self.code = code


class Library:
    """A shared library exporting the mprime C ABI."""

    def __init__(self, path: str | None = None):
        # Without an explicit path the library IS the product's backend: libmprime_hip.so, or another BUILD of it named by
        # MPRIME_LIBRARY (a debug / sanitizer build).  Whatever it is, it must say mp_backend_name() == "hip": no environment variable
        # can put the ABI checker behind a drop-in command (the CPU-only CLI tests patch this class from tests/checker_shim/).
        # Library(path) — what tests and tools do — loads what it is told to.
        implicit = path is None
        path = path or os.environ.get("MPRIME_LIBRARY") or HIP_LIB
        if not os.path.exists(path):
            raise MprimeError(-2, f"{path} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                  "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        self.path = path
        if os.path.basename(path).startswith("libmprime_hip"):
            one_hip_runtime()
        self.dll = C.CDLL(path)
        for name, res, args in SYMBOLS:
            fn = getattr(self.dll, name)     # AttributeError = ABI symbol missing
            fn.restype = res
            fn.argtypes = args
        self.backend = self.dll.mp_backend_name().decode()
        if implicit and self.backend != "hip":
            raise MprimeError(-2, f"{path} reports backend {self.backend!r}, not 'hip': the product runs on libmprime_hip.so only "
                                  "(MPRIME_LIBRARY may name another build of it, not the CPU checker)")

    def context(self, device: int = 0) -> "Context":
        return Context(self, device)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One opaque mp_ctx (one GPU)."""

    def __init__(self, lib: Library, device: int = 0):
        self.lib = lib
        self.d = lib.dll
        h = C.c_void_p()
        rc = self.d.mp_create(device, C.byref(h))
        if rc != 0:
            msg = self.d.mp_last_error(h).decode() if h else "mp_create failed (no usable GPU?)"
            raise MprimeError(rc, msg)
        self.h = h
        self.n_rows = 0
        self.n_win = 0
        self.k = 0

    def close(self):
        if self.h:
            self.d.mp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise MprimeError(rc, self.d.mp_last_error(self.h).decode())

    def set_stream(self, stream_handle: int):
        self._ck(self.d.mp_set_stream(self.h, C.c_void_p(stream_handle)))

    # (1)
    def reserve_columns(self, n_columns: int):
        """Row shards: the alignment is at least this wide even if no local row is (call before load_msa)."""
        self._ck(self.d.mp_reserve_columns(self.h, int(n_columns)))

    def load_msa(self, data: np.ndarray, row_off: np.ndarray):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        row_off = np.ascontiguousarray(row_off, dtype=np.int64)
        self.n_rows = len(row_off) - 1
        self._ck(self.d.mp_load_msa(self.h, _ptr(data), _ptr(row_off), self.n_rows))

    def row_attributes(self):
        lead = np.empty(self.n_rows, np.int32)
        rstrip = np.empty(self.n_rows, np.int32)
        rlen = np.empty(self.n_rows, np.int32)
        self._ck(self.d.mp_row_attributes(self.h, _ptr(lead), _ptr(rstrip), _ptr(rlen)))
        return lead, rstrip, rlen

    def row_histograms(self, n_bins: int):
        """(lead_hist, rstrip_hist) int64 [n_bins]: rows per leading-gap length / per right-stripped length."""
        lead = np.empty(n_bins, np.int64)
        rstrip = np.empty(n_bins, np.int64)
        self._ck(self.d.mp_row_histograms(self.h, n_bins, _ptr(lead), _ptr(rstrip)))
        return lead, rstrip

    # (2)
    def build_windows(self, p0: int, n_windows: int, k: int, v: int) -> int:
        n_ex = C.c_int32(0)
        self._ck(self.d.mp_build_windows(self.h, p0, n_windows, k, v, C.byref(n_ex)))
        self.n_win, self.k = n_windows, k
        return n_ex.value

    def get_exceptions(self, n_ex: int):
        w = np.empty(max(n_ex, 1), np.int32)
        r = np.empty(max(n_ex, 1), np.int32)
        codes = np.empty((max(n_ex, 1), self.k), np.uint8)
        self._ck(self.d.mp_get_exceptions(self.h, max(n_ex, 1), _ptr(w), _ptr(r), _ptr(codes)))
        return w[:n_ex], r[:n_ex], codes[:n_ex]

    def set_extra_rows(self, window: np.ndarray, words: np.ndarray):
        window = np.ascontiguousarray(window, dtype=np.int32)
        words = np.ascontiguousarray(words, dtype=word_dtype(self.k)).reshape(-1, 3)
        self._ck(self.d.mp_set_extra_rows(self.h, len(window), _ptr(window), _ptr(words)))

    def get_window_words(self, w: int, row0: int = 0, n: int | None = None) -> np.ndarray:
        n = self.n_rows - row0 if n is None else n
        out = np.empty((3, n), word_dtype(self.k))
        self._ck(self.d.mp_get_window_words(self.h, w, row0, n, _ptr(out)))
        return out

    # (3)
    def window_unique(self, want_labels: bool = False, cap_entries: int | None = None, sort: bool = True):
        """Returns (win_off[W+1], words[3][n], count[n], first_row[n]).  sort=True: the entries of every window sorted
        by first_row (the ABI leaves the order unspecified) plus the permutation needed to translate device labels;
        sort=False: as the library returned them (the native planning stage orders them itself)."""
        cap = cap_entries or max(1 << 16, min(self.n_rows * self.n_win, 1 << 24))
        n = C.c_int64(0)
        rc = self.d.mp_window_unique(self.h, cap, int(want_labels), C.byref(n))
        if rc == MP_ERR_CAPACITY:
            cap = int(n.value)
            rc = self.d.mp_window_unique(self.h, cap, int(want_labels), C.byref(n))
        self._ck(rc)
        n = int(n.value)
        off = np.empty(self.n_win + 1, np.int64)
        count = np.empty(max(n, 1), np.int32)
        first = np.empty(max(n, 1), np.int32)
        # the C side packs words as [3][n]; allocate exactly so the strides agree
        wbuf = np.empty(3 * max(n, 1), word_dtype(self.k))
        self._ck(self.d.mp_get_unique(self.h, _ptr(off), _ptr(wbuf), _ptr(count), _ptr(first)))
        words = wbuf[:3 * n].reshape(3, n) if n else np.zeros((3, 0), word_dtype(self.k))
        count, first = count[:n], first[:n]
        self._off = off
        if not sort:
            self._label_rank = None
            return off, words, count, first
        win_of = np.repeat(np.arange(self.n_win), np.diff(off))
        order = np.argsort((win_of.astype(np.int64) << 32) | first.astype(np.int64))     # (window, first row): all distinct
        self._label_rank = np.empty(n, np.int64)          # device index -> index within window, first-seen order
        self._label_rank[order] = np.arange(n) - off[win_of[order]]
        self._off = off
        return off, words[:, order], count[order], first[order]

    def window_unique_device(self, cap_entries: int | None = None) -> int:
        """The histograms only (mp_window_unique without labels): the entries stay on the device for host.Plan(device_context=...),
        which reads them back beside the planning.  Returns their number."""
        cap = cap_entries or max(1 << 16, min(self.n_rows * self.n_win, 1 << 24))
        n = C.c_int64(0)
        rc = self.d.mp_window_unique(self.h, cap, 0, C.byref(n))
        if rc == MP_ERR_CAPACITY:
            rc = self.d.mp_window_unique(self.h, int(n.value), 0, C.byref(n))
        self._ck(rc)
        return int(n.value)

    def set_entropy_gate(self, threshold: float):
        """Arm (threshold > 0) or disarm (0) the entropy gate on the device for the following window_unique_device calls."""
        self._ck(self.d.mp_set_entropy_gate(self.h, float(threshold)))

    def entropy_gate_result(self):
        """(number of windows the last histogram call rejected on the device, bool [n_win] which)."""
        n = C.c_int32(0)
        which = np.zeros(max(self.n_win, 1), np.uint8)
        self._ck(self.d.mp_entropy_gate_result(self.h, C.byref(n), _ptr(which)))
        return int(n.value), which[: self.n_win].astype(bool)

    def get_labels(self, w: int) -> np.ndarray:
        lab = np.empty(self.n_rows, np.int32)
        self._ck(self.d.mp_get_labels(self.h, w, _ptr(lab)))
        out = np.full(self.n_rows, -1, np.int64)
        ok = lab >= 0
        out[ok] = self._label_rank[self._off[w] + lab[ok]]
        return out

    def get_labels_raw(self, windows) -> np.ndarray:
        """[len(windows)][n_rows] int32: mp_get_labels of each window as the library returns them (index of the row's entry inside
        the window's segment of mp_get_unique, -1 = the row is not in the histogram)."""
        out = np.empty((max(len(windows), 1), self.n_rows), np.int32)
        wins = np.ascontiguousarray(windows, dtype=np.int32)
        self._ck(self.d.mp_get_labels_many(self.h, len(wins), _ptr(wins), _ptr(out)))
        return out[: len(windows)]

    # (4)
    def window_stats(self):
        """(freq [W][4][k], nn [W][k-1][4][4]) int64: per-window base counts and nearest-neighbour pair counts
        over the universe (state_matrix / trans_matrix, V20:541-577)."""
        freq = np.zeros((self.n_win, 4, self.k), np.int64)
        nn = np.zeros((self.n_win, self.k - 1, 4, 4), np.int64)
        self._ck(self.d.mp_window_stats(self.h, _ptr(freq), _ptr(nn)))
        return freq, nn

    def window_stats_begin(self):
        """window_stats in two halves (mp_window_stats_begin): the kernel and the read-back of its counters start on the context's second
        stream; returns the (freq, nn) arrays the counters will land in — valid after window_stats_end(freq, nn), or after a
        host.Plan(device_context=...) built with them (mp_plan_create_streamed ends a pending begin itself)."""
        freq = np.zeros((self.n_win, 4, self.k), np.int64)
        nn = np.zeros((self.n_win, self.k - 1, 4, 4), np.int64)
        self._ck(self.d.mp_window_stats_begin(self.h))
        return freq, nn

    def window_stats_end(self, freq, nn):
        self._ck(self.d.mp_window_stats_end(self.h, _ptr(freq), _ptr(nn)))

    def eval_candidates(self, cand_window, cand_codes, strictF: int, strictR: int) -> np.ndarray:
        cand_window = np.ascontiguousarray(cand_window, dtype=np.int32)
        cand_codes = np.ascontiguousarray(cand_codes, dtype=np.uint8).reshape(len(cand_window), self.k)
        out = np.zeros((len(cand_window), 3), np.int64)
        self._ck(self.d.mp_eval_candidates(self.h, len(cand_window), _ptr(cand_window), _ptr(cand_codes),
                                           strictF, strictR, _ptr(out)))
        return out

    # -- row shards (mprime.h section 9): RCCL behind the C ABI ---------------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        rc = self.d.mp_comm_unique_id(buf)
        if rc != 0:
            raise MprimeError(rc, "mp_comm_unique_id failed (librccl.so not loadable?)")
        return bytes(buf)

    def comm_init(self, n_ranks: int, rank: int, unique_id: bytes | None):
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id) if unique_id is not None else None
        self._ck(self.d.mp_comm_init(self.h, n_ranks, rank, buf))

    def comm_destroy(self):
        self._ck(self.d.mp_comm_destroy(self.h))

    def comm_describe(self):
        """(ranks the communicator reports, this rank as it reports it, path of the librccl in use or '')."""
        seen = np.zeros(2, np.int32)
        buf = C.create_string_buffer(512)
        self._ck(self.d.mp_comm_describe(self.h, _ptr(seen), buf, 512))
        return int(seen[0]), int(seen[1]), buf.value.decode()

    def comm_allreduce_device(self, device_ptr: int, n: int):
        self._ck(self.d.mp_comm_allreduce_i64(self.h, C.c_void_p(device_ptr), n))

    def comm_sum(self, a) -> np.ndarray:
        """Element-wise sum over the ranks of an int64 host array."""
        out = np.ascontiguousarray(a, dtype=np.int64).copy()
        self._ck(self.d.mp_comm_allreduce_host_i64(self.h, _ptr(out), out.size))
        return out

    def comm_gather_bytes(self, payload: np.ndarray, n_ranks: int):
        """(concatenation over ranks of a byte array whose length differs per rank, lengths per rank)."""
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        counts = np.zeros(n_ranks, np.int64)
        self._ck(self.d.mp_comm_allgather_i64(self.h, payload.size, _ptr(counts)))
        out = np.empty(int(counts.sum()), np.uint8)
        self._ck(self.d.mp_comm_allgatherv(self.h, _ptr(payload) if payload.size else None, payload.size, _ptr(counts),
                                           _ptr(out) if out.size else None))
        return out, counts

    def comm_exchange_bytes(self, payload: np.ndarray, send_counts):
        """Personalised exchange: `payload` holds send_counts[r] bytes for every rank r, in rank order.  Returns (the bytes this rank
        receives, source rank by source rank; their counts)."""
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        send_counts = np.ascontiguousarray(send_counts, dtype=np.int64)
        recv_counts = np.zeros(len(send_counts), np.int64)
        self._ck(self.d.mp_comm_alltoall_counts(self.h, _ptr(send_counts), _ptr(recv_counts)))
        out = np.empty(int(recv_counts.sum()), np.uint8)
        self._ck(self.d.mp_comm_alltoallv(self.h, _ptr(payload) if payload.size else None, _ptr(send_counts),
                                          _ptr(out) if out.size else None, _ptr(recv_counts)))
        return out, recv_counts

    def eval_candidates_allreduce(self, cand_window, cand_codes, strictF: int, strictR: int) -> np.ndarray:
        cand_window = np.ascontiguousarray(cand_window, dtype=np.int32)
        cand_codes = np.ascontiguousarray(cand_codes, dtype=np.uint8).reshape(len(cand_window), self.k)
        out = np.zeros((len(cand_window), 3), np.int64)
        self._ck(self.d.mp_eval_candidates_allreduce(self.h, len(cand_window), _ptr(cand_window), _ptr(cand_codes), strictF, strictR,
                                                     _ptr(out)))
        return out

    def eval_masks(self, cand_window, cand_codes, strictF: int, strictR: int):
        """(not_f, not_r): uint64 [n_cand][(n_rows+63)//64] — sequences a forward / reverse primer does not reach."""
        cand_window = np.ascontiguousarray(cand_window, dtype=np.int32)
        cand_codes = np.ascontiguousarray(cand_codes, dtype=np.uint8).reshape(len(cand_window), self.k)
        nw = (self.n_rows + 63) // 64
        nf = np.zeros((max(len(cand_window), 1), nw), np.uint64)
        nr = np.zeros((max(len(cand_window), 1), nw), np.uint64)
        self._ck(self.d.mp_eval_masks(self.h, len(cand_window), _ptr(cand_window), _ptr(cand_codes), strictF, strictR,
                                      _ptr(nf), _ptr(nr)))
        return nf[: len(cand_window)], nr[: len(cand_window)]

    def eval_masks_resident(self, cand_window, cand_codes, strictF: int, strictR: int):
        """The same masks, left in device memory (fetch with masks_fetch, combine with pair_coverage_resident)."""
        cand_window = np.ascontiguousarray(cand_window, dtype=np.int32)
        cand_codes = np.ascontiguousarray(cand_codes, dtype=np.uint8).reshape(len(cand_window), self.k)
        self._ck(self.d.mp_eval_masks_resident(self.h, len(cand_window), _ptr(cand_window), _ptr(cand_codes), strictF, strictR))
        self.n_masks = len(cand_window)

    def masks_set_bits(self, cand, row, which, value):
        cand = np.ascontiguousarray(cand, dtype=np.int32)
        row = np.ascontiguousarray(row, dtype=np.int32)
        which = np.ascontiguousarray(which, dtype=np.uint8)
        value = np.ascontiguousarray(value, dtype=np.uint8)
        self._ck(self.d.mp_masks_set_bits(self.h, len(cand), _ptr(cand), _ptr(row), _ptr(which), _ptr(value)))

    def masks_fetch(self):
        nw = (self.n_rows + 63) // 64
        nf = np.zeros((max(self.n_masks, 1), nw), np.uint64)
        nr = np.zeros((max(self.n_masks, 1), nw), np.uint64)
        if self.n_masks:
            self._ck(self.d.mp_masks_fetch(self.h, _ptr(nf), _ptr(nr)))
        return nf[: self.n_masks], nr[: self.n_masks]

    def pair_coverage_resident(self, pairs) -> np.ndarray:
        """popcount(not_f[i] | not_r[j]) over the resident masks for every (i, j) in pairs."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        out = np.zeros(max(len(pairs), 1), np.int32)
        self._ck(self.d.mp_pair_coverage_resident(self.h, len(pairs), _ptr(pairs), _ptr(out)))
        return out[: len(pairs)]

    def eval_upload(self, cand_window, cand_codes, strictF: int, strictR: int):
        cand_window = np.ascontiguousarray(cand_window, dtype=np.int32)
        cand_codes = np.ascontiguousarray(cand_codes, dtype=np.uint8).reshape(len(cand_window), self.k)
        self._ck(self.d.mp_eval_upload(self.h, len(cand_window), _ptr(cand_window), _ptr(cand_codes), strictF, strictR))

    def eval_launch(self, out_ptr: int):
        self._ck(self.d.mp_eval_launch(self.h, C.c_void_p(out_ptr)))

    def eval_launch_rotating(self, out_ptr: int, clear_ptr: int = 0):
        """Evaluate into the zeroed block at out_ptr; the same launch zeroes the block at clear_ptr (0: none) for the next one."""
        self._ck(self.d.mp_eval_launch_rotating(self.h, C.c_void_p(out_ptr), C.c_void_p(clear_ptr) if clear_ptr else None))

    def eval_launch_alt(self, out_ptr: int):
        """eval_launch on the context's second stream (mp_eval_launch_alt): alternate with eval_launch, different counter blocks."""
        self._ck(self.d.mp_eval_launch_alt(self.h, C.c_void_p(out_ptr)))

    def eval_sync(self):
        self._ck(self.d.mp_eval_sync(self.h))

    def eval_timing(self, reset: bool = False):
        ms, n = C.c_double(0), C.c_int32(0)
        self._ck(self.d.mp_eval_timing(self.h, int(reset), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def eval_plan_info(self) -> dict:
        """Which kernels the staged candidates go to (mp_eval_plan_info)."""
        info = np.zeros(4, np.int32)
        self._ck(self.d.mp_eval_plan_info(self.h, _ptr(info)))
        return {"chain_items": int(info[0]), "table_items": int(info[1]), "sliding_items": int(info[2]), "first_pass_chain_items": int(info[3])}

    def eval_timing_samples(self) -> np.ndarray:
        """Durations (ms) of the timed launches behind the last eval_timing() call."""
        n = C.c_int32(0)
        buf = np.zeros(4096, np.float32)
        self._ck(self.d.mp_eval_timing_samples(self.h, 4096, _ptr(buf), C.byref(n)))
        return buf[: min(n.value, 4096)].copy()

    # (5)
    def dimer_scan(self, codes: np.ndarray, off: np.ndarray, mode: int, n_new: int, loss_hit: np.ndarray,
                   dg_params: np.ndarray, dg_limit: float, cap: int = 1 << 16) -> np.ndarray:
        """Hit records [n][6] = (x, y, end length, end expansion, y expansion, idx), sorted."""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int32)
        loss_hit = np.ascontiguousarray(loss_hit, dtype=np.uint8)
        dg_params = np.ascontiguousarray(dg_params, dtype=np.float64)
        while True:
            hits = np.empty((max(cap, 1), 6), np.int32)
            n = C.c_int64(0)
            self._ck(self.d.mp_dimer_scan(self.h, len(off) - 1, _ptr(codes), _ptr(off), mode, n_new, _ptr(loss_hit),
                                          _ptr(dg_params), dg_limit, cap, _ptr(hits), C.byref(n)))
            if n.value <= cap:
                h = hits[: n.value]
                return h[np.lexsort((h[:, 1], h[:, 0]))] if len(h) else h
            cap = int(n.value)

    def dimer_any(self, codes, off, mode: int, n_new: int, loss_hit, dg_params, dg_limit: float) -> bool:
        """True iff the scan has at least one hit (no hit records fetched, no re-scan with a larger buffer)."""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int32)
        loss_hit = np.ascontiguousarray(loss_hit, dtype=np.uint8)
        dg_params = np.ascontiguousarray(dg_params, dtype=np.float64)
        n = C.c_int64(0)
        self._ck(self.d.mp_dimer_scan(self.h, len(off) - 1, _ptr(codes), _ptr(off), mode, n_new, _ptr(loss_hit), _ptr(dg_params),
                                      dg_limit, 0, None, C.byref(n)))
        return n.value > 0

    def dimer_pairs(self, codes, off, pairs, loss_hit, dg_params, dg_limit: float) -> np.ndarray:
        """flags[p] = 1 if the ordered pair (pairs[p,0] -> pairs[p,1]) forms a 3'-end dimer."""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int32)
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        loss_hit = np.ascontiguousarray(loss_hit, dtype=np.uint8)
        dg_params = np.ascontiguousarray(dg_params, dtype=np.float64)
        flags = np.zeros(max(len(pairs), 1), np.uint8)
        self._ck(self.d.mp_dimer_pairs(self.h, len(off) - 1, _ptr(codes), _ptr(off), len(pairs), _ptr(pairs), _ptr(loss_hit),
                                       _ptr(dg_params), dg_limit, _ptr(flags)))
        return flags[: len(pairs)]

    def pair_coverage(self, sets_a, sets_b, pairs) -> np.ndarray:
        """popcount(sets_a[i] | sets_b[j]) for every (i, j) in pairs; sets are rows of uint64 words."""
        sets_a = np.ascontiguousarray(sets_a, dtype=np.uint64)
        sets_b = np.ascontiguousarray(sets_b, dtype=np.uint64)
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        out = np.zeros(max(len(pairs), 1), np.int32)
        self._ck(self.d.mp_pair_coverage(self.h, sets_a.shape[0], sets_a.shape[1], _ptr(sets_a), _ptr(sets_b), len(pairs),
                                         _ptr(pairs), _ptr(out)))
        return out[: len(pairs)]

    def pcr_scan(self, data, row_off, codes, off) -> np.ndarray:
        """[n_pairs][n_rows][4] = (forward expansion, amplicon start, reverse expansion, position of RC(reverse)); -1 = none."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        row_off = np.ascontiguousarray(row_off, dtype=np.int64)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int32)
        n_rows, n_pairs = len(row_off) - 1, (len(off) - 1) // 2
        out = np.full((max(n_pairs, 1), max(n_rows, 1), 4), -1, np.int32)
        self._ck(self.d.mp_pcr_scan(self.h, _ptr(data), _ptr(row_off), n_rows, n_pairs, _ptr(codes), _ptr(off), _ptr(out)))
        return out[:n_pairs, :n_rows]

    def kmm_scan(self, data, row_off, pat_codes, pat_off, max_mismatch: int, term: int, cap: int = 1 << 20) -> np.ndarray:
        """Hits [n][4] = (sequence, start, pattern, strand) of the k-mismatch primer-site scan, sorted."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        row_off = np.ascontiguousarray(row_off, dtype=np.int64)
        pat_codes = np.ascontiguousarray(pat_codes, dtype=np.uint8)
        pat_off = np.ascontiguousarray(pat_off, dtype=np.int32)
        while True:
            hits = np.empty((max(cap, 1), 4), np.int32)
            n = C.c_int64(0)
            self._ck(self.d.mp_kmm_scan(self.h, _ptr(data), _ptr(row_off), len(row_off) - 1, len(pat_off) - 1, _ptr(pat_codes),
                                        _ptr(pat_off), int(max_mismatch), int(term), cap, _ptr(hits), C.byref(n)))
            if n.value <= cap:
                h = hits[: n.value]
                return h[np.lexsort((h[:, 3], h[:, 2], h[:, 1], h[:, 0]))] if len(h) else h
            cap = int(n.value)

    # (8b) the resident sequence store
    def seq_load(self, data, row_off):
        """The unaligned database of the PCR / k-mismatch scans, uploaded and packed once (mp_seq_load); the *_resident scans use it."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        row_off = np.ascontiguousarray(row_off, dtype=np.int64)
        self._ck(self.d.mp_seq_load(self.h, _ptr(data) if len(data) else None, _ptr(row_off), len(row_off) - 1))

    def seq_free(self):
        self._ck(self.d.mp_seq_free(self.h))

    def seq_info(self):
        """(sequences, bases, bytes on the device) of the store."""
        n, b, d = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        self._ck(self.d.mp_seq_info(self.h, C.byref(n), C.byref(b), C.byref(d)))
        return n.value, b.value, d.value

    def pcr_scan_resident(self, codes, off) -> np.ndarray:
        """pcr_scan on the stored database."""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int32)
        n_rows, n_pairs = self.seq_info()[0], (len(off) - 1) // 2
        out = np.full((max(n_pairs, 1), max(n_rows, 1), 4), -1, np.int32)
        self._ck(self.d.mp_pcr_scan_resident(self.h, n_pairs, _ptr(codes), _ptr(off), _ptr(out)))
        return out[:n_pairs, :n_rows]

    def kmm_scan_resident(self, pat_codes, pat_off, max_mismatch: int, term: int, cap: int = 1 << 20) -> np.ndarray:
        """kmm_scan on the stored database."""
        pat_codes = np.ascontiguousarray(pat_codes, dtype=np.uint8)
        pat_off = np.ascontiguousarray(pat_off, dtype=np.int32)
        while True:
            hits = np.empty((max(cap, 1), 4), np.int32)
            n = C.c_int64(0)
            self._ck(self.d.mp_kmm_scan_resident(self.h, len(pat_off) - 1, _ptr(pat_codes), _ptr(pat_off), int(max_mismatch), int(term), cap,
                                                 _ptr(hits), C.byref(n)))
            if n.value <= cap:
                h = hits[: n.value]
                return h[np.lexsort((h[:, 3], h[:, 2], h[:, 1], h[:, 0]))] if len(h) else h
            cap = int(n.value)

    def device_bytes(self) -> int:
        b = C.c_int64(0)
        self._ck(self.d.mp_device_bytes(self.h, C.byref(b)))
        return b.value
