"""Host side of the MI355X-native core step: same class API, CLI, TSV and JSON files as
`scripts/multiPrime-core.py` (V20 = multiPrime-core_V20.py).

What runs where
  device (include/mprime.h, csrc/*.hip)   per-character mapping and packing (V20:453), row attributes (V20:625-627),
          the k-mer of every (window, sequence) with edge-gap repair (V20:666-687), the per-window k-mer histograms
          (V20:689-711), state_matrix / trans_matrix (V20:541-577) and every candidate x sequence mismatch
          evaluation (V20:1103-1130, 229-233);
  native host stage (include/mprime_host.h, csrc/hostplan.cpp, fasta.cpp — C++, threads over windows)
          FASTA records (V20:441-455), the insertion-ordered cover / gap dictionaries, gates, entropy, Viterbi and
          most-frequent seeds, the greedy degeneracy refinement and the replay of its stopping rules;
  this file   the sequencing of those calls, Tm / string filters of the few hundred output primers, and the
          byte-exact writers of the TSV and the two JSON side files.

One design change matters for the GPU: the reference interleaves refinement steps with coverage evaluations (one
mis_primer_check per step, V20:883-906).  The refinement sequence itself depends only on the nearest-neighbour
counts, not on the evaluations — those only decide where to stop — so the host stage first derives the whole
refinement chain of every window and seed, ALL chain members of ALL windows are evaluated in ONE batched launch,
and the reference's stopping rules are replayed on the results.  Outputs are identical; launches per alignment drop
from ~12 dependent rounds to one.
"""
from __future__ import annotations

import os
import sys
import threading
import time
from collections import defaultdict
from json.encoder import encode_basestring_ascii as _q

import numpy as np

from . import batchfilters, host, iupac, msa
from ._abi import MP_ERR_SHORT_WINDOW, Library, MprimeError

_B2I = {"A": 0, "C": 1, "G": 2, "T": 3}

HEADERS = ["Position", "Entropy of cover (bit)", "Entropy of total (bit)", "Optimal_primer",
           "primer_degenerate_number", "nonsense_primer_number", "Optimal_coverage", "Mis-F-coverage",
           "Mis-R-coverage", "Tm", "Information"]


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


_TRACE_PY = bool(os.environ.get("MP_TRACE_PY"))


class NN_degenerate(object):
    """Drop-in for the reference class of the same name (V20:342-345, 1133-1180)."""

    def __init__(self, seq_file, primer_length=18, coverage=0.8, number_of_dege_bases=18, score_of_dege_bases=1000,
                 product_len=250, position="2,-1", variation=2, raw_entropy_threshold=3.6, distance=4, GC="0.4,0.6",
                 nproc=10, outfile="", *, library: Library | None = None, device: int = 0, comm=None,
                 write_json: bool = True, write_bitsets: bool = False, keep_bitsets: bool = False, context=None, grid=None):
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
        self.keep_bitsets = keep_bitsets        # leave those bitsets on the device for a pairing stage in this process (pairing.Primers_filter(core=...))
        self.mask_index = {}
        self.comm = comm                        # multiprime_amd.dist.RowShards or None
        # multiprime_amd.dist.ShardGrid: R row shards x G window groups — this rank's row group is the RowShards (or None: one row shard),
        # this rank's window group a contiguous share of the windows; the group leaders' rows are gathered on rank 0 (run)
        self.grid = grid
        if grid is not None:
            if comm is not None:
                raise ValueError("pass either comm (row shards) or grid (row shards x window groups)")
            if (write_bitsets or keep_bitsets) and grid.G > 1:
                raise ValueError("coverage bitsets span every window: not with window groups (grid.G > 1)")
            self.comm = comm = grid.comm
        self._win_part = grid.window_part() if grid is not None else (0, 1)
        k = self.primer_length
        if not 2 <= k <= 63:
            # the reference has no such limit; documented in INTEGRATION.md (one k-mer = one machine word per plane, MP_MAX_K)
            print("Error: primer length must be in [2, 63] for this build (one window word per k-mer).")
            sys.exit(2)
        self.Y_strict, self.Y_strict_R = msa.strict_sets(position, k)
        self._sF = msa.strict_mask(self.Y_strict, k)
        self._sR = msa.strict_mask(self.Y_strict_R, k)
        self.stats = {}

        t0 = time.time()
        self.lib = library if library is not None else Library()      # raises if the HIP library is absent
        # the first device call of a process starts the HIP runtime (0.1-0.3 s): it runs beside the FASTA parser (both release the GIL)
        made = {}

        def make_context():
            try:
                # `context`: a context the caller keeps across alignments (the batch workers): loading a new alignment releases the
                # old one's arrays, the stream and the runtime's warmed-up copy path stay
                made["ctx"] = context if context is not None else self.lib.context(device)
            except BaseException as e:                                 # re-raised below, in the constructor's thread
                made["error"] = e

        starter = threading.Thread(target=make_context)
        starter.start()
        try:
            fa = host.Fasta(seq_file)                                  # native parser (parse_seq's record semantics)
            self._fasta = fa
            self.total_sequence_number = fa.n_rows
            if fa.n_rows == 0:
                raise ValueError("no sequence records in " + str(seq_file))
            # one process, the product library on both sides: the residue bytes go from the parsed file straight to the device
            # (mp_load_msa_fasta) — no 1 GB intermediate array for a 10^6-row alignment; row shards cut their rows out of it first
            streamed_load = comm is None and os.environ.get("MP_LOAD_STREAM", "1") != "0"
            data, row_off = (None, fa.row_offsets()) if streamed_load else fa.rows()
        finally:
            starter.join()
        if "error" in made:
            raise made["error"]
        self.ctx = made["ctx"]
        if comm is not None:
            comm.attach(self.ctx)                                      # GPUs: the library's own RCCL communicator (dist.py)
        self.stats["parse_s"] = time.time() - t0
        t0 = time.time()
        width = int(np.diff(row_off).max())                           # longest record (of ALL rows)
        if comm is not None:
            self.ctx.reserve_columns(width)                            # windows span the whole alignment, not this shard's rows
            data, row_off = comm.take_shard(data, row_off)
        if data is None and host.serves_device_library(self.lib) and hasattr(host.dll(), "mp_load_msa_fasta"):
            fa.load_into(self.ctx)
        else:
            if data is None:
                data, row_off = fa.rows()
            self.ctx.load_msa(data, row_off)
        # seq_attribute (V20:617-640) takes one order statistic of the rows' leading-gap lengths and one of their right-stripped
        # lengths: histograms of both leave the device (8 KB instead of 8 bytes per row), shards add theirs
        lead_h, rstrip_h = self.ctx.row_histograms(width + 1)
        if comm is not None:
            lead_h, rstrip_h = comm.sum_int64(lead_h), comm.sum_int64(rstrip_h)
        start, stop = msa.region_from_histograms(lead_h, rstrip_h, self.coverage)
        if stop - start < int(self.product):     # V20:635-638
            print("Error: max length of PCR product is shorter than the default min Product length with {} "
                  "coverage! Non candidate primers !!!".format(self.coverage))
            sys.exit(1)
        self.start_position, self.stop_position, self.length = start, stop, stop - start
        self.entropy_threshold = self._entropy_threshold(self.length)
        self.stats["load_s"] = time.time() - t0

    @property
    def seq_ids(self):
        return self._fasta.ids

    def _entropy_threshold(self, length):       # V20:642-649
        if length < 5000:
            return self.raw_entropy_threshold
        return self.raw_entropy_threshold * (0.95 if length < 10000 else 0.9)

    # ------------------------------------------------------------------ device tables -> native plan
    def _lap(self, label):
        """MP_TRACE_PY=1: host-side lap times of run() on stderr (milliseconds since the previous lap)."""
        if not _TRACE_PY:
            return
        now = time.time()
        print("[core] %-34s %8.3f ms" % (label, (now - getattr(self, "_lap_t", now)) * 1e3), file=sys.stderr)
        self._lap_t = now

    def _arm_device_gate(self, keep_tables=False):
        """The streamed planning lets the device reject the windows whose entropy exceeds the threshold for certain (mp_set_entropy_gate):
        their entries are neither read back nor planned.  Not when the caller wants every window's tables kept (a rejected window
        has none); MP_DEVICE_GATE=0 leaves every window to the host."""
        on = os.environ.get("MP_DEVICE_GATE", "1") != "0" and not keep_tables
        self.ctx.set_entropy_gate(self.entropy_threshold if on else 0)

    def _plan(self, keep_tables=None):
        """Device stage (windows, statistics, histograms) and the native per-window planning.  Returns the
        host.Plan, or None when the region holds no window."""
        k, v = self.primer_length, self.variation
        w_all = int(self.stop_position - self.start_position - k)              # windows = range(start, stop - k), V20:1141 ...
        g, G = self._win_part                                                   # ... of which this rank's window group takes a contiguous share
        lo, hi = (max(w_all, 0) * g // G, max(w_all, 0) * (g + 1) // G) if G > 1 else (0, w_all)
        self._p0 = p0 = int(self.start_position) + lo
        W = hi - lo
        self.n_windows = W
        if W <= 0:
            return None
        keep = self.write_json if keep_tables is None else keep_tables
        row_base = self.comm.row0 if self.comm is not None else 0
        t0 = time.time()
        self._lap("run: start")
        try:
            n_ex = self.ctx.build_windows(p0, W, k, v)
        except MprimeError as e:
            if e.code != MP_ERR_SHORT_WINDOW:
                raise
            # A row with fewer than k residues: V20:683-687 leaves its k-mer short and the reference dies with a ValueError in
            # Y_distance (V20:230) as soon as an affected window reaches mis_primer_check (tests/golden/short_row.json).
            # This build refuses the alignment up front: same exit status, a message instead of a traceback.
            print("Error: {}. The reference fails on such an alignment too (ValueError in Y_distance); remove sequences with "
                  "fewer than {} residues.".format(str(e).split(": ", 1)[-1], k))
            sys.exit(1)
        self._lap("build_windows (library)")
        # one rank without JSON side files on the HIP library: the histogram entries never come to Python — the planning stage reads them
        # back in bands of windows beside its own work (mp_plan_create_streamed; MP_PLAN_STREAM=0 keeps the two blocking calls)
        streamed = (self.comm is None and not self.write_json and self.lib.backend == "hip" and os.environ.get("MP_PLAN_STREAM", "1") != "0"
                    and host.serves_device_library(self.lib))
        def exceptions_on_the_host():
            """The window k-mers that hold an IUPAC code, and the concrete expansions of those with <= v gaps (V20:701-707) as window words:
            pure host work on the list mp_build_windows left in the context."""
            t_h = [time.time()]
            ex_w, ex_r, ex_codes = self.ctx.get_exceptions(n_ex)
            t_h.append(time.time())
            extra = None
            if n_ex:
                # rows with more than v gaps are gap_sequence entries, the others expand (selection, expansion and the windows of the
                # expansions in one native call  [r6: the numpy selection around mp_expand_kmer_words took 4.4 ms of this thread's 8 at 10^6 rows])
                x_win, words = host.expand_exception_words(ex_w, ex_codes, v)
                if len(x_win):
                    extra = (x_win, words)
            if _TRACE_PY:
                t_h.append(time.time())
                print("[core] exception list: " + " ".join("%.3f" % ((b - a) * 1e3) for a, b in zip(t_h, t_h[1:])) + " ms (get, expand)", file=sys.stderr)
            return ex_w, ex_r, ex_codes, extra

        early_unique = False
        if streamed and n_ex >= 2048:
            # The per-window histograms (mp_window_unique_device: ~0.7 ms of kernels at 131072 rows, 6 ms at 10^6) do not depend on the
            # exception list, and unpacking / expanding that list (1.6 ms / ~5 ms) does not touch the device: the device call runs on this
            # thread (ctypes drops the interpreter lock) while a helper thread does the host work.  Only mp_set_extra_rows — which does
            # touch the context — waits for both.
            box = {}

            def helper_main():
                try:
                    box["r"] = exceptions_on_the_host()
                except BaseException as e:                  # re-raised on the calling thread below
                    box["error"] = e

            helper = threading.Thread(target=helper_main)
            helper.start()
            t_u = time.time()
            try:
                self._arm_device_gate(keep)
                self.ctx.window_unique_device()
            finally:
                helper.join()
            self.stats["unique_s"] = time.time() - t_u
            early_unique = True
            if "error" in box:
                raise box["error"]
            ex_w, ex_r, ex_codes, extra = box["r"]
            self._lap("histograms || exception list (%d)" % n_ex)
        else:
            ex_w, ex_r, ex_codes, extra = exceptions_on_the_host()
            self._lap("get_exceptions + expand_kmer_words (%d)" % n_ex)
        if extra is not None:
            self.ctx.set_extra_rows(*extra)
            self._lap("set_extra_rows")
        self.stats["build_windows_s"] = time.time() - t0 - (self.stats["unique_s"] if early_unique else 0.0)
        t0 = time.time()
        # state_matrix / trans_matrix of every window (V20:541-577) straight from the column planes; shards add up.  The streamed planning
        # of a single process starts the kernel only (mp_window_stats_begin, second stream): its read-back of the histogram entries runs
        # beside it, and mp_plan_create_streamed collects the counters before a planner reads one
        # ([r6] tried: the column planes' half of the kernel queued right after mp_build_windows, beside the histograms — the planners start
        # 1.3 ms earlier, the histogram kernels end 1.7 ms later and the bands arrive later: no gain, not kept)
        stats_beside = streamed and self.comm is None and os.environ.get("MP_STATS_BESIDE", "1") != "0"
        if stats_beside:
            self._freq, self._nn = self.ctx.window_stats_begin()
        else:
            self._freq, self._nn = self.ctx.window_stats()
        if self.comm is not None:
            self._freq, self._nn = self.comm.sum_many_int64([self._freq, self._nn])           # one all-reduce for both tables
        self.stats["stats_s"] = time.time() - t0
        t0 = time.time()
        # per-window histograms; sorted by first row only when the id lists of the JSON files need the labels
        # (the Python JSON writer of the row-sharded path wants them sorted by first row; the native writer takes them as they come)
        # MP_JSON_WRITER=python keeps the Python writer in a single process too (tests compare the two byte for byte)
        self._native_json = (self.write_json and self.comm is None and self._win_part[1] == 1 and
                             os.environ.get("MP_JSON_WRITER", "native") != "python")
        x_row = ex_r.astype(np.int64) + row_base
        if streamed:
            if not early_unique:
                self._arm_device_gate(keep)
                self.ctx.window_unique_device()
                self.stats["unique_s"] = time.time() - t0
            self.ctx.set_entropy_gate(0)
            self.stats["windows_device_gated"] = self.ctx.entropy_gate_result()[0]
            t0 = time.time()
            self._exc = (ex_w, x_row, ex_codes)
            self._win_split = False
            self._dev_entries = None
            plan = host.Plan(k=k, v=v, n_windows=W, total_sequences=self.total_sequence_number, coverage=self.coverage,
                             entropy_threshold=self.entropy_threshold, max_degeneracy=self.score_of_dege_bases,
                             max_dege_positions=self.number_of_dege_bases, row_base=row_base, x_window=ex_w, x_row=x_row, x_codes=ex_codes,
                             freq=self._freq, nn=self._nn, keep_tables=keep, device_context=self.ctx)
            self.stats["plan_s"] = time.time() - t0
            return plan
        off, words, count, first = self.ctx.window_unique(want_labels=self.write_json, sort=self.write_json and not self._native_json)
        self.stats["unique_s"] = time.time() - t0
        t0 = time.time()
        self._dev_entries = (off, words)
        if self.comm is None:
            # one rank: the read-back goes to the planning stage as it stands (window segments, 32-bit counts and first rows)
            self._exc = (ex_w, x_row, ex_codes)
            self._win_split = False
            plan = host.Plan(k=k, v=v, n_windows=W, total_sequences=self.total_sequence_number, coverage=self.coverage,
                             entropy_threshold=self.entropy_threshold, max_degeneracy=self.score_of_dege_bases,
                             max_dege_positions=self.number_of_dege_bases, e_off=off, e_words=words, e_count=count, e_first=first,
                             row_base=row_base, x_window=ex_w, x_row=x_row, x_codes=ex_codes, freq=self._freq, nn=self._nn,
                             keep_tables=keep)
            self.stats["plan_s"] = time.time() - t0
            return plan
        e_window = np.repeat(np.arange(W, dtype=np.int32), np.diff(off))
        e_first = first.astype(np.int64) + row_base
        # Row shards without JSON side files split the planning by WINDOWS: every rank plans a contiguous share of them and needs the
        # histogram entries and exceptions of THOSE windows only, from every rank's rows — a personalised exchange (all-to-all-v: a rank
        # receives about what it holds itself; round 4 all-gathered every table to every rank, world x the bytes, and threw most of it
        # away).  The candidates are then gathered, evaluated by everyone on their own rows, and the results of the shares are
        # concatenated (run()).  The JSON writers need every window's ordered table on rank 0, so with them the planning stays
        # replicated on all-gathered tables (it is O(windows x sequences) output anyway).
        self._win_split = self.comm is not None and not self.write_json and not keep
        if self._win_split:
            e_window, words, count, e_first = self.comm.entries_to_window_owners(W, e_window, words, count, e_first)
            ex_local = (ex_w, x_row, ex_codes)
            ex_w, x_row, ex_codes = self.comm.exceptions_to_window_owners(W, ex_w, x_row, ex_codes, k)
            self._exc = ex_local                                 # this rank's own rows: what its coverage bitsets need
        else:
            if self.comm is not None:
                e_window, words, count, e_first = self.comm.gather_entries(e_window, words, count, e_first)
                ex_w, x_row, ex_codes = self.comm.gather_exceptions(ex_w, x_row, ex_codes, k)
            self._exc = (ex_w, x_row, ex_codes)
        plan = host.Plan(k=k, v=v, n_windows=W, total_sequences=self.total_sequence_number, coverage=self.coverage,
                         entropy_threshold=self.entropy_threshold, max_degeneracy=self.score_of_dege_bases,
                         max_dege_positions=self.number_of_dege_bases, e_window=e_window, e_words=words, e_count=count,
                         e_first=e_first, x_window=ex_w, x_row=x_row, x_codes=ex_codes, freq=self._freq, nn=self._nn,
                         keep_tables=keep)
        self.stats["plan_s"] = time.time() - t0
        return plan

    def _self_dimers(self, codes):
        """dimer_check (V20:487-503) for the primers given as an [n][k] matrix of symbol codes: one mp_dimer_pairs launch over the
        ordered pairs (i -> i)."""
        n = len(codes)
        if n == 0:
            return []
        from .dimer import cached_loss_table, dg_limit, dg_params
        k = codes.shape[1]
        off = np.arange(n + 1, dtype=np.int32) * k
        pairs = np.repeat(np.arange(n, dtype=np.int32), 2).reshape(-1, 2)
        flags = self.ctx.dimer_pairs(np.ascontiguousarray(codes, np.uint8).reshape(-1), off, pairs, cached_loss_table(3.0), dg_params(), dg_limit())
        return flags.astype(bool).tolist()

    # ------------------------------------------------------------------ driver
    def run(self):
        """V20:1133-1180.  Helper threads of a run (self-dimer launch, coverage bitsets) use the context: whatever way this call ends,
        they have ended before it returns — a caller's `finally: ctx.close()` must never destroy a context a helper still works on."""
        self._helpers = []
        try:
            return self._run()
        finally:
            for helper in self._helpers:
                helper.wait_quietly()
            self._helpers = []

    def _beside(self, fn, *args, beside=True):
        helper = _Beside(fn, *args, beside=beside)
        self._helpers.append(helper)
        return helper

    def _run(self):
        k = self.primer_length
        t_run = time.time()
        plan = self._plan()
        self.plan = plan
        rows_out, non_cov_out, gap_out = [], {}, {}
        n_cand = 0
        if plan is not None:
            self.stats["windows_planned"] = plan.n_planned     # passed the gap / entropy / composition gates (V20:713-740)
            n_cand = plan.n_candidates
            t0 = time.time()
            cand_w, codes = plan.candidates()
            if self._win_split:
                # every rank planned its share of the windows: all candidates (rank order = window order) go to everyone, each
                # rank counts them on its rows, one all-reduce; a rank replays the stopping rules on its own slice
                mine = len(cand_w)
                # candidate windows, their codes and the shares' window counts travel together (one pair of collectives)
                (all_w, all_codes, planned), per_rank = self.comm.gather_many(
                    [cand_w, codes.reshape(mine, k), np.asarray([plan.n_planned], np.int64)], with_counts=True)
                first = int(per_rank[: self.comm.rank, 0].sum())
                n_cand = len(all_w)
                ev_all = (self.comm.eval_allreduce(self.ctx, all_w, all_codes, self._sF, self._sR) if n_cand
                          else np.zeros((0, 3), np.int64))
                ev = np.ascontiguousarray(ev_all[first:first + mine])
                self.stats["windows_planned"] = int(planned.sum())
            elif n_cand:
                if self.comm is not None:
                    ev = self.comm.eval_allreduce(self.ctx, cand_w, codes, self._sF, self._sR)
                else:
                    ev = self.ctx.eval_candidates(cand_w, codes, self._sF, self._sR)
            else:
                ev = np.zeros((0, 3), np.int64)
            self.stats["eval_s"] = time.time() - t0
            self.stats["n_candidates"] = n_cand
            t0 = time.time()
            self._lap("plan + eval")
            plan.finish(ev)                      # replay of the stopping rules, NM / MM choice, nonsense counts
            self._lap("plan.finish")
            res = plan.results()
            if self._win_split:
                keys = list(res)                                                          # shares in rank order = window order;
                res = dict(zip(keys, self.comm.gather_many([res[key] for key in keys])))  # all columns in one pair of collectives
            primers = iupac.strings_of(iupac.SYMBOL_LUT[res["codes"]])
            # the 3'-end self-dimer test of every window's primer (dimer_check, V20:487-503) in ONE launch:
            # it is the ordered pair (x -> x) of the dimer scan with Loss >= 3 and the two-term deltaG
            self._lap("results + strings")
            # (the launch and its read-back on a helper thread — one process only: the call holds no interpreter lock — while this thread
            # turns the result columns into lists)
            dimer_out = []
            dimers = self._beside(lambda: dimer_out.append(self._self_dimers(res["codes"])), beside=self.comm is None)
            p0 = self._p0
            wins = res["window"].tolist()
            cbit, tbit = res["cbit"].tolist(), res["tbit"].tolist()
            cov, f_mis, r_mis = res["cov"].tolist(), res["f_mis"].tolist(), res["r_mis"].tolist()
            nonsense, n_dege = res["nonsense"].tolist(), res["n_dege"].tolist()
            # Tm (V20:849-852, 282-336) and the "Information" column (primer_pre_filter, V20:507-521, V20:911) of every primer at once,
            # value for value what thermo.tm / filters.pre_filter give per primer — of ALL windows' primers, the few a self-dimer
            # removes included: the numbers do not depend on the flags, and the flags are still on their way
            tm_list = batchfilters.tm_of_primers(res["codes"])
            info_list = batchfilters.information_of_primers(res["codes"], self.GC, self.distance)
            dimers.join()
            dimer_flag = dimer_out[0]
            self._lap("self dimers || lists, Tm, Information")
            # JSON side files: written natively for a single process (mp_plan_write_side_files); row shards gather id lists per
            # output window and use the Python writer below
            side = self._side_file_builder() if self.write_json and not self._native_json else None
            keep = [i for i, d in enumerate(dimer_flag) if not d]                 # V20:749
            kept_codes = res["codes"][keep] if keep else np.zeros((0, k), np.uint8)
            self._lap("lists")
            # the coverage bitsets of the output rows (a launch, the verdicts of the IUPAC rows on host threads, a patch launch: all native,
            # no interpreter lock held) run beside the Tm / Information columns and the TSV — the output rows are known from here on.
            # One process only, and only when nothing else of this thread touches the context meanwhile (the Python side-file builder does).
            bitsets = None
            if self.write_bitsets or self.keep_bitsets:
                out_wins = np.asarray([wins[i] for i in keep], np.int32)
                out_pos = [p0 + wins[i] for i in keep]
                bitsets = self._beside(self._resident_bitsets, out_wins, kept_codes, out_pos, beside=self.comm is None and side is None)
            for i, primer in enumerate(primers):
                if dimer_flag[i]:
                    continue
                tm_avg, info = tm_list[i], info_list[i]
                pos = p0 + wins[i]
                rows_out.append([pos, cbit[i], tbit[i], primer, n_dege[i], nonsense[i], cov[i], f_mis[i], r_mis[i], tm_avg, info])
                if side is not None:
                    non_cov_out[pos], gap_out[pos] = side(wins[i], primer)
            self._lap("rows_out")
            self.stats["finish_s"] = time.time() - t0
            if bitsets is not None and (self.write_json or self.write_bitsets):
                bitsets.join()                               # the side files and the bitset file read the context
                if self.write_bitsets:
                    t0 = time.time()
                    self._write_bitsets(rows_out)
                    self.stats["bitsets_s"] += time.time() - t0
        if self.grid is not None and self.grid.G > 1:
            # window groups: the leaders' rows (window order = group order) and side-file entries meet on rank 0, which writes
            parts = self.grid.gather_outputs((rows_out, non_cov_out, gap_out))
            if parts is not None:
                rows_out, non_cov_out, gap_out = [], {}, {}
                for part_rows, part_nc, part_gap in parts:
                    rows_out += part_rows
                    non_cov_out.update(part_nc)
                    gap_out.update(part_gap)
                t0 = time.time()
                self._write(rows_out, non_cov_out, gap_out)
                self.stats["write_s"] = time.time() - t0
        elif self.comm is None or self.comm.rank == 0:
            t0 = time.time()
            self._write(rows_out, non_cov_out, gap_out)
            self.stats["write_s"] = time.time() - t0
        if plan is not None and bitsets is not None:
            bitsets.join()
        self.stats["run_s"] = time.time() - t_run
        self.stats["n_windows"] = self.n_windows
        self.stats["n_rows"] = len(rows_out)

    # ------------------------------------------------------------------ JSON side files
    def _exceptions_by_window(self):
        """{window: [(global row, raw IUPAC k-mer)]} in ascending rows."""
        ex_w, x_row, ex_codes = self._exc
        out = defaultdict(list)
        if len(ex_w):
            raw = iupac.strings_of(iupac.SYMBOL_LUT[ex_codes])
            order = np.lexsort((x_row, ex_w))
            for i in order.tolist():
                out[int(ex_w[i])].append((int(x_row[i]), raw[i]))
        return out

    def _rows_by_kmer(self, w):
        """{k-mer string: rows ascending (global indices)} of the plain (non-exception) rows of window w, from the device
        histogram entries and the per-row labels; row shards are merged in rank order."""
        k = self.primer_length
        off, words = self._dev_entries
        a, b = int(off[w]), int(off[w + 1])
        strs = iupac.strings_of(iupac.kmers_of_words(words[:, a:b], k))
        lab = self.ctx.get_labels(w)
        order = np.argsort(lab, kind="stable")
        cnt = np.bincount(lab[lab >= 0], minlength=b - a)
        edge = np.concatenate(([0], np.cumsum(cnt))) + int((lab < 0).sum())
        base = self.comm.row0 if self.comm is not None else 0
        mine = {s: (order[edge[e]:edge[e + 1]] + base).tolist() for e, s in enumerate(strs)}
        if self.comm is None:
            return mine
        merged = {}
        for part in self.comm.gather_dicts(mine):
            for s, rows in part.items():
                merged.setdefault(s, []).extend(rows)
        return merged

    def _side_file_builder(self):
        """F/R non-coverage dicts of a window's final primer (V20:1107-1127) and its gap_seq_id (V20:698)."""
        v = self.variation
        ids = self.seq_ids
        exc = self._exceptions_by_window()
        plan = self.plan
        sym = iupac.SYMBOL_LUT

        def ids_by_kmer(w, rows_of, want, gap_rows):
            out = {s: list(rows_of.get(s, ())) for s in want}
            touched = set()
            for row, raw in exc.get(w, ()):
                if (raw.count("-") > v) != gap_rows:
                    continue
                for e in iupac.expand(raw):
                    if e in out:
                        out[e].append(row)
                        touched.add(e)
            for e in touched:
                out[e].sort()
            return {s: [ids[r] for r in rows] for s, rows in out.items()}

        def build(w, primer):
            codes = iupac.codes_of(primer)
            cover_codes = plan.window_table(w, 0)[0]
            cover_keys = iupac.strings_of(sym[cover_codes])
            gap_raw = iupac.strings_of(sym[plan.window_table(w, 1)[0]])
            # which observed k-mers the primer does not reach (V20:1107-1127), all k-mers of the window at once
            miss = (cover_codes & codes) == 0                                   # symbol not in the primer's set ('-' = 0 misses)
            nd = miss.sum(axis=1)
            pos = np.arange(len(codes))
            hit_f = (miss & (((self._sF >> pos) & 1) == 1)).any(axis=1)
            hit_r = (miss & (((self._sR >> pos) & 1) == 1)).any(axis=1)
            f_keys = [cover_keys[i] for i in np.nonzero((nd > 0) & ((nd > v) | hit_f))[0].tolist()]
            r_keys = [cover_keys[i] for i in np.nonzero((nd > 0) & ((nd > v) | hit_r))[0].tolist()]
            rows_of = self._rows_by_kmer(w)
            got = ids_by_kmer(w, rows_of, set(f_keys) | set(r_keys), False)
            non_cov = [{q: got[q] for q in f_keys}, {q: got[q] for q in r_keys}]
            gap_keys = {}
            for g in gap_raw:
                for e in iupac.expand(g):
                    gap_keys.setdefault(e)
            gids = ids_by_kmer(w, rows_of, set(gap_keys), True)
            return non_cov, {g: gids[g] for g in gap_keys}

        return build

    def _resident_bitsets(self, wins, codes, positions):
        """Per output window, which sequences a forward / reverse primer there does NOT reach — exactly the union the
        pairing stage takes of gap_seq_id and non_coverage_seq_id (get_multiPrime_V8.py:560-567) — as bits, one per
        sequence, computed by one mp_eval_masks_resident launch and LEFT ON THE DEVICE (mask i = output row i);
        the rows whose window held an IUPAC code are decided here on the host and patched in (mp_masks_set_bits)."""
        k, v = self.primer_length, self.variation
        n_out = len(positions)
        codes = np.ascontiguousarray(codes, np.uint8).reshape(n_out, k)      # the output primers as 4-bit base sets (what MASK_LUT gives of their strings)
        t_all = t0 = time.time()
        # the launch (a device call: no interpreter lock held) on a thread of its own, the IUPAC rows' verdicts (native host code) on this one:
        # the two meet at mp_masks_set_bits  [r6: they ran one after the other, 1.2-1.5 + 2.3 ms at 10^6 rows]
        launch = _Beside(self.ctx.eval_masks_resident, wins, codes, self._sF, self._sR, beside=self.comm is None)
        try:
            self.mask_index = {int(pos): i for i, pos in enumerate(positions)}
            # rows whose window held an IUPAC code: every expansion must be reached (V20:701-707 puts the id under each
            # expansion's k-mer), gap-type ones are in gap_seq_id.  Vectorised over all (exception, expansion) pairs.
            ex_w, x_row, ex_codes = self._exc
            patch = None
            if len(ex_w) and n_out:
                # Verdict of an exception row = the OR over its expansions of "not perfectly matched and (more than v mismatches or a
                # mismatch at a strict position)" (V20:701-707 puts the id under every expansion's k-mer, V20:1107-1127).  The
                # expansions need not be listed for that: position j CAN mismatch when it is '-' or when some member of the row's
                # symbol lies outside the primer's; the expansion that takes a mismatching member wherever there is one has the most
                # mismatches, so the row is bad when that count exceeds v, or else when a strict position can mismatch at all.
                # Selection (output windows, this shard's rows), verdicts and the assignment layout in one native call  [r6: the numpy
                # selection and repeat / tile around the verdicts took 2.5 ms at 10^6 rows]
                row0 = self.comm.row0 if self.comm is not None else 0
                slot_of = np.full(self.n_windows, -1, np.int32)
                slot_of[wins] = np.arange(n_out, dtype=np.int32)
                got = host.exception_assignments(ex_w, x_row, ex_codes, slot_of, row0, self.ctx.n_rows, codes, v, self._sF, self._sR)
                self._lap("bitsets: exception verdicts (%d)" % (len(got[0]) // 2))
                if len(got[0]):
                    patch = got
        finally:
            launch.wait_quietly() if sys.exc_info()[0] is not None else launch.join()
        self.stats["bitsets_masks_s"] = time.time() - t0
        self._lap("bitsets: eval_masks_resident (beside the verdicts)")
        if patch is not None:
            self.ctx.masks_set_bits(*patch)
            self._lap("bitsets: masks_set_bits")
        self.stats["bitsets_s"] = time.time() - t_all

    def _write_bitsets(self, rows_out):
        """{out}.coverage_bitsets.npz: the resident masks fetched (and, with row shards, gathered bit by bit) into a file —
        the hand-off to a pairing stage in ANOTHER process.  O(W x N / 8) bytes."""
        n_out = len(rows_out)
        nf, nr = self.ctx.masks_fetch()
        n_local = self.ctx.n_rows
        if self.comm is not None:
            # row shards are not multiples of 64: concatenate bit by bit across ranks, then re-pack
            bits = [np.unpackbits(m.view(np.uint8), axis=1, bitorder="little")[:, :n_local] for m in (nf, nr)]
            bits = [self.comm.gather_columns(b) for b in bits]
            n_total = bits[0].shape[1] if n_out else self.total_sequence_number
            nw = (n_total + 63) // 64
            packed = []
            for b in bits:
                pad = np.zeros((n_out, nw * 64), np.uint8)
                pad[:, :b.shape[1]] = b
                packed.append(np.packbits(pad, axis=1, bitorder="little").view(np.uint64).reshape(n_out, nw))
        else:
            n_total = n_local
            packed = [nf, nr]
        if self.comm is not None and self.comm.rank != 0:
            return
        # ids as the raw bytes of the file + offsets (bitset_ids() decodes them): 10^6 Python strings cost more than the masks
        ids_bytes, ids_off = self._fasta.ids_raw()
        t0 = time.time()
        np.savez(self.outfile + ".coverage_bitsets.npz", positions=np.asarray([int(r[0]) for r in rows_out], np.int64),
                 not_f=packed[0], not_r=packed[1], n_seq=np.int64(n_total), ids_bytes=ids_bytes, ids_off=ids_off)
        self.stats["bitsets_save_s"] = time.time() - t0

    def _write(self, rows_out, non_cov_out, gap_out):
        with open(self.outfile, "w") as fo:                                        # V20:1148-1170
            fo.write("\t".join(HEADERS) + "\n")
            for row in rows_out:
                fo.write("\t".join(map(str, row)) + "\n")
        if self.write_json and getattr(self, "_native_json", False) and self.plan is not None:
            k = self.primer_length
            p0 = self._p0
            wins = np.asarray([int(r[0]) - p0 for r in rows_out], np.int32)
            codes = (iupac.MASK_LUT[np.frombuffer("".join(r[3] for r in rows_out).encode(), np.uint8)].reshape(len(rows_out), k)
                     if rows_out else np.zeros((0, k), np.uint8))
            off, words = self._dev_entries
            ex_w, x_row, ex_codes = self._exc
            ids_bytes, ids_off = self._fasta.ids_raw()
            # The writer takes one label per (output window, sequence) — 4 bytes each on the host, 3.6 GB for 900 windows of a 10^6-row
            # alignment — so it is fed runs of windows whose labels stay under 256 MB (MP_JSON_BATCH: windows per run, for the tests).
            n_out, n_rows = len(wins), max(int(self.ctx.n_rows), 1)
            run = max(1, (1 << 26) // n_rows)
            if os.environ.get("MP_JSON_BATCH"):
                run = max(1, int(os.environ["MP_JSON_BATCH"]))
            pos = [int(r[0]) for r in rows_out]
            for a in range(0, max(n_out, 1), run):
                b = min(n_out, a + run)
                part = (1 if a == 0 else 0) | (2 if b >= n_out else 0)
                self.plan.write_side_files(wins[a:b], pos[a:b], codes[a:b], self._sF, self._sR, off, words,
                                           self.ctx.get_labels_raw(wins[a:b]), ex_w, x_row, ex_codes, ids_bytes, ids_off,
                                           self.outfile + ".non_coverage_seq_id_json", self.outfile + ".gap_seq_id_json", part=part)
        elif self.write_json:
            with open(self.outfile + ".non_coverage_seq_id_json", "w") as fj:      # V20:1172-1173 json.dump(.., indent=4)
                _dump_side_file(non_cov_out, fj, True)
            with open(self.outfile + ".gap_seq_id_json", "w") as fg:               # V20:1175-1176
                _dump_side_file(gap_out, fg, False)


class _Beside:
    """fn(*args) on a helper thread (beside=True) or at once; join() waits and hands an exception of the helper to the caller."""

    def __init__(self, fn, *args, beside=True):
        self._error = None
        self._thread = None
        if not beside:
            fn(*args)
            return

        def main():
            try:
                fn(*args)
            except BaseException as e:          # noqa: BLE001 — re-raised on the calling thread
                self._error = e

        self._thread = threading.Thread(target=main)
        self._thread.start()

    def join(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._error is not None:
            e, self._error = self._error, None
            raise e

    def wait_quietly(self):
        """The helper has ended when this returns; its error, if any, is dropped (the caller is already leaving with its own)."""
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        self._error = None


def bitset_ids(z):
    """Sequence ids of a `.coverage_bitsets.npz` (np.load result), in bit order."""
    raw, off = z["ids_bytes"].tobytes(), z["ids_off"].tolist()
    return [raw[a:b].decode("utf-8", errors="surrogateescape") for a, b in zip(off[:-1], off[1:])]
