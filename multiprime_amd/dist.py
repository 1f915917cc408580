"""Row-sharded execution over several GPUs of one node.

Every O(N) quantity of the core step is a sum over sequences, so the alignment's rows are
split into contiguous blocks, one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests):

  * candidate evaluation: each rank evaluates its rows; ONE all-reduce (sum, int64) of the
    [n_candidates x 3] counter block per alignment — a few hundred KB, latency-bound;
  * per-window histograms: (window, key, count, first row) entries are all-gathered as packed int64
    tensors (RCCL all_gather, no pickling) and merged by key (sum of counts, min of first row) by the
    native planning stage, identically on every rank — the two non-additive consumers (entropy,
    most-frequent seed with first-seen tie-break) need the global table;
  * row attributes (region quantiles) and the IUPAC exception list: all-gathered packed tensors.

The host control flow is replicated on every rank (it is deterministic given the merged
tables), rank 0 writes the files.  Results are identical for 1/2/4/8 shards: integer sums are
associative and the merged tables reproduce the single-process insertion order.

Transport.  On GPUs the collectives are the library's own (mprime.h section 9, csrc/comm.hip): RCCL calls on the context's stream,
the counters' all-reduce queued straight behind the evaluation kernel — `attach()` creates the communicator from an id rank 0
draws and this module broadcasts; torch.distributed then only carries that id and the barrier of the launcher.  With
MP_NATIVE_COMM=0, and in the CPU tests (gloo, the ABI checker has no collectives), the same exchanges go through torch.distributed.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


class RowShards:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.on_gpu = dist.get_backend(group) == "nccl"
        self.row0 = 0
        self.n_local = 0
        self.n_total = 0
        self.global_labels = None
        self.native = None          # the context whose RCCL communicator carries the collectives (attach)
        self.traffic = []           # (what, bytes sent, bytes received) per collective of this alignment; MP_TRACE prints them
        self.tables = {}            # what -> bytes of this rank's own table that went into a personalised exchange

    def _account(self, what, sent, received):
        import os
        import sys
        self.traffic.append((what, int(sent), int(received)))
        if os.environ.get("MP_TRACE"):
            print("[dist] rank %d %-28s sent %12d B  received %12d B" % (self.rank, what, sent, received), file=sys.stderr)

    def attach(self, ctx):
        """Collectives through the library's own communicator when this is a GPU run of the HIP library.  MP_NATIVE_COMM=0 keeps
        them in torch.distributed; MP_NATIVE_COMM=force takes the library's transport even when torch.distributed runs on gloo
        (the tests: several ranks on one GPU with MP_RCCL_LIBRARY pointing at tests/stub_rccl).
        Rank 0 draws the id; what it broadcasts is a status byte + the id, so that a rank 0 that cannot load RCCL makes EVERY rank
        raise (or, unforced, fall back to torch's transport) instead of leaving the others blocked in the broadcast."""
        import os
        mode = os.environ.get("MP_NATIVE_COMM", "1")
        forced = mode == "force"
        if ctx.lib.backend != "hip" or mode == "0" or self.group is not None or not (self.on_gpu or forced):
            return
        dev = self._device()
        raw, why = bytes(129), ""
        if self.rank == 0:
            try:
                raw = b"\x01" + ctx.comm_unique_id()
            except Exception as e:                                   # librccl not loadable, RCCL error: told to everyone below
                why = str(e)
        box = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        dist.broadcast(box, src=0, group=self.group)
        got = bytes(box.cpu().numpy().tobytes())
        if got[0] != 1:
            if forced or self.on_gpu and os.environ.get("MP_NATIVE_COMM_STRICT"):
                raise RuntimeError("rank 0 could not create an RCCL unique id" + (": " + why if why else ""))
            return                                                   # all ranks alike: torch.distributed carries the collectives
        ctx.comm_init(self.world, self.rank, got[1:])
        self.native = ctx

    # -- sharding ------------------------------------------------------------------------------
    def bounds(self, n_rows):
        base, rem = divmod(n_rows, self.world)
        starts = [r * base + min(r, rem) for r in range(self.world + 1)]
        return starts

    def window_range(self, n_windows):
        """[lo, hi): this rank's contiguous share of the windows (planning is split by windows, core.py)."""
        starts = self.bounds(n_windows)
        return starts[self.rank], starts[self.rank + 1]

    def take_shard(self, data, row_off):
        n = len(row_off) - 1
        starts = self.bounds(n)
        self.starts = starts
        a, b = starts[self.rank], starts[self.rank + 1]
        if b <= a:
            raise ValueError("fewer sequences than ranks")
        self.row0, self.n_local, self.n_total = a, b - a, n
        return data[row_off[a]:row_off[b]], row_off[a:b + 1] - row_off[a]

    # -- variable-length all-gather of packed arrays (RCCL on GPUs, gloo in the CPU tests) -------
    def _device(self):
        return torch.device("cuda", torch.cuda.current_device()) if self.on_gpu else torch.device("cpu")

    def gather_var(self, arr, axis=0):
        """Concatenation over ranks (rank order) of a numpy array whose length along `axis` differs per rank: one
        all_gather of the lengths, one all_gather of the byte-packed, padded payloads — tensors, no pickling."""
        arr = np.ascontiguousarray(np.moveaxis(np.asarray(arr), axis, 0))
        tail, dt = arr.shape[1:], arr.dtype
        row_bytes = int(np.prod(tail, dtype=np.int64)) * dt.itemsize
        if self.native is not None:
            flat, counts = self.native.comm_gather_bytes(arr.reshape(-1).view(np.uint8), self.world)
            n = int(counts.sum()) // max(row_bytes, 1)
            self._account("all-gather", arr.nbytes, int(counts.sum()))
            return np.moveaxis(flat.view(dt).reshape((n,) + tail), 0, axis)
        dev = self._device()
        n_local = torch.tensor([arr.shape[0]], dtype=torch.int64, device=dev)
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        dist.all_gather(sizes, n_local, group=self.group)
        sizes = [int(x.item()) for x in sizes]
        cap = max(max(sizes) * row_bytes, 1)
        payload = torch.zeros(cap, dtype=torch.uint8, device=dev)
        if arr.size:
            flat = torch.from_numpy(arr.reshape(-1).view(np.uint8).copy())
            payload[: flat.numel()] = flat.to(dev)
        parts = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(self.world)]
        dist.all_gather(parts, payload, group=self.group)
        out = [p[: n * row_bytes].cpu().numpy().view(dt).reshape((n,) + tail) for p, n in zip(parts, sizes)]
        self._account("all-gather", arr.nbytes, sum(sizes) * row_bytes)
        return np.moveaxis(np.concatenate(out, axis=0), 0, axis)

    def gather_many(self, arrays, with_counts=False):
        """gather_var of several arrays (rows along axis 0, any dtypes) with ONE pair of collectives for all of them: the per-rank row
        counts as a small matrix, the payloads as one byte string.  Returns the concatenations in rank order (and, with_counts, the
        [world][n_arrays] row counts)."""
        arrays = [np.ascontiguousarray(a) for a in arrays]
        if not arrays:
            return ([], np.zeros((self.world, 0), np.int64)) if with_counts else []
        rows = np.asarray([len(a) for a in arrays], np.int64).reshape(1, -1)
        all_rows = self.gather_var(rows)                                              # [world][n_arrays]
        payload = np.concatenate([a.reshape(-1).view(np.uint8) for a in arrays]) if any(a.size for a in arrays) else np.zeros(0, np.uint8)
        blob = self.gather_var(payload)
        out, at = [[] for _ in arrays], 0
        for r in range(self.world):
            for i, a in enumerate(arrays):
                tail = a.shape[1:]
                nb = int(all_rows[r, i]) * int(np.prod(tail, dtype=np.int64)) * a.dtype.itemsize
                out[i].append(blob[at:at + nb].view(a.dtype).reshape((int(all_rows[r, i]),) + tail))
                at += nb
        cat = [np.concatenate(parts, axis=0) for parts in out]
        return (cat, all_rows) if with_counts else cat

    def exchange_rows(self, arr, counts, what="exchange"):
        """Personalised exchange (all-to-all-v): `arr`'s rows are grouped by destination — counts[r] rows for rank r, in rank order.
        Returns the rows this rank receives, source rank by source rank.  A rank receives only what is meant for it; the all-gather
        of the same table would hand every rank all of it (world x the bytes)."""
        arr = np.ascontiguousarray(arr)
        tail, dt = arr.shape[1:], arr.dtype
        row_bytes = int(np.prod(tail, dtype=np.int64)) * dt.itemsize
        counts = np.asarray(counts, np.int64)
        assert len(counts) == self.world and int(counts.sum()) == len(arr)
        self.tables[what] = arr.nbytes
        if self.native is not None:
            flat, got = self.native.comm_exchange_bytes(arr.reshape(-1).view(np.uint8), counts * row_bytes)
            self._account(what, arr.nbytes - int(counts[self.rank]) * row_bytes, int(got.sum()) - int(got[self.rank]))
            return flat.view(dt).reshape((int(got.sum()) // max(row_bytes, 1),) + tail)
        dev = self._device()
        send_n = torch.from_numpy(counts.copy()).to(dev)
        recv_n = torch.empty(self.world, dtype=torch.int64, device=dev)
        dist.all_to_all_single(recv_n, send_n, group=self.group)
        recv_rows = [int(x) for x in recv_n.cpu().tolist()]
        send = torch.from_numpy(arr.reshape(-1).view(np.uint8).copy()).to(dev) if arr.size else torch.zeros(0, dtype=torch.uint8, device=dev)
        recv = torch.empty(sum(recv_rows) * row_bytes, dtype=torch.uint8, device=dev)
        dist.all_to_all_single(recv, send, [n * row_bytes for n in recv_rows], [int(n) * row_bytes for n in counts.tolist()], group=self.group)
        self._account(what, arr.nbytes - int(counts[self.rank]) * row_bytes, (sum(recv_rows) - recv_rows[self.rank]) * row_bytes)
        return recv.cpu().numpy().view(dt).reshape((sum(recv_rows),) + tail)

    def gather_columns(self, arr):
        """[m][n_local] per rank -> [m][n_total] (columns = rows of the alignment)."""
        return self.gather_var(arr, axis=1)

    def gather_dicts(self, d):
        """{k-mer string: [rows]} of every rank, rank order (JSON side files only: O(ids), small alignments)."""
        keys = list(d)
        lens = np.fromiter((len(d[s]) for s in keys), np.int64, len(keys))
        rows = np.fromiter((r for s in keys for r in d[s]), np.int64, int(lens.sum()))
        kb = np.frombuffer("".join(keys).encode("ascii"), np.uint8)
        klen = len(keys[0]) if keys else 0
        # per rank: number of keys, then the three packed arrays
        n_keys = self.gather_var(np.asarray([len(keys), klen], np.int64).reshape(1, 2))
        all_kb, all_lens, all_rows = self.gather_var(kb), self.gather_var(lens), self.gather_var(rows)
        out, ko, lo, ro = [], 0, 0, 0
        for nk, kl in n_keys.tolist():
            part = {}
            for i in range(nk):
                s = all_kb[ko + i * kl: ko + (i + 1) * kl].tobytes().decode("ascii")
                n = int(all_lens[lo + i])
                part[s] = all_rows[ro: ro + n].tolist()
                ro += n
            ko += nk * kl
            lo += nk
            out.append(part)
        return out

    # -- histograms and exceptions ---------------------------------------------------------------
    def gather_entries(self, e_window, words, count, first_global):
        """Every rank's (window, key words, count, first global row) entries, concatenated — for the REPLICATED planning (JSON side
        files: rank 0 needs every window's table).  No merge here: the native planning stage merges by key (counts add, the smallest
        first row wins).  The window-split planning uses entries_to_window_owners instead."""
        return self._unpack_entries(self.gather_var(self._pack_entries(e_window, words, count, first_global)))

    @staticmethod
    def _pack_entries(e_window, words, count, first_global):
        if np.asarray(words).dtype == np.uint64:                  # primers of more than 31 bases: 64-bit window words, one column each
            packed = np.empty((len(e_window), 6), np.int64)
            packed[:, 0] = e_window
            packed[:, 1:4] = np.asarray(words, np.uint64).view(np.int64).T
            packed[:, 4] = count
            packed[:, 5] = first_global
            return packed
        packed = np.empty((len(e_window), 5), np.int64)
        packed[:, 0] = e_window
        packed[:, 1] = np.asarray(words[0], np.int64) | (np.asarray(words[1], np.int64) << 32)
        packed[:, 2] = np.asarray(words[2], np.int64)
        packed[:, 3] = count
        packed[:, 4] = first_global
        return packed

    @staticmethod
    def _unpack_entries(g):
        if g.shape[1] == 6:
            return g[:, 0].astype(np.int32), np.ascontiguousarray(g[:, 1:4].T).view(np.uint64), g[:, 4].copy(), g[:, 5].copy()
        w = np.stack([(g[:, 1] & 0xFFFFFFFF).astype(np.uint32), (g[:, 1] >> 32).astype(np.uint32), g[:, 2].astype(np.uint32)])
        return g[:, 0].astype(np.int32), w, g[:, 3].copy(), g[:, 4].copy()

    def entries_to_window_owners(self, n_windows, e_window, words, count, first_global):
        """The planning is split by windows (window_range): every entry goes to the ONE rank that plans its window — a personalised
        exchange instead of gather_entries' all-gather, which moved world x the bytes for every rank to throw most of them away.
        `e_window` is ascending (window segments), so a destination's entries are one contiguous run."""
        starts = self.bounds(n_windows)
        cuts = np.searchsorted(np.asarray(e_window), np.asarray(starts), side="left")
        g = self.exchange_rows(self._pack_entries(e_window, words, count, first_global), np.diff(cuts), what="histogram entries")
        return self._unpack_entries(g)

    def exceptions_to_window_owners(self, n_windows, ex_w, x_row, ex_codes, k):
        """The IUPAC exception list the same way (one byte row per exception: window, row, codes)."""
        starts = self.bounds(n_windows)
        order = np.argsort(np.asarray(ex_w), kind="stable")                          # (the library hands them over by window already)
        ex_w, x_row, ex_codes = np.asarray(ex_w)[order], np.asarray(x_row)[order], np.asarray(ex_codes, np.uint8).reshape(len(order), k)[order]
        rec = np.empty((len(ex_w), 12 + k), np.uint8)
        rec[:, 0:4] = np.ascontiguousarray(ex_w, np.int32).view(np.uint8).reshape(-1, 4)
        rec[:, 4:12] = np.ascontiguousarray(x_row, np.int64).view(np.uint8).reshape(-1, 8)
        rec[:, 12:] = ex_codes
        cuts = np.searchsorted(ex_w, np.asarray(starts), side="left")
        g = self.exchange_rows(rec, np.diff(cuts), what="exception list")
        return (np.ascontiguousarray(g[:, 0:4]).view(np.int32).reshape(-1), np.ascontiguousarray(g[:, 4:12]).view(np.int64).reshape(-1),
                np.ascontiguousarray(g[:, 12:]))

    def gather_exceptions(self, ex_w, x_row, ex_codes, k):
        head = np.empty((len(ex_w), 2), np.int64)
        head[:, 0] = ex_w
        head[:, 1] = x_row
        g = self.gather_var(head)
        codes = self.gather_var(np.asarray(ex_codes, np.uint8).reshape(len(ex_w), k))
        return g[:, 0].astype(np.int32), g[:, 1].copy(), codes

    # -- evaluation ----------------------------------------------------------------------------
    def sum_many_int64(self, arrays):
        """sum_int64 of several arrays with ONE all-reduce (the per-window base and pair statistics travel together)."""
        shapes = [np.shape(a) for a in arrays]
        flat = np.concatenate([np.asarray(a, np.int64).reshape(-1) for a in arrays]) if arrays else np.zeros(0, np.int64)
        tot = self.sum_int64(flat)
        out, at = [], 0
        for sh in shapes:
            n = int(np.prod(sh, dtype=np.int64))
            out.append(tot[at:at + n].reshape(sh))
            at += n
        return out

    def sum_int64(self, a):
        """Element-wise sum over the ranks of an int64 array every rank holds (per-window statistics)."""
        self._account("all-reduce", np.asarray(a).size * 8, np.asarray(a).size * 8)
        if self.native is not None:
            return self.native.comm_sum(a).reshape(np.shape(a))
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64))
        if self.on_gpu:
            t = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def eval_allreduce(self, ctx, cand_window, cand_codes, sF, sR):
        """Local evaluation + the single all-reduce of the counter block."""
        n = len(cand_window)
        if self.native is not None:      # kernel, RCCL all-reduce and the copy back on one stream, inside the library
            return ctx.eval_candidates_allreduce(cand_window, cand_codes, sF, sR)
        if self.on_gpu:
            dev = torch.device("cuda", torch.cuda.current_device())
            out = torch.zeros((n, 3), dtype=torch.int64, device=dev)
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            ctx.eval_upload(cand_window, cand_codes, sF, sR)
            ctx.eval_launch(out.data_ptr())
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
            return out.cpu().numpy()
        out = torch.from_numpy(ctx.eval_candidates(cand_window, cand_codes, sF, sR))
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out.numpy()


class ShardGrid:
    """R row shards x G window groups of the `world` ranks of one job ("windows are independent, so an alternative / extra
    axis is window-sharding").  Rank r is row shard r // G of window group r % G: it loads the rows of ITS row shard (the alignment is
    replicated along the window axis) and works on ITS contiguous share of the windows only.  The G window groups are independent
    jobs of R ranks each — every collective of RowShards runs inside a row group (R ranks: ranks with the same r % G), over
    1 / G of the windows' tables and counters — and the leaders of the groups (row shard 0: ranks 0 .. G - 1) hand their output
    rows to rank 0.  R = world, G = 1 is the plain row sharding; R = 1, G = world needs no collective at all.

    Shape: `best_shape` — bench.py's `shard_shapes` block measures one rank's share of config 4 under 8x1, 4x2, 2x4 and 1x8 on one
    GPU (round 6: 32.3 / 30.9 / 30.9 / 27.8 us per step); window groups win whenever a GPU can hold every row, because a group
    of one row shard has no all-reduce to wait for and the sliding kernel keeps its long row slices."""

    def __init__(self, R, G):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        world = dist.get_world_size()
        if R < 1 or G < 1 or R * G != world:
            raise ValueError(f"grid {R}x{G} does not cover {world} ranks")
        self.R, self.G, self.world = R, G, world
        self.rank = dist.get_rank()
        self.ri, self.gi = divmod(self.rank, G)
        self.row_group = None
        if R > 1 and G > 1:
            for g in range(G):                                  # every rank creates every group, in the same order
                grp = dist.new_group([r * G + g for r in range(R)])
                if g == self.gi:
                    self.row_group = grp
        # the RowShards of this rank's row group (None: a group of one rank computes alone)
        self.comm = RowShards(group=self.row_group) if R > 1 else None

    @staticmethod
    def best_shape(world, n_rows, n_cols, replica_budget_bytes=None):
        """(R, G): as many window groups as the device memory allows — a rank of shape R x G holds n_rows / R rows at ~6 bytes per
        cell (planes, column planes, window tables: DESIGN.md section 3 measures 5.1 GB for 10^6 x 1000)."""
        import os
        budget = replica_budget_bytes if replica_budget_bytes is not None else int(float(os.environ.get("MP_GRID_REPLICA_GB", "96")) * (1 << 30))
        for G in sorted((g for g in range(1, world + 1) if world % g == 0), reverse=True):
            R = world // G
            if (n_rows + R - 1) // R * n_cols * 6 <= budget:
                return R, G
        return world, 1

    @staticmethod
    def parse(text, world):
        """'RxG' -> (R, G), checked against the world size."""
        try:
            R, G = (int(x) for x in text.lower().split("x"))
        except ValueError:
            raise ValueError(f"grid {text!r}: expected RxG, e.g. 2x4") from None
        if R < 1 or G < 1 or R * G != world:
            raise ValueError(f"grid {text!r} does not cover {world} ranks")
        return R, G

    @property
    def leader(self):
        return self.ri == 0

    def window_part(self):
        return self.gi, self.G

    def gather_outputs(self, payload):
        """The leaders' payloads in window-group order on rank 0 (None elsewhere).  A handful of output rows and, with the JSON side
        files, their id lists: pickled objects (gather_object), once per alignment."""
        box = [None] * self.world if self.rank == 0 else None
        dist.gather_object(payload if self.leader else None, box, dst=0)
        return [box[g] for g in range(self.G)] if self.rank == 0 else None


class StepBuckets:
    """Bucketed all-reduce of per-step counter blocks over a ring of buffers (bench.py, batch pipelines).

    `bucket` consecutive steps write into one [bucket][n][3] int64 buffer that is reduced with ONE collective
    (xGMI rings are latency-bound at a few hundred KB), asynchronously on the backend's stream, while the next
    buckets' steps fill the other buffers.  Every step's block is reduced; `drain()` completes everything.

    Two ways to take a step's block:
      begin_step()      the block; the caller's launch clears it itself (mp_eval_launch)
      begin_rotating()  (block, next block): the block holds zeros already, and the launch that fills it also clears
                        the block of the step after it (mp_eval_launch_rotating) — no fill dispatch between two
                        evaluations.  The next block's buffer must be free of any collective before the launch is
                        enqueued, so the ring is `depth` = 3 buffers deep when there is a collective (the reduction of
                        bucket j has the whole of buckets j + 1 and j + 2, less one step, to complete) and 2 otherwise."""

    def __init__(self, n, bucket, device, world=1, group=None, depth=None):
        self.B = max(1, int(bucket))
        self.D = max(2, int(depth)) if depth else (3 if world > 1 else 2)
        self.world, self.group = world, group
        self.buf = torch.zeros((self.D, self.B, n, 3), dtype=torch.int64, device=device)
        self.works = [None] * self.D
        self.i = 0                  # steps taken since the start, monotonic (a drain rounds it up to a bucket border)
        self.base = 0               # self.i at the start of the run the last drain() closed
        self._run0 = 0

    def _pos(self, i):
        return (i // self.B) % self.D, i % self.B

    def _free(self, b):
        if self.works[b] is not None:
            self.works[b].wait()            # stream-side: the current stream waits for the collective
            self.works[b] = None

    def begin_step(self):
        """The [n][3] block this step writes; waits (stream-side) for the reduction that last used its buffer."""
        b, slot = self._pos(self.i)
        if slot == 0:
            self._free(b)
        return self.buf[b, slot]

    def begin_rotating(self):
        """(this step's block — zeros —, the next step's block, which this step's launch has to clear)."""
        b, slot = self._pos(self.i)
        nb, nslot = self._pos(self.i + 1)
        if slot == 0:
            self._free(b)
        if nslot == 0:
            self._free(nb)
        return self.buf[b, slot], self.buf[nb, nslot]

    def end_step(self):
        b, slot = self._pos(self.i)
        self.i += 1
        if slot == self.B - 1:
            self._flush(b, self.B)

    def _flush(self, b, n_slots):
        if self.world > 1 and n_slots:
            self.works[b] = dist.all_reduce(self.buf[b, :n_slots], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def drain(self):
        """Reduce the last, partial bucket and wait for everything in flight.  The next run starts on a bucket border with a
        zeroed first block (so it may be a rotating one).  `n_done` = the steps really taken in the run this call closes (the slots
        that pad a partial bucket are not steps)."""
        taken = self.i
        if self.i % self.B:
            self._flush(self._pos(self.i)[0], self.i % self.B)
            self.i += self.B - self.i % self.B
        for b in range(self.D):
            self._free(b)
        self.base, self._run0 = self._run0, self.i
        self.n_done = taken - self.base
        b, slot = self._pos(self.i)
        self.buf[b, slot].zero_()           # what the last rotating launch cleared is the block after ITS step, not this one

    def block_of(self, step):
        """Reduced block of `step` (0-based inside the run the last drain closed; one of its last (depth - 1) * bucket steps)."""
        b, slot = self._pos(self.base + step)
        return self.buf[b, slot]
