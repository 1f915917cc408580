"""Row-sharded execution over several GPUs of one node.

Every O(N) quantity of the core step is a sum over sequences, so the alignment's rows are
split into contiguous blocks, one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in the CPU tests):

  * candidate evaluation: each rank evaluates its rows; ONE all-reduce (sum, int64) of the
    [n_candidates x 3] counter block per alignment — a few hundred KB, latency-bound;
  * per-window histograms: (key, count, first row) lists are all-gathered and merged by key
    (sum of counts, min of first row) identically on every rank — the two non-additive
    consumers (entropy, most-frequent seed with first-seen tie-break) need the global table;
  * row attributes (region quantiles): all-gathered ints.

The host control flow is replicated on every rank (it is deterministic given the merged
tables), rank 0 writes the files.  Results are identical for 1/2/4/8 shards: integer sums are
associative and the merged tables reproduce the single-process insertion order.
"""
from __future__ import annotations

from collections import defaultdict

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

    # -- sharding ------------------------------------------------------------------------------
    def bounds(self, n_rows):
        base, rem = divmod(n_rows, self.world)
        starts = [r * base + min(r, rem) for r in range(self.world + 1)]
        return starts

    def take_shard(self, data, row_off):
        n = len(row_off) - 1
        starts = self.bounds(n)
        self.starts = starts
        a, b = starts[self.rank], starts[self.rank + 1]
        if b <= a:
            raise ValueError("fewer sequences than ranks")
        self.row0, self.n_local, self.n_total = a, b - a, n
        return data[row_off[a]:row_off[b]], row_off[a:b + 1] - row_off[a]

    def _gather_objects(self, obj):
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def gather_rows(self, arr):
        return np.concatenate(self._gather_objects(np.asarray(arr)))

    # -- histograms ----------------------------------------------------------------------------
    def merge_tables(self, off, words, count, first, exc, W, labels=None):
        """All-gather the per-rank (window, key, count, first_row) entries and merge by key."""
        win_of = np.repeat(np.arange(W, dtype=np.int64), np.diff(off))
        mine = (win_of, words, count.astype(np.int64), first.astype(np.int64), dict(exc))
        parts = self._gather_objects(mine)
        win_all = np.concatenate([p[0] for p in parts])
        words_all = np.concatenate([p[1] for p in parts], axis=1)
        count_all = np.concatenate([p[2] for p in parts])
        first_all = np.concatenate([p[3] for p in parts])
        key = np.stack([win_all.astype(np.uint64), words_all[0].astype(np.uint64), words_all[1].astype(np.uint64),
                        words_all[2].astype(np.uint64)], axis=1)
        uniq, inv = np.unique(key, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        m = len(uniq)
        cnt = np.bincount(inv, weights=count_all, minlength=m).astype(np.int64)
        fst = np.full(m, np.iinfo(np.int64).max, np.int64)
        np.minimum.at(fst, inv, first_all)
        order = np.lexsort((fst, uniq[:, 0]))
        rank_of = np.empty(m, np.int64)
        rank_of[order] = np.arange(m)
        uw = uniq[order, 0].astype(np.int64)
        moff = np.zeros(W + 1, np.int64)
        np.cumsum(np.bincount(uw, minlength=W), out=moff[1:])
        mwords = np.stack([uniq[order, 1], uniq[order, 2], uniq[order, 3]]).astype(np.uint32)
        # position of this rank's local entries inside the merged table (for the id lists)
        lo = sum(len(p[0]) for p in parts[: self.rank])
        self._local_to_merged = rank_of[inv[lo: lo + len(win_of)]] - moff[win_of]
        self._local_off = off
        mexc = defaultdict(list)
        for p in parts:
            for w_, lst in p[4].items():
                mexc[w_].extend(lst)
        for lst in mexc.values():
            lst.sort()
        return moff, mwords, cnt[order], fst[order], mexc

    def gather_labels(self, ctx, W):
        """Per-row histogram labels of every window, translated to merged-table indices and
        gathered from all ranks: [W][n_total].  O(W x N) — only for the JSON side files."""
        local = np.empty((W, self.n_local), np.int64)
        for w in range(W):
            lab = ctx.get_labels(w)
            ok = lab >= 0
            row = np.full(self.n_local, -1, np.int64)
            row[ok] = self._local_to_merged[self._local_off[w] + lab[ok]]
            local[w] = row
        parts = self._gather_objects(local)
        self.global_labels = np.concatenate(parts, axis=1)

    def labels(self, w):
        return self.global_labels[w]

    # -- evaluation ----------------------------------------------------------------------------
    def sum_int64(self, a):
        """Element-wise sum over the ranks of an int64 array every rank holds (per-window statistics)."""
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64))
        if self.on_gpu:
            t = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def eval_allreduce(self, ctx, cand_window, cand_codes, sF, sR):
        """Local evaluation + the single all-reduce of the counter block."""
        n = len(cand_window)
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


class StepBuckets:
    """Bucketed, double-buffered all-reduce of per-step counter blocks (bench.py, batch pipelines).

    `bucket` consecutive steps write into one [bucket][n][3] int64 buffer that is reduced with ONE collective
    (xGMI rings are latency-bound at a few hundred KB), asynchronously on the backend's stream, while the next
    bucket's steps fill the other buffer.  Every step's block is reduced; `drain()` completes everything."""

    def __init__(self, n, bucket, device, world=1, group=None):
        self.B = max(1, int(bucket))
        self.world, self.group = world, group
        self.buf = torch.zeros((2, self.B, n, 3), dtype=torch.int64, device=device)
        self.works = [None, None]
        self.i = 0

    def begin_step(self):
        """The [n][3] block this step writes; waits (stream-side) for the reduction that last used its buffer."""
        b, slot = (self.i // self.B) & 1, self.i % self.B
        if slot == 0 and self.works[b] is not None:
            self.works[b].wait()
            self.works[b] = None
        return self.buf[b, slot]

    def end_step(self):
        b, slot = (self.i // self.B) & 1, self.i % self.B
        self.i += 1
        if slot == self.B - 1:
            self._flush(b, self.B)

    def _flush(self, b, n_slots):
        if self.world > 1 and n_slots:
            self.works[b] = dist.all_reduce(self.buf[b, :n_slots], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def drain(self):
        """Reduce the last, partial bucket and wait for everything in flight; the step counter restarts at 0."""
        if self.i % self.B:
            self._flush((self.i // self.B) & 1, self.i % self.B)
        for b in (0, 1):
            if self.works[b] is not None:
                self.works[b].wait()
                self.works[b] = None
        self.n_done, self.i = self.i, 0

    def block_of(self, step):
        """Reduced block of `step` (one of the last 2 * bucket steps before the last drain)."""
        return self.buf[(step // self.B) & 1, step % self.B]
