"""One rank of tests/test_comm_ranks.py: a process with its own HIP context on the box's single GPU, driving the n_ranks > 1 branch
of csrc/comm.hip through the C ABI (no torch in this process).  MP_RCCL_LIBRARY points at the shared-memory stand-in of
tests/stub_rccl.  argv: rank, n_ranks, id (hex), work directory.  Writes rank<r>.json = {"ok": true, ...} or the failure."""
import json
import os
import sys
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def payload_of(rank, n):
    """Rank r's bytes for the variable-length gather: a seeded pattern of length n."""
    return np.random.default_rng([77, rank]).integers(0, 256, size=n, dtype=np.uint8)


def bounds(n_rows, world):
    base, rem = divmod(n_rows, world)
    return [r * base + min(r, rem) for r in range(world + 1)]


def main():
    rank, world, uid, wd = int(sys.argv[1]), int(sys.argv[2]), bytes.fromhex(sys.argv[3]), sys.argv[4]
    res = {"ok": False, "rank": rank}
    try:
        from multiprime_amd._abi import Library
        lib = Library()
        assert lib.backend == "hip"
        ctx = lib.context(0)
        ctx.comm_init(world, rank, uid)
        seen = ctx.comm_describe()
        assert seen[0] == world and seen[1] == rank, seen
        assert os.path.basename(seen[2]) == "librccl_stub.so", seen
        res["library"] = seen[2]
        spec = json.load(open(os.path.join(wd, "spec.json")))
        # 1. lengths + payloads: skewed, empty, one byte, beyond one transport chunk (mp_comm_allgather_i64 + mp_comm_allgatherv's padded slots)
        for lens in spec["gathers"]:
            got, counts = ctx.comm_gather_bytes(payload_of(rank, lens[rank]), world)
            assert counts.tolist() == lens, (counts.tolist(), lens)
            want = np.concatenate([payload_of(r, lens[r]) for r in range(world)]) if sum(lens) else np.zeros(0, np.uint8)
            assert got.shape == want.shape and np.array_equal(got, want), ("gather", lens)
        # 1b. the personalised exchange (mp_comm_alltoall_counts + mp_comm_alltoallv): piece (src -> dst) is a seeded pattern of
        # spec["exchanges"][i][src][dst] bytes — empty pieces, one byte, skewed, beyond one transport chunk
        def piece(src, dst, n):
            return np.random.default_rng([79, src, dst]).integers(0, 256, size=n, dtype=np.uint8)
        for M in spec["exchanges"]:
            mine = [piece(rank, d, M[rank][d]) for d in range(world)]
            got, counts = ctx.comm_exchange_bytes(np.concatenate(mine) if sum(M[rank]) else np.zeros(0, np.uint8), M[rank])
            assert counts.tolist() == [M[s][rank] for s in range(world)], (counts.tolist(), M)
            want = [piece(s, rank, M[s][rank]) for s in range(world)]
            want = np.concatenate(want) if sum(len(x) for x in want) else np.zeros(0, np.uint8)
            assert got.shape == want.shape and np.array_equal(got, want), ("exchange", M)
        # 2. host all-reduce (mp_comm_allreduce_host_i64): per-window statistics travel this way
        for n in spec["sums"]:
            mine = np.random.default_rng([78, rank]).integers(-2 ** 40, 2 ** 40, size=n, dtype=np.int64)
            want = sum(np.random.default_rng([78, r]).integers(-2 ** 40, 2 ** 40, size=n, dtype=np.int64) for r in range(world))
            got = ctx.comm_sum(mine)
            assert np.array_equal(got, want if n else np.zeros(0, np.int64)), ("sum", n)
        # 3. the fused evaluation + all-reduce on this rank's row shard (mp_eval_candidates_allreduce) == the single-process counts
        z = np.load(os.path.join(wd, "eval_case.npz"))
        rows = z["rows"]
        b = bounds(rows.shape[0], world)
        shard = rows[b[rank]:b[rank + 1]]
        L = rows.shape[1]
        ctx.load_msa(np.ascontiguousarray(shard).reshape(-1), np.arange(shard.shape[0] + 1, dtype=np.int64) * L)
        k, v = int(z["k"]), int(z["v"])
        n_ex = ctx.build_windows(int(z["p0"]), int(z["W"]), k, v)
        if n_ex:
            from multiprime_amd import host
            ex_w, ex_r, ex_codes = ctx.get_exceptions(n_ex)
            sel = (ex_codes == 0).sum(axis=1) <= v
            if sel.any():
                words, src = host.expand_kmer_words(ex_codes[sel])
                ctx.set_extra_rows(ex_w[sel][src], words)
        for tag in ("nested", "unrelated"):
            got = ctx.eval_candidates_allreduce(z["cw_" + tag], z["codes_" + tag], int(z["sF"]), int(z["sR"]))
            assert np.array_equal(got, z["want_" + tag]), ("eval", tag, int((got != z["want_" + tag]).sum()))
        # 4. device all-reduce in place behind a kernel on the context's stream (mp_comm_allreduce_i64): the statistics tables
        freq, nn = ctx.window_stats()
        assert np.array_equal(ctx.comm_sum(freq), z["freq"]) and np.array_equal(ctx.comm_sum(nn), z["nn"])
        ctx.comm_destroy()
        ctx.close()
        res["ok"] = True
    except BaseException:
        res["error"] = traceback.format_exc()
    with open(os.path.join(wd, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
