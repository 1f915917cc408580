"""eval_chain_x_kernel (csrc/evalx.hpp, round 6): the bit-sliced nested-chain evaluation for primers of 32..63 bases and for v = 4, 5
— every staged candidate set of those cases — against the oracle, and against the row-per-lane kernels it replaces (MP_EVAL_NO_X):
refinement chains read in either direction, runs longer than 8 members (cut into several items), unrelated candidates (runs of
one), strict positions beyond bit 31, IUPAC / edge-gap / ragged rows (patch planes), every words-per-thread shape."""
import numpy as np
import pytest
import torch  # noqa: F401

from multiprime_amd import iupac
from test_hip_parity import both, fuzz_msa

pytestmark = pytest.mark.gpu


def chain_candidates(rng, rows_ascii, p0, W, k, per_window):
    """Per window: a refinement chain from a real k-mer of some row (more degenerate member by member), listed upwards or downwards,
    then unrelated candidates up to `per_window`."""
    cw, codes = [], []
    n, L = rows_ascii.shape
    for w in range(W):
        src = rows_ascii[int(rng.integers(0, n)), p0 + w:p0 + w + k]
        base = iupac.MASK_LUT[src].copy()
        base[(base == 0) | (base > 8) | ((base & (base - 1)) != 0)] = 1            # gaps / IUPAC codes of the source row -> A
        chain = [base.copy()]
        for _ in range(int(rng.integers(1, 12))):
            nxt = chain[-1].copy()
            nxt[int(rng.integers(0, k))] |= np.uint8(1 << int(rng.integers(0, 4)))
            chain.append(nxt)
        if rng.random() < 0.5:
            chain.reverse()
        cand = chain[:per_window]
        while len(cand) < per_window:
            r = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, size=k)]
            r[int(rng.integers(0, k))] = np.uint8(rng.integers(1, 16))
            cand.append(r)
        for c in cand:
            cw.append(w)
            codes.append(c)
    return np.asarray(cw, np.int32), np.asarray(codes, np.uint8)


CASES = [  # seed, n, L, ragged, k, v
    (1, 300, 150, False, 36, 1), (2, 257, 170, False, 45, 2), (3, 9000, 110, False, 63, 3), (4, 1000, 120, False, 32, 0),
    (5, 20000, 100, False, 40, 4), (6, 700, 140, False, 50, 5), (7, 300, 120, False, 18, 4), (8, 40000, 80, False, 22, 5),
    (9, 4200, 100, False, 31, 4), (10, 12000, 100, False, 33, 5),
]


@pytest.mark.parametrize("seed,n,L,ragged,k,v", CASES)
def test_chain_x_kernel_equals_oracle_and_row_kernels(hip_lib, oracle_lib, monkeypatch, seed, n, L, ragged, k, v):
    data, off, maxlen = fuzz_msa(seed, n, L, ragged, p_iupac=0.002)
    h, o = both(hip_lib, oracle_lib, data, off)
    try:
        W = (int(np.sort(np.diff(off))[n // 4]) if ragged else maxlen) - k - 2
        assert h.build_windows(2, W, k, v) == o.build_windows(2, W, k, v)
        # IUPAC rows expanded by the host, as the product does it
        ne = h.build_windows(2, W, k, v)
        ew, er, ec = h.get_exceptions(ne)
        raw = iupac.strings_of(iupac.SYMBOL_LUT[ec]) if ne else []
        xw, xk = [], []
        for w_, s in zip(ew.tolist(), raw):
            if s.count("-") <= v and iupac.degeneracy(s) <= 16:
                for e in iupac.expand(s):
                    xw.append(w_)
                    xk.append(e)
        if xw:
            words = iupac.words_of_kmers(np.frombuffer("".join(xk).encode(), np.uint8).reshape(len(xk), k))
            for c in (h, o):
                c.set_extra_rows(np.asarray(xw, np.int32), words)
        rng = np.random.default_rng(seed)
        rows_ascii = np.zeros((n, maxlen), np.uint8) + ord("-")
        for r in range(min(n, 400)):
            rows_ascii[r, : off[r + 1] - off[r]] = data[off[r]:off[r + 1]]
        rows_ascii = rows_ascii[: min(n, 400)]
        cw, codes = chain_candidates(rng, rows_ascii, 2, W, k, 11)
        sF = (1 << 1) | (1 << (k - 1)) | (1 << (k // 2))
        sR = (0b11 << (k - 3)) | 1
        want = o.eval_candidates(cw, codes, sF, sR)
        got = h.eval_candidates(cw, codes, sF, sR)
        assert np.array_equal(got, want)
        assert want[:, 0].sum() > 0 and (v == 0 or (want[:, 1] + want[:, 2]).sum() > 0)          # every counter moved
        monkeypatch.setenv("MP_EVAL_NO_X", "1")
        assert np.array_equal(h.eval_candidates(cw, codes, sF, sR), want)              # the row-per-lane kernels: the round-5 path
        monkeypatch.delenv("MP_EVAL_NO_X")
        for shape in ("0", "1", "2", "3"):
            monkeypatch.setenv("MP_EVAL_X_SHAPE", shape)
            assert np.array_equal(h.eval_candidates(cw, codes, sF, sR), want), shape
    finally:
        h.close()
        o.close()


@pytest.mark.parametrize("n,k,v", [(5000, 18, 1), (300000, 18, 1), (20000, 36, 2)])
def test_launches_on_the_second_stream_equal_the_first(hip_lib, oracle_lib, n, k, v):
    """mp_eval_launch_alt: evaluations of one staged set alternately on the context's two streams into different counter blocks, twice
    each — every block equals the oracle's counters; the first launch of a staged set must be mp_eval_launch; mp_window_stats_begin /
    _end (the second stream's other user) give mp_window_stats' tables."""
    from multiprime_amd._abi import MprimeError
    data, off, maxlen = fuzz_msa(40 + k, min(n, 20000), 90, False, p_iupac=0.0)
    if n > 20000:                                                   # deep: the sliding kernel's size
        reps = n // 20000
        rows = data.reshape(-1, 90)
        data = np.tile(rows, (reps, 1)).reshape(-1)
        off = np.arange(reps * rows.shape[0] + 1, dtype=np.int64) * 90
    h, o = both(hip_lib, oracle_lib, data, off)
    try:
        W = maxlen - k - 2
        assert h.build_windows(2, W, k, v) == o.build_windows(2, W, k, v)
        rng = np.random.default_rng(k)
        rows_ascii = data[: 400 * 90].reshape(400, 90)
        cw, codes = chain_candidates(rng, rows_ascii, 2, W, k, 8)
        sF, sR = (1 << 2) | (1 << 3), (1 << (k - 2)) | (1 << 2)
        want = o.eval_candidates(cw, codes, sF, sR)
        h.set_stream(torch.cuda.current_stream().cuda_stream)
        h.eval_upload(cw, codes, sF, sR)
        blocks = torch.zeros((4, len(cw), 3), dtype=torch.int64, device="cuda")
        with pytest.raises(MprimeError, match="first launch"):      # nothing has built the windows' patch planes yet: that is the first stream's job
            h.eval_launch_alt(blocks[1].data_ptr())
        fw, nw_ = o.window_stats()
        f0, n0 = h.window_stats_begin()
        h.window_stats_end(f0, n0)
        assert np.array_equal(f0, fw) and np.array_equal(n0, nw_)
        h.eval_launch(blocks[0].data_ptr())
        h.eval_launch_alt(blocks[1].data_ptr())
        h.eval_launch(blocks[2].data_ptr())
        h.eval_launch_alt(blocks[3].data_ptr())
        h.eval_launch(blocks[0].data_ptr())                        # a block is cleared by the launch that fills it
        h.eval_launch_alt(blocks[1].data_ptr())
        h.eval_sync()
        torch.cuda.synchronize()
        got = blocks.cpu().numpy()
        for i in range(4):
            assert np.array_equal(got[i], want), i
    finally:
        h.close()
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,L,ragged,k,v", [(3, 6000, 120, False, 18, 1), (4, 70000, 80, False, 18, 2), (5, 3000, 150, False, 36, 2), (6, 900, 100, False, 25, 0)])
def test_exception_records_fetched_on_demand(hip_lib, oracle_lib, seed, n, L, ragged, k, v):
    """The exception records mp_build_windows leaves on the device (ex_fetch: copied and sorted for whoever asks first) are the oracle's whether
    the histograms' device gate or mp_get_exceptions asks first, on a context that builds its windows several times; the statistics after
    the host's extra rows, in two halves and blocking."""
    data, off, maxlen = fuzz_msa(seed, n, L, ragged, p_iupac=0.004)
    h, o = both(hip_lib, oracle_lib, data, off)
    try:
        W = (int(np.sort(np.diff(off))[n // 4]) if ragged else maxlen) - k - 2
        for round_ in range(3):
            ne = h.build_windows(2, W, k, v)
            assert ne == o.build_windows(2, W, k, v) and ne > 0
            if round_ == 1:
                h.set_entropy_gate(3.6)                            # the device gate counts the exception rows per window: the histograms ask first
                h.window_unique_device()
                h.set_entropy_gate(0)
            want_ex = o.get_exceptions(ne)
            got_ex = h.get_exceptions(ne)
            for a, b in zip(got_ex, want_ex):
                assert np.array_equal(a, b)
            ew, er, ec = got_ex
            raw = iupac.strings_of(iupac.SYMBOL_LUT[ec])
            xw, xk = [], []
            for w_, s in zip(ew.tolist(), raw):
                if s.count("-") <= v and iupac.degeneracy(s) <= 16:
                    for e in iupac.expand(s):
                        xw.append(w_)
                        xk.append(e)
            assert xw
            words = iupac.words_of_kmers(np.frombuffer("".join(xk).encode(), np.uint8).reshape(len(xk), k))
            for c in (h, o):
                c.set_extra_rows(np.asarray(xw, np.int32), words)
            fw, nw_ = o.window_stats()
            if round_ == 2:
                f0, n0 = h.window_stats()
            else:
                f0, n0 = h.window_stats_begin()
                h.window_stats_end(f0, n0)
            assert np.array_equal(f0, fw) and np.array_equal(n0, nw_)
            f1, n1 = h.window_stats()
            assert np.array_equal(f1, fw) and np.array_equal(n1, nw_)
    finally:
        h.close()
        o.close()


@pytest.mark.gpu
def test_contexts_that_used_both_streams_close_and_reopen(hip_lib, oracle_lib):
    """[r6] mp_destroy released the library's second stream BEFORE the stages whose release waits for that stream (free_eval): a wait on a
    destroyed stream, one process in fifteen died at a context's close.  Forty short-lived contexts that used both streams (statistics in two
    halves, evaluations on the second stream), each closed while the next already exists; the last one still equals the oracle."""
    data, off, maxlen = fuzz_msa(77, 3000, 90, False, p_iupac=0.001)
    k, v = 18, 1
    W = maxlen - k - 2
    rng = np.random.default_rng(1)
    rows_ascii = data[: 400 * 90].reshape(400, 90)
    cw, codes = chain_candidates(rng, rows_ascii, 2, W, k, 8)
    sF, sR = (1 << 2) | (1 << 3), (1 << (k - 2)) | (1 << 2)
    prev = None
    for i in range(40):
        h = hip_lib.context(0)
        h.load_msa(data, off)
        h.build_windows(2, W, k, v)
        h.set_stream(torch.cuda.current_stream().cuda_stream)
        f0, n0 = h.window_stats_begin()
        h.window_stats_end(f0, n0)
        h.eval_upload(cw, codes, sF, sR)
        blocks = torch.zeros((2, len(cw), 3), dtype=torch.int64, device="cuda")
        h.eval_launch(blocks[0].data_ptr())
        h.eval_launch_alt(blocks[1].data_ptr())
        h.eval_sync()
        torch.cuda.synchronize()
        if prev is not None:
            prev.close()
        prev = h
    o = oracle_lib.context(0)
    try:
        o.load_msa(data, off)
        o.build_windows(2, W, k, v)
        want = o.eval_candidates(cw, codes, sF, sR)
        got = blocks.cpu().numpy()
        assert np.array_equal(got[0], want) and np.array_equal(got[1], want)
    finally:
        prev.close()
        o.close()
