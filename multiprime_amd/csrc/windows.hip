// windows.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of
// include/mprime.h.  Window scan: which (window, sequence) k-mers are plain column slices, the patch list of the repaired ones, the IUPAC exception list (mp_build_windows).
#include <type_traits>

#include "common.hpp"
#include "workers.hpp"
#include <thread>
#include "winwords.hpp"

using namespace mp;

namespace {

// ----------------------------------------------------------------------------------------------
// (2) window k-mers with edge-gap repair (V20:666-687)
// ----------------------------------------------------------------------------------------------
// Rows whose k-mer is NOT the plain column slice (edge-gap repair, IUPAC, ragged end) are flagged per
// (window, 64-row word) in `excl` and collected, per window, in a compact patch list {row, window words}:
// the bit-sliced kernels skip them in the column planes and meet them again in the patch planes; the histograms take
// them from the list.  `excl` also carries the plain rows with more than v gaps in the window (outside every count,
// V20:689) and the padding rows, so the bit-sliced kernels need no gap bookkeeping of their own.
//
// Two kernels.  classify_kernel only tells plain slices from the rest — bit-parallel over the 32 windows starting in a
// chunk, no divergence; pass 0 writes `excl` and counts the slow pairs per window, pass 1 (offsets known) lists them.  repair_kernel then gives every slow pair a lane of its own for the
// line-by-line restatement of V20:668-687.  (Round 1 and the first round-2 version ran the repair inside the sliding
// loop: a quarter of the (wave, window) steps had one lane in the repair path and 63 waiting — 1.36 ms per pass at
// 131072 x 1000; see profiles/r02_bench_eval.txt.)
struct PatchOut {
    int pass;
    unsigned long long *excl;     // [W][Npad/64]
    int32_t *count;               // [W]
    const int32_t *off;           // [W+1]   (pass 1)
    int32_t *cursor;              // [W]     (pass 1)
    int32_t *rows;                // [n]     (pass 1) row of every slow pair, grouped by window
    int32_t *wins;                // [n]     (pass 1) its window
};

// thread = row, workgroup = 256 rows x the (up to) 32 windows that START in one 32-column chunk.  The classification is
// bit-parallel over those windows: with N = the row's 64-bit "holds a residue" word of chunk c and c+1 and M = its "holds an
// IUPAC code" word, a sliding OR over k columns (five shift-OR doublings) tells for all 32 start offsets at once whether a
// window touches an IUPAC code / holds any residue, and two more shifts whether it starts and ends on a residue — bit o of
// `fast` = the k-mer at offset o is the plain column slice (fast_words() of winwords.hpp, 32 windows per instruction).
// Only the gap count (> v gaps: outside every count, V20:689) is taken window by window (shift, mask, popcount).
// Per (wave, window) there remain two ballots that turn lane bits into the row-bit words of `excl` and the slow-pair lists.
constexpr int kCtrStride = 32;            // ints between two windows' global counters: one 128-byte line each
constexpr int kClsBlock = 1024;          // 16 waves = 1024 rows per workgroup: a window's 16 `excl` words leave as one 128-byte store

// WIDE (primers of 32..63 bases): the same bit tricks on the 96 columns of chunk c .. c + 2 (unsigned __int128).
template <bool WIDE>
__global__ __launch_bounds__(kClsBlock) void classify_kernel(const MsaArgs M, int p0, int n_win, int k, int v, PatchOut po) {
    typedef typename std::conditional<WIDE, unsigned __int128, unsigned long long>::type Cols;      // residue / IUPAC flags of the columns from chunk c on
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type W;
    // ballot words of (window offset, wave): the `excl` array is window-major, so the 32 x 16 words of a workgroup are
    // transposed through LDS and stored 16 consecutive words at a time (a lone 8-byte store per (wave, window) cost
    // 0.8 ms per pass at 131072 x 1000: 2 M scattered partial-line writes)
    __shared__ unsigned long long s_flag[32][kClsBlock / 64];
    __shared__ int s_slow[32], s_base[32];   // slow pairs of this workgroup per window offset / their first slot in the window's list
    if (threadIdx.x < 32) s_slow[threadIdx.x] = 0;
    __syncthreads();
    const int r = blockIdx.x * kClsBlock + threadIdx.x;
    const int n_rows = M.n_rows;
    const int c = (p0 >> 5) + blockIdx.y;                    // chunk of the window starts handled here
    const int o_lo = max(0, p0 - c * 32), o_hi = min(32, p0 + n_win - c * 32);      // start offsets [o_lo, o_hi) are windows
    const W kmask = kmask_of<W>(k);
    const size_t np = (size_t)M.n_pad;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in_pad = r < M.n_pad, real_row = r < n_rows;
    uint32_t fast = 0, flag = 0xFFFFFFFFu;                   // rows that do not exist: not plain, flagged
    if (real_row) {
        const int len = M.rlen[r];
        const uint32_t *P = M.planes + ((size_t)c * 4) * np + r;
        const uint32_t loA = P[0], loC = P[np], loG = P[2 * np], loT = P[3 * np];
        const uint32_t hiA = P[4 * np], hiC = P[5 * np], hiG = P[6 * np], hiT = P[7 * np];      // chunk c+1 exists: n_chunks is padded by two
        const uint32_t lo1 = loA | loC, lo2 = loG | loT, hi1 = hiA | hiC, hi2 = hiG | hiT;
        Cols N = (Cols)((unsigned long long)(lo1 | lo2) | ((unsigned long long)(hi1 | hi2) << 32));
        Cols Mu = (Cols)((unsigned long long)((loA & loC) | (loG & loT) | (lo1 & lo2)) |
                         ((unsigned long long)((hiA & hiC) | (hiG & hiT) | (hi1 & hi2)) << 32));
        if (WIDE) {                                                                     // chunk c+2 exists: n_chunks is padded by two
            const uint32_t xA = P[8 * np], xC = P[9 * np], xG = P[10 * np], xT = P[11 * np];
            const uint32_t x1 = xA | xC, x2 = xG | xT;
            N |= (Cols)(x1 | x2) << 64;
            Mu |= (Cols)((xA & xC) | (xG & xT) | (x1 & x2)) << 64;
        }
        // sliding OR over k columns: X_t covers t columns, t the largest power of two <= k; cover(o) = X_t(o) | X_t(o + k - t)
        auto cover = [k](Cols X) {
            int t = 1;
            while (2 * t <= k) { X |= X >> t; t *= 2; }
            return X | (X >> (k - t));
        };
        const uint32_t any_iupac = (uint32_t)cover(Mu), any_res = (uint32_t)cover(N);
        const uint32_t ends_ok = (uint32_t)N & (uint32_t)(N >> (k - 1));
        const int room = len - k - c * 32;                                             // offsets 0..room lie inside the row
        const uint32_t inrow = room < 0 ? 0u : (room >= 31 ? 0xFFFFFFFFu : ((2u << room) - 1u));
        fast = inrow & ~any_iupac & (~any_res | ends_ok);
        flag = ~fast;                                                                   // excl: not a plain slice, or more than v gaps
        if (po.pass == 0)
            for (int o = o_lo; o < o_hi; o++)
                if (popcw((W)(~(W)(N >> o) & kmask)) > v) flag |= 1u << o;
    }
    // The per-window counters are hot: ~5 x 10^5 (wave, window) steps hold a slow pair at 131072 x 1000.  Device-scope
    // atomics on one cache line serialise (~50 ns each, measured: 0.8 ms per pass with a counter per 4 bytes), so a workgroup
    // first adds up in LDS, issues ONE global atomic per window, and the global counters sit 128 bytes apart (kCtrStride).
    int my_pos[32];                          // pass 1: rank of this lane's slow pair inside the workgroup, per window offset
#pragma unroll
    for (int o = 0; o < 32; o++) {           // fixed trip count: my_pos stays in registers
        my_pos[o] = -1;
        if (o < o_lo || o >= o_hi) continue;
        const bool is_slow = real_row && !((fast >> o) & 1u);
        const unsigned long long slow = __ballot(is_slow);
        if (po.pass == 0) {
            const unsigned long long flg = __ballot((flag >> o) & 1u);
            if (lane == 0) {
                s_flag[o][wave] = flg;
                if (slow) atomicAdd(&s_slow[o], (int)__popcll(slow));
            }
        } else {
            int base = 0;
            if (slow) {
                const int leader = __ffsll((long long)slow) - 1;
                if (lane == leader) base = atomicAdd(&s_slow[o], (int)__popcll(slow));
                base = __shfl(base, leader);
            }
            my_pos[o] = is_slow ? base + (int)__popcll(slow & ((1ull << lane) - 1ull)) : -1;
        }
    }
    __syncthreads();
    if (threadIdx.x < 32 && threadIdx.x >= o_lo && threadIdx.x < o_hi && s_slow[threadIdx.x]) {
        const int w = c * 32 + threadIdx.x - p0;
        if (po.pass == 0) atomicAdd(&po.count[(size_t)w * kCtrStride], s_slow[threadIdx.x]);
        else s_base[threadIdx.x] = atomicAdd(&po.cursor[(size_t)w * kCtrStride], s_slow[threadIdx.x]);
    }
    if (po.pass != 0) {
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 32; o++) {
            if (my_pos[o] < 0) continue;
            const int w = c * 32 + o - p0;
            const int slot = po.off[w] + s_base[o] + my_pos[o];
            po.rows[slot] = r;
            po.wins[slot] = w;
        }
        return;
    }
    // 32 windows x 16 row words: thread t stores word (t % 16) of window offset (t / 16) — 16 lanes, 128 contiguous bytes
    const int o = threadIdx.x >> 4, j = threadIdx.x & 15;
    if (threadIdx.x < 32 * 16 && o >= o_lo && o < o_hi) {
        const size_t word = (size_t)blockIdx.x * (kClsBlock / 64) + j;
        if (word < np / 64) po.excl[(size_t)(c * 32 + o - p0) * (np / 64) + word] = s_flag[o][j];
    }
    (void)in_pad;
}

// thread = one slow (window, row) pair: V20:668-687 line by line (winwords.hpp).  Writes the pair's window words (SKIP when the
// k-mer holds an IUPAC code or the row is too short), appends IUPAC k-mers to the exception list, raises the short-row flag.
template <typename W>
__global__ __launch_bounds__(kBlock) void repair_kernel(const MsaArgs M, int p0, int k, int n, const int32_t *__restrict__ rows,
                                                        const int32_t *__restrict__ wins, W *__restrict__ words,
                                                        ExRec *__restrict__ ex, int *__restrict__ ex_count, int *__restrict__ err) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= n) return;
    const int r = rows[e], w = wins[e], p = p0 + w;
    const W kmask = kmask_of<W>(k);
    const size_t np = (size_t)M.n_pad;
    const PlaneWords<W> q = load_plane_words<W>(M.planes + ((size_t)(p >> 5) * 4) * np + r, np);
    W b0, b1, g;
    NibOf<W> buf;
    const int rc = slow_words<W>(M, r, p, k, kmask, M.rlen[r], q, b0, b1, g, buf);
    if (rc == 1) {
        const int idx = atomicAdd(ex_count, 1);
        ex[idx].win = w; ex[idx].row = r;                                               // capacity = n: cannot overflow
#pragma unroll
        for (int i = 0; i < 4; i++) ex[idx].q[i] = i < WordTraits<W>::kNibWords ? buf.q[i < WordTraits<W>::kNibWords ? i : 0] : 0ull;
        b0 = 0; b1 = 0; g = WordTraits<W>::kSkip | kmask;
    } else if (rc == 2) {
        atomicMax(err, 1);
        err[1] = w; err[2] = r;
        b0 = 0; b1 = 0; g = WordTraits<W>::kSkip | kmask;
    }
    words[3 * (size_t)e] = b0; words[3 * (size_t)e + 1] = b1; words[3 * (size_t)e + 2] = g;
}

// parity / debug: window words of rows [row0, row0 + n) of one window, derived on the fly
template <typename W>
__global__ __launch_bounds__(kBlock) void window_words_kernel(const MsaArgs M, int p, int k, int row0, int n, W *__restrict__ out) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    W b0, b1, g;
    FlyViewT<W>(M, p, k, kmask_of<W>(k)).load(row0 + i, b0, b1, g);
    out[i] = b0; out[(size_t)n + i] = b1; out[2 * (size_t)n + i] = g;
}


}  // namespace

namespace mp {

// The exception records of mp_build_windows on the host, by (window, row): waits for the copy the build queued, then a counting sort over
// the windows (positions only) and the rows inside each window's short run; the 40-byte records move once (std::sort on the records:
// 0.85 ms for 22 666 of them, on packed keys 0.78).  Thread-safe: the caller's helper thread and the histogram's host part both ask.
int ex_fetch(mp_ctx *c) {
    std::lock_guard<std::mutex> lock(c->ex_mu);
    if (!c->ex_pending) return MP_OK;
    const auto t0 = std::chrono::steady_clock::now();
    const int cnt = c->ex_pending, n_win = c->n_win;
    c->ex_pending = 0;
    c->ex_host.clear();
    const size_t bytes = sizeof(ExRec) * (size_t)cnt;
    hipError_t e = hipSetDevice(c->dev);
    // The landing buffer is plain host memory kept by the context (a registered one was not faster for these 7 MB — 0.9-1.1 ms against 0.65 —
    // and would be registered HERE, on a helper thread, beside the calling thread's launches).  The stream is the library's own: the first
    // one is busy with the histograms, and mp_build_windows waited for the kernels that wrote the records.
    if ((size_t)cnt > c->ex_raw.size()) c->ex_raw.resize((size_t)cnt + (size_t)cnt / 4);
    if (e == hipSuccess && !c->ex_stream) e = hipStreamCreateWithFlags(&c->ex_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpyAsync(c->ex_raw.data(), c->ex, bytes, hipMemcpyDeviceToHost, c->ex_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->ex_stream);
    if (e != hipSuccess) return fail(c, MP_ERR_DEVICE, "exception records: %s", hipGetErrorString(e));
    const auto t1 = std::chrono::steady_clock::now();
    const ExRec *rec = c->ex_raw.data();
    for (int i = 0; i < cnt; i++)
        if (rec[i].win < 0 || rec[i].win >= n_win) return fail(c, MP_ERR_DEVICE, "exception record %d names window %d", i, rec[i].win);
    std::vector<int32_t> first((size_t)n_win + 1, 0);
    for (int i = 0; i < cnt; i++) first[(size_t)rec[i].win + 1]++;
    for (int w = 0; w < n_win; w++) first[(size_t)w + 1] += first[(size_t)w];
    std::vector<uint32_t> order((size_t)cnt);
    {
        std::vector<int32_t> cur(first.begin(), first.end() - 1);
        for (int i = 0; i < cnt; i++) order[(size_t)cur[(size_t)rec[i].win]++] = (uint32_t)i;
    }
    // (10^6 rows with IUPAC codes at 1e-5: 1.8e5 records, 7 MB — the sorts of the windows' runs and the one move of the records
    // are spread over a few threads, each with a contiguous range of windows: 4.8 -> ~1 ms)
    std::vector<ExRec> sorted((size_t)cnt);
    auto part = [&](int w0, int w1) {
        for (int w = w0; w < w1; w++)
            std::sort(order.begin() + first[(size_t)w], order.begin() + first[(size_t)w + 1], [&](uint32_t a, uint32_t b) { return rec[a].row < rec[b].row; });
        for (size_t i = (size_t)first[(size_t)w0]; i < (size_t)first[(size_t)w1]; i++) sorted[i] = rec[order[i]];
    };
    const int n_thr = cnt >= 16384 ? std::max(1, std::min({16, (int)std::thread::hardware_concurrency(), cnt / 8192})) : 1;
    if (n_thr <= 1) part(0, n_win);
    else {
        run_on_threads(n_thr, [&](int t) {            // window ranges of about equal record counts
            const int w0 = (int)(std::lower_bound(first.begin(), first.end(), (int32_t)((long long)cnt * t / n_thr)) - first.begin());
            const int w1 = t + 1 == n_thr ? n_win : (int)(std::lower_bound(first.begin(), first.end(), (int32_t)((long long)cnt * (t + 1) / n_thr)) - first.begin());
            part(std::min(w0, n_win), std::min(w1, n_win));
        });
    }
    c->ex_host.swap(sorted);
    if (getenv("MP_TRACE")) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[mprime] exceptions: %d records; copied in %.3f ms, sorted in %.3f ms\n", cnt,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    return MP_OK;
}

}  // namespace mp

extern "C" {

int mp_build_windows(mp_ctx *c, int32_t p0, int32_t n_win, int32_t k, int32_t v, int32_t *n_exc) {
    if (!c) return MP_ERR_ARG;
    if (!c->planes) return fail(c, MP_ERR_ARG, "no alignment loaded");
    if (k < 2 || k > MP_MAX_K || n_win <= 0 || p0 < 0 || v < 0 || v >= k)
        return fail(c, MP_ERR_ARG, "bad window arguments (k=%d v=%d n_windows=%d)", k, v, n_win);
    if (p0 + n_win > c->max_len) return fail(c, MP_ERR_ARG, "windows run past the longest row");
    HIPCK(c, hipSetDevice(c->dev));
    Lap lap(c->stream);
    free_windows(c);
    lap("build_windows: free");
    c->p0 = p0; c->n_win = n_win; c->k = k; c->v = v;
    // free_windows() above released every array whose byte count depends on the word width — mp_device_bytes' accounting (dev_free
    // takes the count from the CURRENT wsz(c)) is only right if nothing sized with the old width outlives this line
    if (c->u_b0 || c->u_b1 || c->u_g || c->cand_n || c->patch_words || c->extra_words)
        return fail(c, MP_ERR_DEVICE, "internal: arrays of the previous word width are still allocated");
    c->wide = k > MP_NARROW_K;            // 64-bit window words (winwords.hpp); wsz() = uint32 units per word
    size_t np = (size_t)c->n_pad;
    int rc;
    // histogram keys: one packed u64 when 3k bits fit (unique.hip), else three u32 words compared through a representative row
    c->p64 = 3 * k <= 63 && !getenv("MP_WIN_NO_PACK");
    if ((rc = dev_alloc(c, &c->ex_count, 1))) return rc;
    if ((rc = dev_alloc(c, &c->err_flag, 4))) return rc;
    if ((rc = dev_alloc(c, &c->extra_off, (size_t)n_win + 1))) return rc;
    const size_t nw = np / 64;
    if ((rc = dev_alloc(c, &c->excl, (size_t)n_win * nw))) return rc;
    if ((rc = dev_alloc(c, &c->patch_count, (size_t)n_win * kCtrStride))) return rc;
    if ((rc = dev_alloc(c, &c->patch_off, (size_t)n_win + 1))) return rc;
    if ((rc = dev_alloc(c, &c->patch_cursor, (size_t)n_win * kCtrStride))) return rc;
    const dim3 grid((unsigned)((c->n_pad + kClsBlock - 1) / kClsBlock), (unsigned)(((p0 + n_win - 1) >> 5) - (p0 >> 5) + 1));   // y: chunks holding window starts
    const MsaArgs M = msa_args(c);
    const FillSeg init[5] = {{c->extra_off, sizeof(int32_t) * ((size_t)n_win + 1), 0u}, {c->ex_count, sizeof(int), 0u}, {c->err_flag, 4 * sizeof(int), 0u},
                             {c->excl, sizeof(unsigned long long) * (size_t)n_win * nw, 0u}, {c->patch_count, sizeof(int32_t) * (size_t)n_win * kCtrStride, 0u}};
    if ((rc = fill_segments(c, init, 5))) return rc;
    lap("build_windows: alloc+memset");
    hipLaunchKernelGGL(c->wide ? classify_kernel<true> : classify_kernel<false>, grid, dim3(kClsBlock), 0, c->stream, M, p0, n_win, k, v,
                       PatchOut{0, c->excl, c->patch_count, nullptr, nullptr, nullptr, nullptr});
    HIPCK(c, hipGetLastError());
    lap("build_windows: classify");
    // slow pairs per window -> offsets on the host, then the listing pass and the repair
    std::vector<int32_t> pc((size_t)n_win), po((size_t)n_win + 1, 0), padded((size_t)n_win * kCtrStride);
    HIPCK(c, hipMemcpyAsync(padded.data(), c->patch_count, sizeof(int32_t) * padded.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    lap("build_windows: counts d2h");
    for (int w = 0; w < n_win; w++) pc[(size_t)w] = padded[(size_t)w * kCtrStride];
    long long tot = 0;
    c->max_patch = 0;
    for (int w = 0; w < n_win; w++) {
        po[(size_t)w] = (int32_t)tot;
        tot += pc[(size_t)w];
        c->max_patch = std::max(c->max_patch, (int)pc[(size_t)w]);
    }
    if (tot > 0x7fffffffLL / 4) return fail(c, MP_ERR_NOMEM, "patch list too large (%lld rows)", tot);
    po[(size_t)n_win] = (int32_t)tot;
    c->n_patch = (int)tot;
    HIPCK(c, hipMemcpyAsync(c->patch_off, po.data(), sizeof(int32_t) * po.size(), hipMemcpyHostToDevice, c->stream));
    c->h_patch_off = po;
    c->h_extra_off.assign((size_t)n_win + 1, 0);
    c->pp_dirty = true; c->qp_dirty = true;
    c->ex_host.clear();
    if (n_exc) *n_exc = 0;
    lap("build_windows: counts+offsets");
    if (tot) {
        int32_t *d_wins = nullptr;
        if ((rc = dev_alloc(c, &c->patch_words, (size_t)3 * (size_t)tot * wsz(c)))) return rc;
        if ((rc = dev_alloc(c, &c->patch_rows, (size_t)tot))) return rc;
        if ((rc = dev_alloc(c, &d_wins, (size_t)tot))) return rc;
        if ((rc = dev_alloc(c, &c->ex, (size_t)tot))) { dev_free(c, &d_wins, (size_t)tot); return rc; }
        c->ex_cap = (int)tot;
        hipError_t e = hipMemsetAsync(c->patch_cursor, 0, sizeof(int32_t) * (size_t)n_win * kCtrStride, c->stream);
        int cnt = 0, errv[4] = {0, 0, 0, 0};
        if (e == hipSuccess) {
            hipLaunchKernelGGL(c->wide ? classify_kernel<true> : classify_kernel<false>, grid, dim3(kClsBlock), 0, c->stream, M, p0, n_win, k, v,
                               PatchOut{1, c->excl, c->patch_count, c->patch_off, c->patch_cursor, c->patch_rows, d_wins});
            if (c->wide)
                hipLaunchKernelGGL(repair_kernel<uint64_t>, dim3((unsigned)((tot + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, M, p0, k, (int)tot,
                                   (const int32_t *)c->patch_rows, (const int32_t *)d_wins, reinterpret_cast<uint64_t *>(c->patch_words), c->ex, c->ex_count,
                                   c->err_flag);
            else
                hipLaunchKernelGGL(repair_kernel<uint32_t>, dim3((unsigned)((tot + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, M, p0, k, (int)tot,
                                   (const int32_t *)c->patch_rows, (const int32_t *)d_wins, c->patch_words, c->ex, c->ex_count, c->err_flag);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&cnt, c->ex_count, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(errv, c->err_flag, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        lap("build_windows: list+repair");
        dev_free(c, &d_wins, (size_t)tot);
        lap("build_windows: scratch released");
        if (e != hipSuccess) return fail(c, MP_ERR_DEVICE, "mp_build_windows: %s", hipGetErrorString(e));
        if (errv[0])
            return fail(c, MP_ERR_SHORT_WINDOW, "row %d has fewer than %d residues at window %d", errv[2], k, p0 + errv[1]);
        // the records (40 bytes each) stay on the device until someone asks: ex_fetch() — the caller's helper thread (mp_get_exceptions) or
        // the histogram's host part, both beside the histogram kernels instead of in front of them
        if (cnt && !c->ex_stream) HIPCK(c, hipStreamCreateWithFlags(&c->ex_stream, hipStreamNonBlocking));     // (created on the calling thread)
        {
            std::lock_guard<std::mutex> lock(c->ex_mu);
            c->ex_pending = cnt;
        }
        if (n_exc) *n_exc = cnt;
    }
    return MP_OK;
}

int mp_get_exceptions(mp_ctx *c, int32_t cap, int32_t *ew, int32_t *er, uint8_t *codes) {
    if (!c) return MP_ERR_ARG;
    if (!c->excl) return fail(c, MP_ERR_ARG, "no windows built");
    if (int rc = ex_fetch(c)) return rc;
    int n = (int)c->ex_host.size();
    if (cap < n) return fail(c, MP_ERR_CAPACITY, "exception buffer too small: need %d", n);
    auto part = [&](int i0, int i1) {
        for (int i = i0; i < i1; i++) {
            const ExRec &e = c->ex_host[(size_t)i];
            ew[i] = e.win; er[i] = e.row;
            for (int j = 0; j < c->k; j++)
                codes[(size_t)i * c->k + j] = (uint8_t)((e.q[j >> 4] >> (4 * (j & 15))) & 15u);
        }
    };
    const int n_thr = n >= 16384 ? std::max(1, std::min({16, (int)std::thread::hardware_concurrency(), n / 8192})) : 1;
    if (n_thr <= 1) part(0, n);
    else {
        mp::run_on_threads(n_thr, [&](int t) { part((int)((long long)n * t / n_thr), (int)((long long)n * (t + 1) / n_thr)); });
    }
    return MP_OK;
}

int mp_set_extra_rows(mp_ctx *c, int32_t n, const int32_t *win, const void *words) {
    if (!c) return MP_ERR_ARG;
    if (!c->excl) return fail(c, MP_ERR_ARG, "no windows built");
    HIPCK(c, hipSetDevice(c->dev));
    dev_free(c, &c->extra_words, (size_t)3 * c->n_extra * wsz(c));
    c->n_extra = 0;
    std::vector<int32_t> off((size_t)c->n_win + 1, 0);
    for (int i = 0; i < n; i++) {
        if (win[i] < 0 || win[i] >= c->n_win || (i && win[i] < win[i - 1]))
            return fail(c, MP_ERR_ARG, "extra rows must be sorted by window and in range");
        off[(size_t)win[i] + 1]++;
    }
    for (int w = 0; w < c->n_win; w++) off[(size_t)w + 1] += off[(size_t)w];
    HIPCK(c, hipMemcpy(c->extra_off, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice));
    c->h_extra_off = off;
    c->pp_dirty = true; c->qp_dirty = true;
    if (n > 0) {
        int rc;
        if ((rc = dev_alloc(c, &c->extra_words, (size_t)3 * n * wsz(c)))) return rc;
        c->n_extra = n;
        HIPCK(c, hipMemcpy(c->extra_words, words, sizeof(uint32_t) * 3 * (size_t)n * wsz(c), hipMemcpyHostToDevice));
    }
    return MP_OK;
}

int mp_get_window_words(mp_ctx *c, int32_t w, int32_t row0, int32_t n, void *out) {
    if (!c) return MP_ERR_ARG;
    if (!c->excl) return fail(c, MP_ERR_ARG, "no windows built");
    if (w < 0 || w >= c->n_win || row0 < 0 || n < 0 || row0 + n > c->n_rows) return fail(c, MP_ERR_ARG, "bad range");
    HIPCK(c, hipSetDevice(c->dev));
    if (n == 0) return MP_OK;
    uint32_t *d_out = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &d_out, (size_t)3 * n * wsz(c)))) return rc;
    if (c->wide)
        hipLaunchKernelGGL(window_words_kernel<uint64_t>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, msa_args(c),
                           c->p0 + w, c->k, row0, n, reinterpret_cast<uint64_t *>(d_out));
    else
        hipLaunchKernelGGL(window_words_kernel<uint32_t>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, msa_args(c),
                           c->p0 + w, c->k, row0, n, d_out);
    hipError_t e = hipMemcpyAsync(out, d_out, sizeof(uint32_t) * 3 * (size_t)n * wsz(c), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(c, &d_out, (size_t)3 * n * wsz(c));
    if (e != hipSuccess) return fail(c, MP_ERR_DEVICE, "mp_get_window_words: %s", hipGetErrorString(e));
    return MP_OK;
}


}  // extern "C"
