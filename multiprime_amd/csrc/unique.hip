// unique.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of
// include/mprime.h.  Per-window k-mer histograms in an LDS hash table (mp_window_unique).
#include "common.hpp"

using namespace mp;

namespace {

// ----------------------------------------------------------------------------------------------
// (3) per-window k-mer histogram (V20:689-711)
// ----------------------------------------------------------------------------------------------
__device__ inline uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + b * 0x85EBCA77u;
    h = (h ^ (h >> 13)) + c * 0xC2B2AE3Du;
    return h ^ (h >> 16);
}

struct UniqueOut {
    uint32_t *b0, *b1, *g;
    int32_t *count, *first;
    long long cap;
    unsigned long long *total;   // entries allocated so far (may exceed cap: caller checks)
    int64_t *win_base;           // [W]
    int32_t *win_count;          // [W]
    int32_t *labels;             // [W][Npad] or nullptr
    int32_t *overflow;           // [W] set to 1 when the table did not fit
};

// One block per window.  Rows stream through in lanes; equal keys inside a wave are folded with
// ballots first (conserved windows put the same k-mer in almost every lane), then one lane per
// distinct key updates the table.  A slot stores the row of a representative; key comparison
// reads the representative's window words back (immutable, L2-resident).
// TABLE_IN_LDS = false: same algorithm on a global-memory table (windows with more distinct k-mers
// than the LDS table holds).
template <bool TABLE_IN_LDS, bool P64>
__global__ __launch_bounds__(kBlock) void unique_kernel(const void *__restrict__ win, int k, int n_rows, int n_pad,
                                                        const int32_t *__restrict__ win_list, int slots, int limit,
                                                        uint32_t *__restrict__ gtable, UniqueOut out) {
    __shared__ uint32_t s_rep[TABLE_IN_LDS ? kHashSlots : 1];
    __shared__ uint32_t s_cnt[TABLE_IN_LDS ? kHashSlots : 1];
    __shared__ uint32_t s_min[TABLE_IN_LDS ? kHashSlots : 1];
    __shared__ int s_used, s_over, s_nout;
    __shared__ unsigned long long s_base;
    const int w = win_list ? win_list[blockIdx.x] : blockIdx.x;
    uint32_t *rep, *cnt, *mn;
    if (TABLE_IN_LDS) { rep = s_rep; cnt = s_cnt; mn = s_min; }
    else { rep = gtable + (size_t)blockIdx.x * 3 * slots; cnt = rep + slots; mn = cnt + slots; }
    const uint32_t mask = slots - 1;
    for (int i = threadIdx.x; i < slots; i += kBlock) { rep[i] = kEmpty; cnt[i] = 0; mn[i] = kEmpty; }
    if (threadIdx.x == 0) { s_used = 0; s_over = 0; s_nout = 0; }
    __syncthreads();
    const size_t np = (size_t)n_pad;
    const WinView<P64> V(win, w, np, k, (1u << k) - 1u);
    const int lane = threadIdx.x & 63;
    for (int base = 0; base < n_pad; base += kBlock) {
        int r = base + threadIdx.x;
        uint32_t b0 = 0, b1 = 0, g = MP_WIN_SKIP;
        if (r < n_rows) V.load(r, b0, b1, g);
        bool todo = !(g & MP_WIN_SKIP);
        unsigned long long pending = __ballot(todo);
        while (pending) {
            int lead = __ffsll((long long)pending) - 1;
            uint32_t k0 = __shfl(b0, lead), k1 = __shfl(b1, lead), k2 = __shfl(g, lead);
            bool same = todo && b0 == k0 && b1 == k1 && g == k2;
            unsigned long long grp = __ballot(same);
            if (lane == lead) {
                uint32_t c = (uint32_t)__popcll(grp);
                uint32_t h = hash3(b0, b1, g) & mask;
                for (int probe = 0; probe < slots; probe++) {
                    uint32_t old = atomicCAS(&rep[h], kEmpty, (uint32_t)r);
                    bool hit = old == kEmpty;
                    if (hit) {
                        if (atomicAdd(&s_used, 1) + 1 > limit) s_over = 1;
                    } else {
                        uint32_t o0, o1, o2;
                        V.load((int)old, o0, o1, o2);
                        hit = o0 == b0 && o1 == b1 && o2 == g;
                    }
                    if (hit) { atomicAdd(&cnt[h], c); atomicMin(&mn[h], (uint32_t)r); break; }
                    h = (h + 1) & mask;
                }
            }
            todo = todo && !same;
            pending &= ~grp;
        }
        if (s_over) break;       // benign race: every thread re-checks after the barrier below
    }
    __syncthreads();
    if (s_over) {
        if (threadIdx.x == 0) { out.overflow[w] = 1; out.win_count[w] = 0; out.win_base[w] = 0; }
        return;
    }
    // compaction: one reservation in the global entry list per window (s_used = distinct k-mers),
    // then every occupied slot takes a dense index inside the window's segment
    if (threadIdx.x == 0) {
        s_base = atomicAdd(out.total, (unsigned long long)s_used);
        out.win_base[w] = (int64_t)s_base;
        out.win_count[w] = s_used;
        out.overflow[w] = 0;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    for (int i = threadIdx.x; i < slots; i += kBlock) {
        uint32_t rr = rep[i];
        if (rr == kEmpty) continue;
        int idx = atomicAdd(&s_nout, 1);
        unsigned long long e = base + idx;
        if ((long long)e < out.cap) {
            uint32_t o0, o1, o2;
            V.load((int)rr, o0, o1, o2);
            out.b0[e] = o0; out.b1[e] = o1; out.g[e] = o2;
            out.count[e] = (int32_t)cnt[i];
            out.first[e] = (int32_t)mn[i];
        }
        cnt[i] = (uint32_t)idx;            // count consumed: reuse the word as the slot's dense index
    }
    __syncthreads();
    if (!out.labels) return;
    for (int base_r = 0; base_r < n_pad; base_r += kBlock) {
        int r = base_r + threadIdx.x;
        if (r >= n_rows) continue;
        uint32_t b0, b1, g;
        V.load(r, b0, b1, g);
        int32_t lab = -1;
        if (!(g & MP_WIN_SKIP)) {
            uint32_t h = hash3(b0, b1, g) & mask;
            for (int probe = 0; probe < slots; probe++) {
                uint32_t rr = rep[h];
                if (rr == kEmpty) break;
                uint32_t o0, o1, o2;
                V.load((int)rr, o0, o1, o2);
                if (o0 == b0 && o1 == b1 && o2 == g) { lab = (int32_t)cnt[h]; break; }
                h = (h + 1) & mask;
            }
        }
        out.labels[(size_t)w * np + r] = lab;
    }
}


}  // namespace

extern "C" {

int mp_window_unique(mp_ctx *c, int64_t cap, int32_t want_labels, int64_t *n_entries) {
    if (!c) return MP_ERR_ARG;
    if (!c->win) return fail(c, MP_ERR_ARG, "no windows built");
    if (cap <= 0) return fail(c, MP_ERR_ARG, "cap_entries must be positive");
    HIPCK(c, hipSetDevice(c->dev));
    free_unique(c);
    size_t W = (size_t)c->n_win, np = (size_t)c->n_pad;
    int rc;
    c->u_cap = cap;
    if ((rc = dev_alloc(c, &c->u_b0, (size_t)cap))) return rc;
    if ((rc = dev_alloc(c, &c->u_b1, (size_t)cap))) return rc;
    if ((rc = dev_alloc(c, &c->u_g, (size_t)cap))) return rc;
    if ((rc = dev_alloc(c, &c->u_count, (size_t)cap))) return rc;
    if ((rc = dev_alloc(c, &c->u_first, (size_t)cap))) return rc;
    if ((rc = dev_alloc(c, &c->u_over, W))) return rc;
    if ((rc = dev_alloc(c, &c->u_wcount, W))) return rc;
    if ((rc = dev_alloc(c, &c->u_wbase, W))) return rc;
    if ((rc = dev_alloc(c, &c->u_total, 1))) return rc;
    if (want_labels && (rc = dev_alloc(c, &c->labels, W * np))) return rc;
    HIPCK(c, hipMemsetAsync(c->u_total, 0, sizeof(unsigned long long), c->stream));
    UniqueOut uo{c->u_b0, c->u_b1, c->u_g, c->u_count, c->u_first, (long long)cap, c->u_total,
                 c->u_wbase, c->u_wcount, c->labels, c->u_over};
    if (c->p64)
        hipLaunchKernelGGL((unique_kernel<true, true>), dim3((unsigned)W), dim3(kBlock), 0, c->stream, (const void *)c->win, c->k,
                           c->n_rows, c->n_pad, (const int32_t *)nullptr, kHashSlots, kHashLimit, (uint32_t *)nullptr, uo);
    else
        hipLaunchKernelGGL((unique_kernel<true, false>), dim3((unsigned)W), dim3(kBlock), 0, c->stream, (const void *)c->win, c->k,
                           c->n_rows, c->n_pad, (const int32_t *)nullptr, kHashSlots, kHashLimit, (uint32_t *)nullptr, uo);
    HIPCK(c, hipGetLastError());
    std::vector<int32_t> over(W);
    HIPCK(c, hipMemcpyAsync(over.data(), c->u_over, sizeof(int32_t) * W, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    std::vector<int32_t> big;
    for (size_t w = 0; w < W; w++) if (over[w]) big.push_back((int32_t)w);
    if (!big.empty()) {
        // windows with more distinct k-mers than the LDS table holds: same kernel on a global table
        int slots = 1;
        while (slots < 2 * c->n_rows + 64) slots <<= 1;
        const size_t batch = 64;
        uint32_t *gtable = nullptr;
        int32_t *d_list = nullptr;
        if ((rc = dev_alloc(c, &gtable, batch * 3 * (size_t)slots))) return rc;
        if ((rc = dev_alloc(c, &d_list, batch))) return rc;
        for (size_t i = 0; i < big.size(); i += batch) {
            size_t nb = std::min(batch, big.size() - i);
            HIPCK(c, hipMemcpy(d_list, big.data() + i, sizeof(int32_t) * nb, hipMemcpyHostToDevice));
            if (c->p64)
                hipLaunchKernelGGL((unique_kernel<false, true>), dim3((unsigned)nb), dim3(kBlock), 0, c->stream, (const void *)c->win,
                                   c->k, c->n_rows, c->n_pad, (const int32_t *)d_list, slots, slots - 32, gtable, uo);
            else
                hipLaunchKernelGGL((unique_kernel<false, false>), dim3((unsigned)nb), dim3(kBlock), 0, c->stream, (const void *)c->win,
                                   c->k, c->n_rows, c->n_pad, (const int32_t *)d_list, slots, slots - 32, gtable, uo);
            HIPCK(c, hipGetLastError());
            HIPCK(c, hipStreamSynchronize(c->stream));
        }
        dev_free(c, &gtable, batch * 3 * (size_t)slots);
        dev_free(c, &d_list, batch);
    }
    unsigned long long total = 0;
    c->h_wbase.resize(W); c->h_wcount.resize(W);
    HIPCK(c, hipMemcpy(&total, c->u_total, sizeof(total), hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(c->h_wbase.data(), c->u_wbase, sizeof(int64_t) * W, hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(c->h_wcount.data(), c->u_wcount, sizeof(int32_t) * W, hipMemcpyDeviceToHost));
    if (n_entries) *n_entries = (int64_t)total;
    if ((long long)total > cap) { c->u_n = 0; return fail(c, MP_ERR_CAPACITY, "unique table needs %llu entries", total); }
    c->u_n = (long long)total;
    return MP_OK;
}

int mp_get_unique(mp_ctx *c, int64_t *win_off, uint32_t *words, int32_t *count, int32_t *first_row) {
    if (!c) return MP_ERR_ARG;
    if (c->h_wbase.empty()) return fail(c, MP_ERR_ARG, "mp_window_unique has not run");
    HIPCK(c, hipSetDevice(c->dev));
    size_t n = (size_t)c->u_n, W = (size_t)c->n_win;
    std::vector<uint32_t> b0(n + 1), b1(n + 1), g(n + 1);
    std::vector<int32_t> cn(n + 1), fr(n + 1);
    if (n) {
        HIPCK(c, hipMemcpy(b0.data(), c->u_b0, 4 * n, hipMemcpyDeviceToHost));
        HIPCK(c, hipMemcpy(b1.data(), c->u_b1, 4 * n, hipMemcpyDeviceToHost));
        HIPCK(c, hipMemcpy(g.data(), c->u_g, 4 * n, hipMemcpyDeviceToHost));
        HIPCK(c, hipMemcpy(cn.data(), c->u_count, 4 * n, hipMemcpyDeviceToHost));
        HIPCK(c, hipMemcpy(fr.data(), c->u_first, 4 * n, hipMemcpyDeviceToHost));
    }
    // the kernel reserved each window's segment with one atomic; lay the segments out in window order
    int64_t o = 0;
    for (size_t w = 0; w < W; w++) {
        win_off[w] = o;
        size_t src = (size_t)c->h_wbase[w], m = (size_t)c->h_wcount[w];
        for (size_t i = 0; i < m; i++) {
            words[(size_t)o + i] = b0[src + i];
            words[n + (size_t)o + i] = b1[src + i];
            words[2 * n + (size_t)o + i] = g[src + i];
            count[(size_t)o + i] = cn[src + i];
            first_row[(size_t)o + i] = fr[src + i];
        }
        o += (int64_t)m;
    }
    win_off[W] = o;
    return MP_OK;
}

int mp_get_labels(mp_ctx *c, int32_t w, int32_t *labels) {
    if (!c) return MP_ERR_ARG;
    if (!c->labels) return fail(c, MP_ERR_ARG, "labels were not requested");
    if (w < 0 || w >= c->n_win) return fail(c, MP_ERR_ARG, "bad window");
    HIPCK(c, hipSetDevice(c->dev));
    HIPCK(c, hipMemcpy(labels, c->labels + (size_t)w * c->n_pad, sizeof(int32_t) * (size_t)c->n_rows, hipMemcpyDeviceToHost));
    return MP_OK;
}


}  // extern "C"
