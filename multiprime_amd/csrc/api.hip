// api.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of
// include/mprime.h.  Context lifetime and bookkeeping.
#include <sys/mman.h>
#include <execinfo.h>
#include <exception>
#include <algorithm>

#include <thread>
#include <vector>

#include "common.hpp"
#include "workers.hpp"

namespace mp {

namespace {
struct FillArgs { uint32_t *p[kMaxFillSegs]; unsigned long long words[kMaxFillSegs]; uint32_t value[kMaxFillSegs]; unsigned first_block[kMaxFillSegs + 1]; int n; };
// thread = 4 consecutive words of one segment
__global__ __launch_bounds__(kBlock) void fill_kernel(const FillArgs A) {
    int s = 0;
    while (s + 1 < A.n && blockIdx.x >= A.first_block[s + 1]) s++;
    const unsigned long long w0 = ((unsigned long long)(blockIdx.x - A.first_block[s]) * kBlock + threadIdx.x) * 4ull;
    const uint32_t val = A.value[s];
    uint32_t *dst = A.p[s] + w0;
    if (w0 + 4 <= A.words[s]) *reinterpret_cast<uint4 *>(dst) = uint4{val, val, val, val};
    else
        for (unsigned long long w = w0; w < A.words[s]; w++) A.p[s][w] = val;
}
}  // namespace

int fill_segments(mp_ctx *c, const FillSeg *segs, int n) {
    FillArgs A{};
    unsigned blocks = 0;
    for (int i = 0; i < n && A.n < kMaxFillSegs; i++) {
        if (!segs[i].p || !segs[i].bytes) continue;
        A.p[A.n] = (uint32_t *)segs[i].p; A.words[A.n] = segs[i].bytes / 4; A.value[A.n] = segs[i].value; A.first_block[A.n] = blocks;
        blocks += (unsigned)((A.words[A.n] + (size_t)kBlock * 4 - 1) / ((size_t)kBlock * 4));
        A.n++;
    }
    if (!A.n) return MP_OK;
    A.first_block[A.n] = blocks;
    hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(kBlock), 0, c->stream, A);
    HIPCK(c, hipGetLastError());
    return MP_OK;
}

void *host_map(size_t bytes) {
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    (void)madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}
void host_unmap(void *p, size_t bytes) { if (p) (void)munmap(p, bytes); }

void prefault_host(void *p, size_t bytes) {
    constexpr size_t kPage = 4096, kMin = (size_t)4 << 20;
    if (!p || bytes < kMin || getenv("MP_NO_PREFAULT")) return;
    unsigned hw = std::thread::hardware_concurrency();
    const size_t n_thr = std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)(hw ? hw : 1), bytes / ((size_t)2 << 20)}));
    uint8_t *b = static_cast<uint8_t *>(p);
    auto touch = [=](size_t t) {
        const size_t lo = bytes * t / n_thr, hi = bytes * (t + 1) / n_thr;
        for (size_t o = (lo + kPage - 1) / kPage * kPage; o < hi; o += kPage) *reinterpret_cast<volatile uint8_t *>(b + o) = 0;
        if (lo < hi) *reinterpret_cast<volatile uint8_t *>(b + lo) = 0;
    };
    mp::run_on_threads((int)n_thr, [&](int t) { touch((size_t)t); });
}

void free_eval(mp_ctx *c) {
    // [r6] launches on the context's second stream (mp_eval_launch_alt) read what is released below, and the pool's "released" events are
    // recorded on the FIRST stream: whatever the second one still runs has finished before any of it changes hands
    if (c->alt_stream) (void)hipStreamSynchronize(c->alt_stream);
    free_slide(c);
    c->h_chains.clear(); c->h_events.clear(); c->h_cand_out.clear();
    dev_free(c, &c->chain_prog, c->chain_prog_n);
    c->chain_prog_n = 0;
    c->prog_shape = -1;
    dev_free(c, &c->items, (size_t)c->n_items);
    dev_free(c, &c->cand_n, (size_t)c->n_padded * wsz(c));
    dev_free(c, &c->cand_out, (size_t)c->n_padded);
    dev_free(c, &c->cand_symT, (size_t)c->n_items * 32);
    dev_free(c, &c->cand_diff, (size_t)c->n_items);
    dev_free(c, &c->chain_items, (size_t)c->n_chain);
    if (c->n_chain) dev_free(c, &c->chain_events, (size_t)c->n_events + 1);
    dev_free(c, &c->table_ids, (size_t)c->n_table);
    if (c->x_items) { void *q = c->x_items; const size_t real = pool_forget(c, q, 0); if (!real || !pool_give(c, q, real)) (void)hipFree(q); c->bytes -= (int64_t)real; c->x_items = nullptr; }
    dev_free(c, &c->x_events, (size_t)c->x_n_events + 1); dev_free(c, &c->x_cand_out, (size_t)c->x_n * 8);
    c->x_n = c->x_n_events = 0;
    c->n_chain = c->n_table = c->n_events = 0;
    c->n_items = c->n_padded = c->n_cand = 0;
}

void free_unique(mp_ctx *c) {
    size_t cap = (size_t)c->u_cap, W = (size_t)c->n_win;
    dev_free(c, &c->u_b0, cap * wsz(c)); dev_free(c, &c->u_b1, cap * wsz(c)); dev_free(c, &c->u_g, cap * wsz(c));
    dev_free(c, &c->u_count, cap); dev_free(c, &c->u_first, cap);
    dev_free(c, &c->labels, W * c->n_pad);
    dev_free(c, &c->u_over, W); dev_free(c, &c->u_wcount, W); dev_free(c, &c->u_wbase, W);
    dev_free(c, &c->u_total, 1);
    dev_free(c, &c->g_key, W * (size_t)c->g_slots); dev_free(c, &c->g_cnt, (MP_HIST_CM64 ? 2 : 1) * W * (size_t)c->g_slots);
    dev_free(c, &c->g_min, MP_HIST_CM64 ? (size_t)1 : W * (size_t)c->g_slots); dev_free(c, &c->g_idx, W * (size_t)c->g_slots);
    dev_free(c, &c->g_gap, W * (size_t)c->g_slots);
    c->g_slots = 0;
    c->u_cap = c->u_n = 0;
    c->h_wbase.clear(); c->h_wcount.clear();
}

void free_windows(mp_ctx *c) {
    {   // (exception records nobody collected go with their windows)
        std::lock_guard<std::mutex> lock(c->ex_mu);
        c->ex_pending = 0;
    }
    free_eval(c);
    dev_free(c, &c->mask_f, c->mask_words); dev_free(c, &c->mask_r, c->mask_words);
    c->mask_words = 0; c->n_masks = 0;
    free_unique(c);
    dev_free(c, &c->excl, (size_t)c->n_win * (c->n_pad / 64));
    dev_free(c, &c->patch_count, (size_t)c->n_win * 32);
    dev_free(c, &c->patch_off, (size_t)c->n_win + 1);
    dev_free(c, &c->patch_cursor, (size_t)c->n_win * 32);
    dev_free(c, &c->patch_words, (size_t)3 * c->n_patch * wsz(c));
    dev_free(c, &c->patch_rows, (size_t)c->n_patch);
    c->n_patch = c->max_patch = 0;
    dev_free(c, &c->ex, (size_t)c->ex_cap);
    dev_free(c, &c->ex_count, 1);
    dev_free(c, &c->err_flag, 4);
    dev_free(c, &c->qplanes, c->pp_words); dev_free(c, &c->qvalid, c->pv_words);
    dev_free(c, &c->pplanes, c->pp_words); dev_free(c, &c->pvalid, c->pv_words); dev_free(c, &c->pwin, (size_t)c->n_win);
    c->pp_words = c->pv_words = 0; c->max_npw = 0; c->pp_dirty = true; c->qp_dirty = true;
    dev_free(c, &c->extra_off, (size_t)c->n_win + 1);
    dev_free(c, &c->extra_words, (size_t)3 * c->n_extra * wsz(c));
    c->ex_cap = 0; c->n_extra = 0; c->n_win = 0;
    c->ex_host.clear();
}

void free_msa(mp_ctx *c) {
    free_windows(c);
    size_t np = (size_t)c->n_pad;
    dev_free(c, &c->planes, (size_t)c->n_chunks * 4 * np);
    dev_free(c, &c->cols, ((size_t)c->n_chunks * 32 * 4 + 1) * (np / 64));
    dev_free(c, &c->cum, ((size_t)c->n_chunks + 1) * np);
    dev_free(c, &c->ung, np * c->ustride);
    dev_free(c, &c->lead, np); dev_free(c, &c->rstrip, np); dev_free(c, &c->rlen, np);
    c->n_rows = c->n_pad = c->n_chunks = 0;
}


}  // namespace mp

namespace mp {

constexpr size_t kPoolBlocks = 96, kPoolBytes = (size_t)3 << 30, kPoolBlockMax = (size_t)1 << 30;

static bool pool_enabled(mp_ctx *c) {
    if (c->pool_on < 0) {
        const char *e = getenv("MP_DEVICE_POOL");
        c->pool_on = e && atoi(e) == 0 ? 0 : 1;
    }
    return c->pool_on == 1;
}

// live contexts of the process (pool_register): a failed allocation drains all their pools, and contexts on one device share kPoolBytes
static std::mutex g_ctx_mu;
static std::vector<mp_ctx *> g_ctxs;

void pool_register(mp_ctx *c, bool alive) {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    if (alive) g_ctxs.push_back(c);
    else g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), c), g_ctxs.end());
}

static size_t pool_byte_limit(const mp_ctx *c) {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    size_t same = 0;
    for (const mp_ctx *o : g_ctxs) same += o->dev == c->dev;
    return kPoolBytes / std::max<size_t>(same, 1);
}

void *pool_take(mp_ctx *c, size_t bytes) {
    void *p = nullptr;
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> g(c->pool_mu);
        if (!pool_enabled(c)) return nullptr;
        for (size_t i = c->pool.size(); i-- > 0;)               // the youngest block of the size: the likeliest to be in a cache still
            if (c->pool[i].bytes == bytes) {
                p = c->pool[i].p;
                ev = c->pool[i].released;
                c->pool.erase(c->pool.begin() + (long)i);
                c->pool_bytes -= bytes;
                c->pool_hits++;
                break;
            }
        if (!p) { c->pool_misses++; return nullptr; }
    }
    // whatever the context's stream had queued when the block was released has finished before the block is used again — by a kernel
    // on that stream (already ordered), by a blocking copy on the null stream or by a caller's non-blocking stream (not ordered otherwise)
    if (ev) {
        (void)hipEventSynchronize(ev);
        std::lock_guard<std::mutex> g(c->pool_mu);
        c->pool_events.push_back(ev);
    }
    return p;
}

bool pool_give(mp_ctx *c, void *p, size_t bytes) {
    if (bytes > kPoolBlockMax) return false;
    const size_t limit = pool_byte_limit(c);
    std::lock_guard<std::mutex> g(c->pool_mu);
    if (!pool_enabled(c) || bytes > limit) return false;
    while (!c->pool.empty() && (c->pool.size() >= kPoolBlocks || c->pool_bytes + bytes > limit)) {
        (void)hipFree(c->pool.front().p);                   // (hipFree waits for the device itself)
        if (c->pool.front().released) c->pool_events.push_back(c->pool.front().released);
        c->pool_bytes -= c->pool.front().bytes;
        c->pool.erase(c->pool.begin());
    }
    hipEvent_t ev = nullptr;
    if (!c->pool_events.empty()) { ev = c->pool_events.back(); c->pool_events.pop_back(); }
    else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipEventRecord(ev, c->stream) != hipSuccess) { (void)hipGetLastError(); c->pool_events.push_back(ev); return false; }
    c->pool.push_back({p, bytes, ev});
    c->pool_bytes += bytes;
    return true;
}

void pool_note(mp_ctx *c, void *p, size_t bytes) {
    std::lock_guard<std::mutex> g(c->pool_mu);
    c->pool_live[p] = bytes;
}

size_t pool_forget(mp_ctx *c, void *p, size_t claimed) {
    std::lock_guard<std::mutex> g(c->pool_mu);
    auto it = c->pool_live.find(p);
    if (it == c->pool_live.end()) return 0;
    const size_t real = it->second;
    c->pool_live.erase(it);
    if (real != claimed && getenv("MP_TRACE"))
        fprintf(stderr, "[mprime] dev_free: a block of %zu bytes released as %zu bytes\n", real, claimed);
    return real;
}

void pool_drain(mp_ctx *c) {
    std::lock_guard<std::mutex> g(c->pool_mu);
    for (auto &b : c->pool) { (void)hipFree(b.p); if (b.released) c->pool_events.push_back(b.released); }
    c->pool.clear();
    c->pool_bytes = 0;
}

void pool_drain_all() {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    for (mp_ctx *o : g_ctxs) pool_drain(o);
}

}  // namespace mp

using namespace mp;

// Copies between the device and ordinary (pageable) host memory: from a minimum size on, the HIP runtime page-locks the user's
// range for the transfer instead of going through its own staging buffers — pure cost for buffers that are used once (the 58 MB of
// histogram entries of a 131072 x 1000 alignment: 27-32 ms with it, 6.5 ms without).  The threshold (GPU_PINNED_MIN_XFER_SIZE) is a
// process-wide runtime setting read when the runtime starts, so it is NOT this library's to change: the drop-in command lines raise
// it before anything starts the HIP runtime (multiprime_amd/_abi.py prefer_staged_copies); a host application decides for itself.

extern "C" {

const char *mp_backend_name(void) { return "hip"; }
const char *mp_last_error(const mp_ctx *c) { return c ? c->err : "mp_create failed: no usable HIP device"; }

int mp_create(int device, mp_ctx **out) {
    if (!out) return MP_ERR_ARG;
    *out = nullptr;
    if (getenv("MP_DEBUG_TERMINATE")) {        // debugging: the stack of whoever lets an exception escape (the runtime's threads included)
        std::set_terminate([] {
            void *bt[64];
            const int n = backtrace(bt, 64);
            fprintf(stderr, "[mprime] std::terminate: %d frames\n", n);
            backtrace_symbols_fd(bt, n, 2);
            abort();
        });
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return MP_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MP_ERR_DEVICE;
    mp_ctx *c = new mp_ctx();
    c->dev = device;
    if (pack_init() != MP_OK || dimer_init() != MP_OK) { delete c; return MP_ERR_DEVICE; }
    {   // The runtime sets up its path for copies to and from pageable host memory on first use: 8 ms in front of the first counter
        // read-back of mp_build_windows (MP_TRACE).  The drop-in creates the context beside the FASTA parse, so that is paid here.
        std::vector<uint32_t> h((size_t)1 << 16, 0u);
        uint32_t *d = nullptr;
        if (hipMalloc((void **)&d, h.size() * sizeof(uint32_t)) == hipSuccess) {
            (void)hipMemcpyAsync(d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
            (void)hipMemcpyAsync(h.data(), d, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
            (void)hipStreamSynchronize(c->stream);
            (void)hipFree(d);
        }
    }
    pool_register(c, true);
    *out = c;
    return MP_OK;
}

void mp_destroy(mp_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->dev);
    (void)hipDeviceSynchronize();
    free_comm(c);
    free_msa(c);
    free_seq(c);
    // The library's own streams go AFTER the stages: free_eval() waits for the second stream before it releases what launches on it read.
    // [r6] They went first, so that wait named a destroyed stream (freed memory of the runtime).  Found by reading while hunting two of 57
    // `bench.py` runs that died inside the runtime (std::bad_variant_access / SIGSEGV); 3000 contexts in either order did not reproduce a
    // crash (profiles/r06_close_stress.txt), so this is a fix of a real fault, not a proven cause: bench.py measures in a child process.
    if (c->alt_stream) { (void)hipStreamDestroy(c->alt_stream); c->alt_stream = nullptr; }
    if (c->ex_stream) { (void)hipStreamDestroy(c->ex_stream); c->ex_stream = nullptr; }
    dev_free(c, &c->tmp_out, (size_t)c->tmp_out_n);
    dev_free(c, &c->stats_buf, c->stats_buf_n);
    if (getenv("MP_TRACE")) fprintf(stderr, "[mprime] device blocks: %lld reused, %lld from the runtime, %zu waiting (%.1f MB)\n", c->pool_hits, c->pool_misses,
                                    c->pool.size(), c->pool_bytes / 1048576.0);
    if (c->h_stats) { if (c->h_stats_pinned) (void)hipHostUnregister(c->h_stats); host_unmap(c->h_stats, c->h_stats_bytes); }
    if (c->stats_ev) (void)hipEventDestroy(c->stats_ev);
    if (c->h_ring) {
        if (c->h_ring_pinned) (void)hipHostUnregister(c->h_ring);
        host_unmap(c->h_ring, (size_t)96 << 20);
        for (hipEvent_t ev : c->h_ring_ev) if (ev) (void)hipEventDestroy(ev);
    }
    if (c->h_stage_pinned) (void)hipHostUnregister(c->h_stage);
    host_unmap(c->h_stage, c->h_stage_bytes);
    dev_free(c, &c->dm_loss, (size_t)(MP_DIMER_MAX_LEN + 1) * (MP_DIMER_MAX_LEN + 1) * 64);
    dev_free(c, &c->dm_dg, (size_t)(16 + 32 + MP_DIMER_MAX_LEN + 1 + 1));
    for (auto &p : c->ev_busy) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &p : c->ev_free) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    pool_drain(c);                                       // (last: the frees above went through the pool)
    for (hipEvent_t ev : c->pool_events) (void)hipEventDestroy(ev);
    pool_register(c, false);
    delete c;
}

int mp_set_stream(mp_ctx *c, void *s) {
    if (!c) return MP_ERR_ARG;
    if (c->stream != (hipStream_t)s) {
        // device blocks change hands between stages without a wait (common.hpp: the pool) because everything is ordered on ONE stream:
        // what the old stream still has in flight is finished before the new one takes over
        HIPCK(c, hipSetDevice(c->dev));
        HIPCK(c, hipStreamSynchronize(c->stream));
    }
    c->stream = (hipStream_t)s;
    return MP_OK;
}

int mp_device_bytes(mp_ctx *c, int64_t *b) {
    if (!c || !b) return MP_ERR_ARG;
    *b = c->bytes;
    return MP_OK;
}


}  // extern "C"
