// common.hpp — shared declarations of libmprime_hip.so (hand-written HIP for gfx950 / MI355X behind
// the C ABI of include/mprime.h).
//
// Data layout in HBM:
//   planes  [n_chunks][4][Npad] u32   base-set bit planes (A,C,G,T membership) of 32 alignment columns
//                                     per word, sequences along the fastest axis (coalesced per wave)
//   cols    [L][4][Npad/64] u64       column planes A,C,G,T (one-hot, gap = none) — one bit per sequence — for the bit-sliced evaluation
//   cum     [n_chunks+1][Npad] u32    residues (non-gap symbols) left of each 32-column chunk
//   ung     [ustride][Npad] u32       gap-free residue codes, 8 nibbles per word, word-major so that a wave's stores and
//                                     loads of one word index coalesce (edge-gap repair only)
//   The k-mer of a (window, sequence) pair ("window words" b0,b1,g) is never stored: every consumer derives it from
//   the planes on the fly (winwords.hpp) — round 1 kept a [W][Npad] u64 array of them (1 GB at the bench shard).
//   excl / patch list, histogram tables and entries, labels: see windows.hip / unique.hip
// Translation units: pack.hip (mp_load_msa), windows.hip (mp_build_windows), unique.hip (histograms),
// eval.hip (candidate x sequence evaluation), dimer.hip (3'-end dimer scan, pair coverage), api.hip.
// No MFMA anywhere: this is bit-mask work bounded by integer ALU / HBM.  gfx950 only.
#pragma once

#include "../../include/mprime.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <chrono>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace mp {

constexpr int kBlock = 256;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr int kHashSlots = 4096;          // LDS hash table slots per window (unique_kernel)
constexpr int kHashLimit = 3584;          // load limit before the window is handed to the global-table path
constexpr int kEvalCC = 8;                // candidates evaluated per block pass

struct ExRec { int32_t win, row; uint64_t q[4]; };          // exception k-mer: up to 64 symbol nibbles (NibT words)
struct EvalItem { int32_t win, cand0; };                       // one block's work: window + first padded candidate
// Patch planes of one window: k x 4 plane rows of npw words from pplanes[poff], npw validity words from pvalid[voff]
struct PatchWin { int32_t poff, voff, npw; };
// One nested run of candidates for eval_chain_kernel: 8 output slots from cand0, n_steps of them used,
// n_ev events (position | lost base << 8 | step << 16) from ev0.
struct ChainItem {
    int32_t win, cand0, n_steps, ev0, n_ev;
    uint32_t sym[4];                   // nibble j = symbol of the first (most degenerate) candidate at position j
    uint32_t pos1, pos2, pos4;         // positions where that symbol has one / two / more bases
};

struct SlideBand;

// up to 16 * NQ symbol codes, one nibble each (position j = nibble j & 15 of word j >> 4).  NQ = 2: the k-mers of 32-bit window
// words (the code of rounds 1-3, two named halves); NQ = 4: primers of 32..63 bases — the word index is a run-time value there, taken
// through selects so that the words stay in registers.
template <int NQ> struct NibT {
    uint64_t q[NQ];
    __device__ void clear() {
#pragma unroll
        for (int i = 0; i < NQ; i++) q[i] = 0;
    }
    __device__ uint32_t get(int j) const {
        const int wi = j >> 4;
        uint64_t w = q[0];
#pragma unroll
        for (int i = 1; i < NQ; i++) w = wi == i ? q[i] : w;
        return (uint32_t)(w >> (4 * (j & 15))) & 15u;
    }
    __device__ void set(int j, uint32_t v) {
        const int wi = j >> 4, sh = 4 * (j & 15);
#pragma unroll
        for (int i = 0; i < NQ; i++)
            if (wi == i) q[i] = (q[i] & ~(15ull << sh)) | ((uint64_t)v << sh);
    }
    __device__ void shift_up(int n) {      // move every nibble n positions towards the 3' end
        const int s = 4 * n, ws = s >> 6, bs = s & 63;
        if (s == 0) return;
        uint64_t out[NQ];
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            uint64_t a = 0, b = 0;           // a = q[i - ws], b = q[i - ws - 1]
#pragma unroll
            for (int t = 0; t < NQ; t++) {
                if (t == i - ws) a = q[t];
                if (t == i - ws - 1) b = q[t];
            }
            out[i] = bs ? (a << bs) | (b >> (64 - bs)) : a;
        }
#pragma unroll
        for (int i = 0; i < NQ; i++) q[i] = out[i];
    }
};
typedef NibT<2> Nib;

}  // namespace mp

// ================================================================================================
// the opaque context
// ================================================================================================
// MP_TRACE=1: host-side stage times of the library calls on stderr (each lap waits for the stream first)
struct Lap {
    bool on = getenv("MP_TRACE") != nullptr;
    hipStream_t st;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    explicit Lap(hipStream_t s) : st(s) {}
    void operator()(const char *what) {
        if (!on) return;
        (void)hipStreamSynchronize(st);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mprime] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

#ifndef MP_HIST_CM64
#define MP_HIST_CM64 2                       // (unique.hip; mp_ctx::g_cnt below)
#endif

struct mp_ctx {
    char err[512] = {0};
    int dev = 0;
    hipStream_t stream = nullptr;
    const void *rot_block = nullptr;         // mp_eval_launch_rotating: the block the last rotating launch cleared, and for how many counters
    size_t rot_cleared = 0;
    hipStream_t alt_stream = nullptr;        // mp_eval_launch_alt: the library's own second stream (eval.hip)
    int64_t bytes = 0;
    // alignment
    int n_rows = 0, n_pad = 0, n_chunks = 0, max_len = 0, ustride = 0;
    int reserve_cols = 0;                    // mp_reserve_columns: minimum alignment width (row shards)
    uint32_t *planes = nullptr, *cum = nullptr, *ung = nullptr;
    unsigned long long *cols = nullptr;      // [n_chunks*32][4][n_pad/64]
    int32_t *lead = nullptr, *rstrip = nullptr, *rlen = nullptr;
    // windows
    int p0 = 0, n_win = 0, k = 0, v = 0;
    bool wide = false;                       // k > MP_NARROW_K: window words are 64-bit (patch_words, extra_words, u_b0/u_b1/u_g hold two uint32 per word)
    unsigned long long *excl = nullptr;      // [W][n_pad/64]; non-null = windows are built
    int32_t *patch_count = nullptr, *patch_off = nullptr, *patch_cursor = nullptr;
    uint32_t *patch_words = nullptr;         // [n_patch][3] window words of the slow pairs (SKIP = exception / too short)
    int32_t *patch_rows = nullptr;           // [n_patch] their rows, grouped by window like patch_words
    int n_patch = 0, max_patch = 0;
    bool p64 = false;                        // 3k <= 63: histogram keys are one packed u64
    mp::ExRec *ex = nullptr;
    int ex_cap = 0;
    int *ex_count = nullptr, *err_flag = nullptr;
    std::vector<mp::ExRec> ex_host;          // sorted by (window, row); complete only after ex_fetch()
    // [r6] the records leave the device BESIDE the histogram launch: mp_build_windows only notes how many there are; whoever needs ex_host
    // first — mp_get_exceptions on the caller's helper thread, the device gate's per-window counts (unique.hip) — copies them into ex_raw
    // on ex_stream and sorts: ex_fetch(), one at a time under ex_mu
    std::vector<mp::ExRec> ex_raw;           // as they come off the device (kept: its pages exist the second time)
    hipStream_t ex_stream = nullptr;
    int ex_pending = 0;                      // records on the device (c->ex), not yet in ex_host
    std::mutex ex_mu;
    int32_t *extra_off = nullptr;
    uint32_t *extra_words = nullptr;
    int n_extra = 0;
    // patch planes: the patch rows and the IUPAC expansion rows of every window as one-hot planes of their own
    // (position-major, like `cols` but per window), so the bit-sliced kernels cover them too; built on demand
    std::vector<int32_t> h_patch_off, h_extra_off;       // host copies of the two per-window offset tables
    uint32_t *pplanes = nullptr, *pvalid = nullptr;
    uint32_t *qplanes = nullptr, *qvalid = nullptr;      // plain column slices of the patch-list rows (sliding evaluation), same layout
    bool qp_dirty = true;
    mp::PatchWin *pwin = nullptr;                        // [W]
    size_t pp_words = 0, pv_words = 0;
    int max_npw = 0;                                     // widest window, in 32-row words
    bool pp_dirty = true;
    // unique: global hash tables [W][g_slots] (k <= 21) behind the per-workgroup LDS tables
    unsigned long long *g_key = nullptr;
    // [r6] MP_HIST_CM64 = 2 (default): a slot's count and first row are the two halves of ONE 64-bit word of g_cnt (count << 32 | ~first row; zero =
    // empty; g_min is a one-element stand-in) — the flush's unreturned add and max land on one line where two arrays cost two (hist2_kernel 5.77 ->
    // 5.41 ms at 10^6 rows, compact_kernel 0.15 -> 0.125; profiles/r06_hist_experiments.txt).  0: two arrays; 1: one word updated by a CAS loop.
    uint32_t *g_cnt = nullptr, *g_min = nullptr;
    uint32_t *g_gap = nullptr;               // k = 22..31: gap words of the keys that carry a gap (unique.hip)
    int32_t *g_idx = nullptr;                // dense index of a slot's entry inside its window (labels)
    int g_slots = 0;
    long long u_cap = 0, u_n = 0;
    uint32_t *u_b0 = nullptr, *u_b1 = nullptr, *u_g = nullptr;
    int32_t *u_count = nullptr, *u_first = nullptr, *labels = nullptr, *u_over = nullptr, *u_wcount = nullptr;
    int64_t *u_wbase = nullptr;
    unsigned long long *u_total = nullptr;
    std::vector<int64_t> h_wbase;
    std::vector<int32_t> h_wcount;
    // the entropy gate on the device (mp_set_entropy_gate): armed threshold (0: off), and after the histograms which windows it
    // rejected — their entries are neither compacted nor read back nor planned (mp_plan_create_streamed)
    double gate_threshold = 0;
    std::vector<uint8_t> h_wskip;
    // dimer tables (Loss decisions, deltaG constants): cached across calls, re-uploaded only when the contents change
    uint8_t *dm_loss = nullptr;
    double *dm_dg = nullptr;
    std::vector<uint8_t> dm_loss_host;
    std::vector<double> dm_dg_host;
    // device-resident coverage masks of the last mp_eval_masks(_resident): [n_masks][n_pad/64] each
    unsigned long long *mask_f = nullptr, *mask_r = nullptr;
    size_t mask_words = 0;
    int n_masks = 0;
    // eval staging
    int n_cand = 0, n_items = 0, n_padded = 0;
    mp::EvalItem *items = nullptr;
    uint4 *cand_n = nullptr;
    uint32_t *cand_symT = nullptr, *cand_diff = nullptr, *chain_events = nullptr;
    mp::ChainItem *chain_items = nullptr;
    // eval_chain_x_kernel (evalx.hpp): chain items of primers of 32..63 bases / of v = 4, 5 — beside the row-per-lane arrays above
    void *x_items = nullptr;
    uint32_t *x_events = nullptr;
    int32_t *x_cand_out = nullptr;
    int x_n = 0, x_n_events = 0;
    int32_t *table_ids = nullptr;
    int n_chain = 0, n_table = 0, n_events = 0, max_steps = 0;     // max_steps: members of the longest chain item
    // host copies of the staged chain items (evalslide.hip builds its plan from them)
    std::vector<mp::ChainItem> h_chains;     // host copies of the chain items (ascending windows) and their events
    std::vector<uint32_t> h_events;
    std::vector<int32_t> h_cand_out;         // output slot of every padded candidate
    uint32_t *chain_prog = nullptr;          // fetch programs of the chain items (evalprog.hip), chains of up to 8 members only
    size_t chain_prog_n = 0;
    int prog_shape = -1;                     // eval_prog_kernel shape the programs were written for (-1: eval_chain_kernel runs the chains)
    // sliding evaluation (evalslide.hip): plan of the staged chain items; slide_items = 0: the first-pass kernels run every item
    mp::SlideBand *slide_bands = nullptr;
    uint32_t *slide_iters = nullptr, *slide_recs = nullptr;
    size_t slide_n_iters = 0;
    int slide_items = 0, slide_n_bands = 0, slide_max_items = 0, slide_ns = 0, slide_gw = 0;
    uint32_t slide_spos = 0, slide_fmask = 0, slide_rmask = 0, slide_fpos = 0, slide_rpos = 0;
    bool slide_fast = false;                 // the strict positions as two-bit counts per side (slidecore.hpp FAST)
    mp::ChainItem *chain_rest = nullptr;     // the chain items the plan leaves to the first-pass kernel
    mp::ChainItem *chain_slid = nullptr;     // the chain items that slide (slide_items of them): the patch pass subtracts their plain slices
    int n_rest = 0, rest_max_steps = 0;
    unsigned launch_seq = 0;                 // launches since the last timing reset
    int32_t *cand_out = nullptr;
    uint64_t sF = 0, sR = 0;
    unsigned long long *tmp_out = nullptr;
    int tmp_out_n = 0;
    struct PoolBlock { void *p; size_t bytes; hipEvent_t released; };     // `released`: recorded on the context's stream when the block came back
    std::vector<hipEvent_t> pool_events;         // events of blocks that were handed out again (reused by the next release)
    std::unordered_map<void *, size_t> pool_live; // blocks of dev_alloc that are out: the size they were requested with
    std::vector<PoolBlock> pool;                 // released device blocks, oldest first (pool_take / pool_give, api.hip)
    size_t pool_bytes = 0;
    long long pool_hits = 0, pool_misses = 0;
    int pool_on = -1;                            // -1: MP_DEVICE_POOL not read yet
    std::mutex pool_mu;
    unsigned long long *stats_buf = nullptr;     // mp_window_stats: counters of the last call's size, kept (a hipMalloc + hipFree pair per call otherwise)
    size_t stats_buf_n = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_busy, ev_free;
    double ev_ms = 0;
    std::vector<float> ev_samples, ev_last;     // per-launch durations since the last reset / as of the last mp_eval_timing call
    int ev_n = 0;
    int eval_variant = 0;
    // host staging area of the streamed planning (mp_plan_create_streamed): kept for the context's life — handing 58 MB back to the
    // kernel and faulting them in again costs more than the copy that fills them
    uint8_t *h_stage = nullptr;              // host_map() memory
    size_t h_stage_bytes = 0;
    bool h_stage_pinned = false;             // registered with the runtime (hipHostRegister): copies into it are plain DMA
    uint8_t *h_stats = nullptr;              // mp_window_stats_begin: the counters' landing buffer (host_map, registered)
    size_t h_stats_bytes = 0, stats_pending_f = 0, stats_pending_t = 0;       // pending: counters of a begin nobody has ended yet
    bool h_stats_pinned = false;
    hipEvent_t stats_ev = nullptr;
    uint8_t *h_ring = nullptr;               // mp_load_msa_fasta: 3 x 32 MB transfer buffers (host_map, registered), kept for the context's life
    bool h_ring_pinned = false;
    hipEvent_t h_ring_ev[3] = {nullptr, nullptr, nullptr};
    // the resident sequence store (mp_seq_load, scan.hip): the unaligned database of the PCR / k-mismatch scans
    uint8_t *sq_bytes = nullptr;             // the characters as loaded (fall-back paths only)
    int64_t *sq_roff = nullptr;              // [sq_n + 1] byte offsets
    unsigned long long *sq_code = nullptr, *sq_flag = nullptr;   // [sq_words] 32 bases per word: 2-bit codes / {not ACGT, lower case} flags
    int64_t *sq_woff = nullptr;              // [sq_n + 1] word offsets
    int32_t sq_n = 0;
    size_t sq_total = 0, sq_words = 0;
    std::vector<int64_t> sq_roff_host;
    // row-shard collectives (comm.hip): an RCCL communicator (ncclComm_t) when n_ranks > 1
    void *comm = nullptr;
    int n_ranks = 0, rank = 0;               // n_ranks 0: mp_comm_init has not run
    uint8_t *comm_scratch = nullptr;
    size_t comm_scratch_n = 0;
};

namespace mp {


inline size_t wsz(const mp_ctx *c) { return c->wide ? 2 : 1; }      // uint32 units per window word

inline int fail(mp_ctx *c, int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCK(c, call)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return fail((c), MP_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(e_));  \
    } while (0)

// Device blocks a stage has released wait in the context for the next request of exactly their size (api.hip): a core step allocates
// and frees ~40 blocks, hipMalloc costs 20-100 us and hipFree waits for the device on top — and a worker that runs one alignment after
// the other (or a bench that repeats one) asks for the same sizes again.  Everything the library launches is ordered on the context's
// stream or synchronised before a block is released, so a block can change hands without the wait hipFree implied.  [r6, advisor] That
// held only for users ON the context's stream: a blocking copy on the null stream into a recycled block is not ordered against a
// non-blocking stream the caller passed through mp_set_stream.  A released block now carries an event recorded on the context's stream
// and whoever takes it waits for that event first (the host waits: after it, a copy or kernel on ANY stream may touch the block).  At
// most kPoolBlocks blocks / kPoolBytes bytes wait per context — the byte limit is divided by the number of live contexts on the device
// (eight --batch-workers contexts used to hold 3 GB each) — the oldest go back to the runtime first; mp_destroy empties the pool, a
// failed hipMalloc empties the pools of EVERY context of the process (pool_drain_all); MP_DEVICE_POOL=0 switches it off.
void *pool_take(mp_ctx *c, size_t bytes);               // a waiting block of exactly `bytes`, or null
bool pool_give(mp_ctx *c, void *p, size_t bytes);        // false: not taken (the caller frees it)
void pool_drain(mp_ctx *c);
void pool_drain_all();                                  // every live context of the process (a failed hipMalloc: what waits anywhere may be what is missing)
void pool_register(mp_ctx *c, bool alive);              // mp_create / mp_destroy
// the size a block was REQUESTED with is kept by the context (a block goes back into the pool under that size, whatever count the
// releasing call site passes: a site that got its count wrong used to skew mp_device_bytes — with the pool it would hand out a short block)
void pool_note(mp_ctx *c, void *p, size_t bytes);
size_t pool_forget(mp_ctx *c, void *p, size_t claimed);  // the recorded size (0: not a block of dev_alloc); a differing claim is reported under MP_TRACE

template <typename T>
int dev_alloc(mp_ctx *c, T **p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    if (void *q = pool_take(c, n * sizeof(T))) {
        *p = (T *)q;
        pool_note(c, q, n * sizeof(T));
        c->bytes += (int64_t)(n * sizeof(T));
        return MP_OK;
    }
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e != hipSuccess) {                       // (what waits in the pool may be what is missing)
        (void)hipGetLastError();
        pool_drain_all();
        e = hipMalloc((void **)p, n * sizeof(T));
    }
    if (e != hipSuccess) return fail(c, MP_ERR_NOMEM, "hipMalloc(%zu bytes): %s", n * sizeof(T), hipGetErrorString(e));
    pool_note(c, (void *)*p, n * sizeof(T));
    c->bytes += (int64_t)(n * sizeof(T));
    return MP_OK;
}

template <typename T>
void dev_free(mp_ctx *c, T **p, size_t n) {
    if (*p) {
        const size_t claimed = (n ? n : 1) * sizeof(T), real = pool_forget(c, (void *)*p, claimed);
        if (!real || !pool_give(c, (void *)*p, real)) (void)hipFree(*p);
        c->bytes -= (int64_t)(real ? real : claimed);
        *p = nullptr;
    }
}

// Several arrays set to a 32-bit pattern by ONE kernel launch (a stage's counters, cursors and tables used to cost one runtime fill
// dispatch each, ~6 us apiece).  Sizes in bytes, multiples of 4; pointers from hipMalloc.  api.hip
struct FillSeg { void *p; size_t bytes; uint32_t value; };
constexpr int kMaxFillSegs = 8;
int fill_segments(mp_ctx *c, const FillSeg *segs, int n);

// A host buffer that is about to receive a large copy from the device and has never been touched: its pages are faulted in here on
// several threads (one write per page; the contents are about to be overwritten).  58 MB come off the device in 1.1 ms once the pages
// exist and in 4.7-6.7 ms when the copy itself has to fault them in one after the other (tools/ubench/d2h_bench.hip) — that, not the
// transfer, was the read-back time of the histogram entries.  api.hip
void prefault_host(void *p, size_t bytes);
// anonymous host memory marked for transparent huge pages (what numpy does for its large arrays): 2 MB faults instead of 4 KB ones —
// 66 MB are faulted in on 16 threads in ~1.3 ms with it and in ~4.7 ms without.  host_unmap releases it.  api.hip
void *host_map(size_t bytes);
void host_unmap(void *p, size_t bytes);

// Registered memory (hipHostRegister: the copy is one DMA at the link's rate instead of a walk through the runtime's staging buffers —
// 58 MB: 0.3 ms to register pages that exist + 1.1 ms to copy, against 3.2-3.4 ms staged, tools/ubench/d2h_bench.hip) is used for ONE
// thing: the library's own staging area of the streamed planning (h_stage: a fresh mapping, registered once, unregistered before it is
// unmapped).  Registering the CALLER's buffers for the duration of a transfer was tried for the residue bytes and the blocking entry
// read-back (6.5 -> 3.7 ms, 6.0 -> 4.0 ms) and withdrawn: one run of the GPU suite in four died with SIGABRT inside mp_load_msa at such
// a registration of memory owned by the Python allocator, without a message — not understood, so not shipped.

// release the device arrays of one stage (and of every stage that depends on it); api.hip
int ex_fetch(mp_ctx *c);         // windows.hip: ex_host complete (see mp_ctx::ex_pending)
void free_eval(mp_ctx *c);
void free_slide(mp_ctx *c);
void free_comm(mp_ctx *c);
void free_unique(mp_ctx *c);
void free_seq(mp_ctx *c);        // scan.hip
void free_windows(mp_ctx *c);
void free_msa(mp_ctx *c);
// per-translation-unit device constants (called by mp_create on the context's device)
int pack_init();
int dimer_init();

}  // namespace mp
