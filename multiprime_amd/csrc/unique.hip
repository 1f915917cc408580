// unique.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of
// include/mprime.h.  Per-window k-mer histograms (mp_window_unique), straight from the bit planes.
//
// k <= 21 (3k bits fit one u64 key) — hist_kernel: a workgroup owns (window, slice of rows).  Every thread derives its
//   row's k-mer from the eight plane words that cover the window (winwords.hpp; next iteration's words are already in
//   flight), the wave folds the k-mer of its first lane (conserved windows put the same k-mer in almost every lane) and
//   the remaining lanes insert theirs in parallel into an LDS table {key, count, first row} with 64-bit ds_cmpst.  The LDS
//   table is a write-combining front of the window's table in HBM: at the end of the slice its entries are merged into the
//   global table with 64-bit CAS / add / min atomics; a key that finds neither itself nor a free slot within 24 probes of its
//   hash (more distinct k-mers in a slice than the table takes) goes to the global table directly — any number of distinct
//   k-mers per slice, and no barrier in the row loop ([r4]; rounds 2-3 met at one per iteration to decide about a mid-way flush:
//   0.77 -> 0.72 ms).  compact_kernel then lays the occupied slots out as per-window entry segments.
//   (Round 1 stored a [W][Npad] u64 array of window words first and ran ONE workgroup per window over it: 1.5 s at
//   10^6 rows.)
// k >= 22 — unique_kernel: one workgroup per window, LDS table of representative rows; keys are compared by
//   re-deriving the representative's k-mer.  Same algorithm as round 1, minus the stored window words.
#include <atomic>
#include <memory>
#include <thread>

#include "common.hpp"
#include "planstream.hpp"
#include "workers.hpp"
#include "winwords.hpp"

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
__device__ inline uint32_t hash3(uint64_t a, uint64_t b, uint64_t c) {      // 64-bit window words (k > 31): the high halves fold in first
    return hash3((uint32_t)a ^ ((uint32_t)(a >> 32) * 0x27D4EB2Fu), (uint32_t)b ^ ((uint32_t)(b >> 32) * 0x165667B1u),
                 (uint32_t)c ^ ((uint32_t)(c >> 32) * 0x9E3779B1u));
}
// three 24-bit multiplies (full rate on gfx950; a 32-bit v_mul_lo is quarter rate and the first version's splitmix hash was a
// fifth of the kernel's VALU time) over the key's bits 0-23 / 24-47 / 48-63, two folds: on the bench alignment's k-mers the LDS
// insert needs 1.66 probe rounds per wave and the global table 1.02 probes per key, the same as with two 32-bit multiplies
__device__ inline uint32_t hash64(unsigned long long x) {
    uint32_t h = __umul24((uint32_t)x & 0xFFFFFFu, 0x9E3779u) ^ __umul24((uint32_t)(x >> 24) & 0xFFFFFFu, 0x85EBCBu) ^
                 __umul24((uint32_t)(x >> 48), 0xC2B2AFu);
    h ^= h >> 15; h ^= h >> 7;
    return h;
}

constexpr unsigned long long kNoKey = ~0ull;
constexpr uint32_t kNoGap = 0xFFFFFFFFu;  // a slot's gap word before the first flagged key claims it (a gap word has k <= 31 bits)
constexpr unsigned long long kGapFlag = 1ull << 62;
constexpr int kLdsSlots = 2048;           // 32 KB of LDS per workgroup -> 4-5 workgroups per CU

// Key forms.  k <= 21 (WIDE = false): the 3k bits b0 | b1 << k | g << 2k ARE the key.  k = 22..31 (WIDE = true, round 6): the key is the
// 2k base bits b0 | b1 << k (<= 62 bits) plus, for the few rows with a gap inside the window, bit 62 and the gap word g in a u32 BESIDE
// the slot: a slot is claimed in two steps — 64-bit CAS on the key word, then, for a flagged key only, a 32-bit CAS of the gap word from
// kNoGap.  Whoever sets the gap word owns the slot; a flagged key that finds another gap word there treats the slot as taken by another
// key and probes on.  Probe sequences are deterministic and slots are never released, so every (key, gap) lives in exactly one slot.
// A gap position carries b0 = b1 = 0 (fast_from_slices), so (key bits, g) determines the k-mer.  Unflagged keys — 96 % of the rows —
// never touch the gap words: their cost is that of the narrow form.
template <bool WIDE>
__device__ inline void make_key(uint32_t b0, uint32_t b1, uint32_t g, int k, unsigned long long &key, uint32_t &gap) {
    if (WIDE) {
        key = (unsigned long long)b0 | ((unsigned long long)b1 << k) | (g ? kGapFlag : 0ull);
        gap = g;
    } else {
        key = (unsigned long long)b0 | ((unsigned long long)b1 << k) | ((unsigned long long)g << (2 * k));
        gap = 0;
    }
}

struct HistArgs {
    MsaArgs M;
    int p0, k, n_win, rows_per_block, n_slices, win_per_xcd;
    unsigned long long *g_key;            // [W][g_slots]
    uint32_t *g_cnt, *g_min;
    uint32_t *g_gap;                      // [W][g_slots] gap words of flagged keys (WIDE), else null
    int g_slots;
    int32_t *g_over;                      // [W] 1 = the global table of the window is too small
    const int32_t *patch_off;             // [W+1] slow pairs of every window (mp_build_windows): rows and window words
    const int32_t *patch_rows;
    const uint32_t *patch_words;
    unsigned long long *prof;             // MP_HIST_PROF: [workgroup][8] shader-clock stamps of hist_kernel's phases (null: none)
};

// the scope of the atomics on the windows' tables in HBM (experiment knob: __HIP_MEMORY_SCOPE_WORKGROUP executes them in the XCD's L2)
#ifndef MP_HIST_SCOPE
#define MP_HIST_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
__device__ inline unsigned long long g_cas(unsigned long long *p, unsigned long long expect, unsigned long long val) {
    __hip_atomic_compare_exchange_strong(p, &expect, val, __ATOMIC_RELAXED, __ATOMIC_RELAXED, MP_HIST_SCOPE);
    return expect;
}
__device__ inline uint32_t g_cas(uint32_t *p, uint32_t expect, uint32_t val) {
    __hip_atomic_compare_exchange_strong(p, &expect, val, __ATOMIC_RELAXED, __ATOMIC_RELAXED, MP_HIST_SCOPE);
    return expect;
}
// MP_HIST_CM64 (common.hpp; default 2): count and first row of a slot in ONE 64-bit word (count << 32 | ~first row: zero = empty).  2: its halves
// updated by two unreturned 32-bit atomics (add, max) on one line; 1 (experiment): by a compare-and-swap loop; 0: two arrays, two lines
__device__ inline void g_count_min(uint32_t *cnt_base, uint32_t *min_base, size_t at, uint32_t cnt, uint32_t row, bool fresh);
__device__ inline void g_add(uint32_t *p, uint32_t v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, MP_HIST_SCOPE); }
__device__ inline void g_min(uint32_t *p, uint32_t v) { (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, MP_HIST_SCOPE); }

__device__ inline void g_count_min(uint32_t *cnt_base, uint32_t *min_base, size_t at, uint32_t cnt, uint32_t row, bool fresh) {
#if MP_HIST_CM64 == 2
    // the two words of the slot's 64-bit (count << 32 | ~first row) by two UNRETURNED 32-bit atomics on ONE line (add on the high word, max on the low)
    uint32_t *cm = cnt_base + 2 * at;
    (void)__hip_atomic_fetch_max(cm, ~row, __ATOMIC_RELAXED, MP_HIST_SCOPE);
    g_add(cm + 1, cnt);
    (void)min_base; (void)fresh;
#elif MP_HIST_CM64
    unsigned long long *cm = reinterpret_cast<unsigned long long *>(cnt_base) + at;
    unsigned long long old = fresh ? 0ull : *cm;
    for (;;) {
        const uint32_t lo = (uint32_t)old, nr = ~row;
        const unsigned long long neu = ((unsigned long long)((uint32_t)(old >> 32) + cnt) << 32) | (unsigned long long)(lo > nr ? lo : nr);
        const unsigned long long got = g_cas(cm, old, neu);
        if (got == old) break;
        old = got;
    }
    (void)min_base;
#else
    (void)fresh;
    g_add(cnt_base + at, cnt);
    g_min(min_base + at, row);
#endif
}

// does slot h (whose key word answered `old` to the claim of `key`, with kNoKey already replaced by key) hold (key, gap)?
template <bool WIDE>
__device__ inline bool slot_holds(unsigned long long old, unsigned long long key, uint32_t *gap_words, size_t h, uint32_t gap) {
    if (old != key) return false;
    if (!WIDE || !(key & kGapFlag)) return true;
    const uint32_t og = atomicCAS(&gap_words[h], kNoGap, gap);
    return og == kNoGap || og == gap;
}

// merge one (key, count, first row) into the window's global table, probing from slot h
template <bool WIDE>
__device__ inline void global_insert_from(const HistArgs &A, int w, unsigned long long key, uint32_t gap, uint32_t cnt, uint32_t row, uint32_t h, int probe0) {
    const uint32_t mask = (uint32_t)A.g_slots - 1u;
    const size_t base = (size_t)w * A.g_slots;
    unsigned long long *K = A.g_key + base;
    for (int probe = probe0; probe < A.g_slots; probe++) {
        unsigned long long old = K[h];
        bool fresh = false;
        if (old == kNoKey) {
            old = g_cas(&K[h], kNoKey, key);
            if (old == kNoKey) { old = key; fresh = true; }     // claimed (occupied slots are counted afterwards, table_sums_kernel)
        }
        if (slot_holds<WIDE>(old, key, A.g_gap, base + h, gap)) {
            g_count_min(A.g_cnt, A.g_min, base + h, cnt, row, fresh);
            return;
        }
        h = (h + 1) & mask;
    }
    A.g_over[w] = 1;
}
template <bool WIDE>
__device__ inline void global_insert(const HistArgs &A, int w, unsigned long long key, uint32_t gap, uint32_t cnt, uint32_t row) {
    global_insert_from<WIDE>(A, w, key, gap, cnt, row, hash64(key) & ((uint32_t)A.g_slots - 1u), 0);
}

// The workgroup's LDS table into the window's global table.  A thread owns SLOTS / kBlock slots and takes them through the merge four
// at a time: their first-probe reads together, then the claims, then the (unreturned) count / first-row atomics — two global round
// trips per four slots instead of two or three per occupied slot one after the other (the flush is a third of a workgroup's life and
// all of it is global latency); a slot whose first probe meets another key walks on alone.
template <int SLOTS, bool WIDE>
__device__ inline void flush_table(const HistArgs &A, int w, unsigned long long *s_key, uint32_t *s_cnt, uint32_t *s_min, uint32_t *s_gap) {
#ifndef MP_HIST_FLUSH_S
#define MP_HIST_FLUSH_S 4
#endif
    constexpr int S = MP_HIST_FLUSH_S;
    static_assert(SLOTS % (S * kBlock) == 0, "whole groups of slots per thread");
    const uint32_t mask = (uint32_t)A.g_slots - 1u;
    const size_t base = (size_t)w * A.g_slots;
    unsigned long long *K = A.g_key + base;
#pragma unroll 1
    for (int i0 = threadIdx.x; i0 < SLOTS; i0 += S * kBlock) {
        unsigned long long key[S], old[S];
        uint32_t cnt[S], mn[S], h[S], gap[S];
#pragma unroll
        for (int u = 0; u < S; u++) {
            const int i = i0 + u * kBlock;
            key[u] = s_key[i]; cnt[u] = s_cnt[i]; mn[u] = s_min[i];
            s_key[i] = kNoKey; s_cnt[i] = 0; s_min[i] = kEmpty;
            gap[u] = 0;
            if (WIDE) { gap[u] = s_gap[i]; s_gap[i] = kNoGap; }
            h[u] = hash64(key[u]) & mask;
        }
#pragma unroll
        for (int u = 0; u < S; u++) old[u] = key[u] != kNoKey ? K[h[u]] : 0ull;
        bool fresh[S];
#pragma unroll
        for (int u = 0; u < S; u++) {
            fresh[u] = false;
            if (key[u] != kNoKey && old[u] == kNoKey) {
                old[u] = g_cas(&K[h[u]], kNoKey, key[u]);
                if (old[u] == kNoKey) { old[u] = key[u]; fresh[u] = true; }
            }
        }
#pragma unroll
        for (int u = 0; u < S; u++)
            if (key[u] != kNoKey) {
                if (slot_holds<WIDE>(old[u], key[u], A.g_gap, base + h[u], gap[u])) {
                    g_count_min(A.g_cnt, A.g_min, base + h[u], cnt[u], mn[u], fresh[u]);
                } else {
                    global_insert_from<WIDE>(A, w, key[u], gap[u], cnt[u], mn[u], (h[u] + 1) & mask, 1);
                }
            }
    }
}

__device__ inline unsigned long long readlane64(unsigned long long x, int lane) {
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, lane) |
           ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), lane) << 32);
}
__device__ inline unsigned long long uniform64(unsigned long long x) {
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) |
           ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32);
}

// SLOTS x 16 (20 with gap words) bytes of LDS per workgroup decide how many workgroups a CU holds (160 KB: 5 at 2048 slots).  kMaxProbe
// bounds a key's walk through the LDS table: beyond it the key is not there, and if no free slot turned up either it goes to the table
// in HBM.  FOLDS: rounds of wave-level folding in front of the LDS table (below).
template <int SLOTS, bool WIDE, int FOLDS>
__global__ __launch_bounds__(kBlock) void hist_kernel(const HistArgs A) {
    constexpr int kLdsSlots = SLOTS, kMaxProbe = 24;
    __shared__ unsigned long long s_key[kLdsSlots];
    __shared__ uint32_t s_cnt[kLdsSlots];
    __shared__ uint32_t s_min[kLdsSlots];
    __shared__ uint32_t s_gap[WIDE ? kLdsSlots : 1];
    __shared__ int s_used, s_flag;                        // slots claimed since the last flush; the workgroup's decision to flush
    // workgroup b runs on XCD b % 8 and every XCD has its own L2: an XCD owns a band of consecutive windows and walks it
    // window-fastest, so the workgroups resident on it at any time read the same few 32-column chunks of one row slice
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int wl = q % A.win_per_xcd, slice = q / A.win_per_xcd;
    const int w = xcd * A.win_per_xcd + wl;
    if (w >= A.n_win || slice >= A.n_slices) return;
    const int k = A.k;
    const uint32_t kmask = (1u << k) - 1u;
    const int p = A.p0 + w;
    const size_t np = (size_t)A.M.n_pad;
    const int r0 = slice * A.rows_per_block;
    const int r1 = min(r0 + A.rows_per_block, A.M.n_pad);
    auto stamp = [&](int i) { if (A.prof && threadIdx.x == 0) A.prof[(size_t)blockIdx.x * 8 + i] = clock64(); };
    stamp(0);
    for (int i = threadIdx.x; i < kLdsSlots; i += kBlock) {
        s_key[i] = kNoKey; s_cnt[i] = 0; s_min[i] = kEmpty;
        if (WIDE) s_gap[i] = kNoGap;
    }
    if (threadIdx.x == 0) { s_used = 0; s_flag = 0; }
    __syncthreads();
    stamp(1);
    const int lane = threadIdx.x & 63;
    const uint32_t *P = A.M.planes + ((size_t)(p >> 5) * 4) * np;
    auto flush = [&]() { flush_table<SLOTS, WIDE>(A, w, s_key, s_cnt, s_min, s_gap); };
    // A thread takes FOUR consecutive rows per iteration: the eight plane words of the four rows arrive as eight 16-byte buffer loads
    // (row byte offset in a vector register, plane offset in a scalar one).  One-word loads are what round 2 used: a CU returns them at
    // 20 B/clk (`ubench`: 11 TB/s over the chip, against 31 TB/s for 16-byte loads), and 18.8 M of them per launch were half the
    // kernel's time on that path alone.
    constexpr int RPT = 4;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(P), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(A.M.rlen), 0, 0x7FFFFFFF, 0x00020000);
    const int plane_bytes = (int)(np * 4);
    // two register sets take turns: while one iteration's rows are hashed the next one's plane words arrive in the other set
    // (a single set copied at the top of every iteration cost 36 register moves per thread and iteration)
    struct Rows { u32x4 w[8], len; };
    auto fetch = [&](Rows &R, int r4) {
        if (r4 < r1) {                                        // r0, r1 and n_pad are multiples of 256: a thread's four rows exist together
#pragma unroll
            for (int j = 0; j < 8; j++) R.w[j] = __builtin_amdgcn_raw_buffer_load_b128(prs, r4 * 4, j * plane_bytes, 0);
            R.len = __builtin_amdgcn_raw_buffer_load_b128(lrs, r4 * 4, 0, 0);
        }
    };
    // The wave's running consensus: the k-mer that led the largest group so far (wave-uniform, scalar registers).  [r6] What bound this
    // kernel was not instruction issue (VALU 24 % busy) but the LDS unit serialising atomics on ONE address: rounds 2-5 folded only the
    // k-mer of a step's FIRST live lane, and whenever that lane did not carry the window's consensus (43 % of the steps on the bench
    // alignment) ~36 lanes sent a CAS, an add and a min to the same slot, one after the other.  Now a step folds the lanes that carry
    // the running consensus (no readlane needed), then FOLDS more groups led by the first lane still pending; only what is left — lanes
    // with mostly distinct k-mers — goes to the table one lane at a time.
#ifndef MP_HIST_CONS_CARRY
#define MP_HIST_CONS_CARRY 0          // 1: the consensus is handed from row index to row index inside a step (serialises the four chains)
#endif
    unsigned long long cons = kNoKey;
    auto hash_rows = [&](const Rows &R, int r4) {
        const unsigned long long cons_in = cons;
        (void)cons_in;
        const u32x4 (&cw)[8] = R.w;
        const u32x4 len4 = R.len;
        const bool mine = r4 < r1;
        // (1) straight-line: the window words of the thread's four rows (a lane without rows computes on stale registers and is masked
        // by `ok`); group leaders (the lowest lane of a group = its lowest row) carry the group's count
        unsigned long long key[RPT];
        uint32_t cnt[RPT], h[RPT], gap[RPT];
        bool todo[RPT];
        uint32_t claims = 0;                                  // slots this lane claimed in this step (<= RPT)
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            uint32_t b0, b1, g;
            // plain column slices only: the repaired / IUPAC / ragged rows of the window come from the patch list below
            const bool plain = fast_words(p, k, kmask, (int)len4[u], cw[0][u], cw[1][u], cw[2][u], cw[3][u], cw[4][u], cw[5][u], cw[6][u], cw[7][u], b0, b1, g);
            const bool ok = plain & mine & (r4 + u < A.M.n_rows);
            make_key<WIDE>(b0, b1, g, k, key[u], gap[u]);
            unsigned long long pending = __ballot(ok);
            const unsigned long long cref = MP_HIST_CONS_CARRY ? cons : cons_in;   // (the four row indices of a step are independent chains)
            const bool s0 = ok & (key[u] == cref);                      // (a flagged key never equals the consensus: it is adopted unflagged only)
            const unsigned long long g0 = __ballot(s0);
            pending &= ~g0;
            const int lead0 = g0 ? __ffsll((long long)g0) - 1 : -1;
            uint32_t c = 1u;
            bool go = false;
            if (lane == lead0) { c = (uint32_t)__popcll(g0); go = true; }
            int best_n = (int)__popcll(g0);
            unsigned long long best_key = cref;
#pragma unroll
            for (int f = 0; f < FOLDS; f++) {
                const int lead = pending ? __ffsll((long long)pending) - 1 : 0;
                const unsigned long long kf = readlane64(key[u], lead);
                bool same = ((pending >> lane) & 1ull) && key[u] == kf;
                if (WIDE) same = same && gap[u] == (uint32_t)__builtin_amdgcn_readlane((int)gap[u], lead);
                const unsigned long long grp = __ballot(same);             // (every lane votes: not inside a conditional)
                pending &= ~grp;
                const int n = (int)__popcll(grp);
                if (lane == lead && n) { c = (uint32_t)n; go = true; }
                if (n > best_n && !(WIDE && (kf & kGapFlag))) { best_n = n; best_key = kf; }
            }
            if (MP_HIST_CONS_CARRY || u == 0) cons = best_key;
            cnt[u] = c;
            todo[u] = go | (bool)((pending >> lane) & 1ull);
            h[u] = hash64(key[u]) & (kLdsSlots - 1);
        }
        // (2) the first probes of all four rows leave together (four LDS round trips overlap instead of following each other); a row
        // whose first slot holds another key walks on alone
        unsigned long long old[RPT];
#pragma unroll
        for (int u = 0; u < RPT; u++) old[u] = todo[u] ? atomicCAS(&s_key[h[u]], kNoKey, key[u]) : key[u];
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            if (todo[u]) {
                const uint32_t row = (uint32_t)(r4 + u);
                if (old[u] == kNoKey) { old[u] = key[u]; claims++; }
                uint32_t hh = h[u];
                // a key sits within kMaxProbe slots of its hash or not in the LDS table at all: when the walk finds neither the key
                // nor a free slot (a workgroup's rows hold more distinct k-mers than the table takes), the key goes to the window's
                // table in HBM directly.  No fill count, no check, no barrier in the row loop (round 4; until then the workgroup met at
                // a barrier every iteration to see whether the table had to be flushed: the waves of a workgroup ran in lock step)
                bool here = slot_holds<WIDE>(old[u], key[u], s_gap, hh, gap[u]);
                for (int probe = 0; !here && probe < kMaxProbe; probe++) {
                    hh = (hh + 1) & (kLdsSlots - 1);
                    old[u] = atomicCAS(&s_key[hh], kNoKey, key[u]);
                    if (old[u] == kNoKey) { old[u] = key[u]; claims++; }
                    here = slot_holds<WIDE>(old[u], key[u], s_gap, hh, gap[u]);
                }
                if (here) {
                    atomicAdd(&s_cnt[hh], cnt[u]);
                    atomicMin(&s_min[hh], row);
                } else {
                    global_insert<WIDE>(A, w, key[u], gap[u], cnt[u], row);
                }
            }
        }
        // the wave's claims of this step (three ballots: a lane claims at most RPT = 4 slots) -> the workgroup's fill count
        const int n_claimed = (int)__popcll(__ballot(claims & 1u)) + 2 * (int)__popcll(__ballot(claims & 2u)) + 4 * (int)__popcll(__ballot(claims & 4u));
        if (lane == 0 && n_claimed) atomicAdd(&s_used, n_claimed);
    };
    // [r6] The table is flushed MID-SLICE once it is more than kFlushAt full (checked every second step: two barriers per 2048 rows).
    // Rounds 4-5 never flushed before the end of the slice and let a key that found no free slot within kMaxProbe probes go to the table
    // in HBM itself: at 10^6 rows a 24576-row slice of a variable window holds ~3500 distinct k-mers, the 2048 slots were full after a
    // third of the slice, and from then on EVERY new k-mer walked 24 slots (24 returning LDS atomics, one after the other) before its
    // lane-at-a-time insert into HBM — the windows the entropy gate rejects were the slowest, and the k = 22 histograms took twice the
    // time of the k = 18 ones.  The batched flush moves an entry for ~30 cycles of the workgroup; a lane's own walk and insert cost ~10x.
#ifndef MP_HIST_FLUSH_16THS
#define MP_HIST_FLUSH_16THS 7
#endif
    constexpr int kFlushAt = SLOTS * MP_HIST_FLUSH_16THS / 16;
    auto maybe_flush = [&]() {
        __syncthreads();
        if (threadIdx.x == 0) s_flag = s_used > kFlushAt;
        __syncthreads();
        if (s_flag) {
            flush();
            if (threadIdx.x == 0) s_used = 0;
            __syncthreads();
        }
    };
    {
        constexpr int kStep = kBlock * RPT;
        Rows Ra, Rb;
        const int t4 = (int)threadIdx.x * RPT;
        fetch(Ra, r0 + t4);
        for (int base = r0; base < r1; base += 2 * kStep) {
            fetch(Rb, base + kStep + t4);
            hash_rows(Ra, base + t4);
            if (base + kStep >= r1) break;
            fetch(Ra, base + 2 * kStep + t4);
            hash_rows(Rb, base + kStep + t4);
            if (base + 2 * kStep < r1) maybe_flush();
        }
    }
    __syncthreads();
    stamp(2);
    if (slice == 0 && A.patch_off) {
        // the window's slow pairs (edge-gap repair, ragged end): their k-mers were derived once by repair_kernel (spreading them over
        // the window's slices was tried: every workgroup then pays the section, 764 -> 799 us)
        const int e0 = A.patch_off[w], e1 = A.patch_off[w + 1];
        for (int eb = e0; eb < e1; eb += kBlock) {
            const int e = eb + threadIdx.x;
            if (e < e1) {
                const uint32_t b0 = A.patch_words[3 * (size_t)e], b1 = A.patch_words[3 * (size_t)e + 1], g = A.patch_words[3 * (size_t)e + 2];
                if (!(g & MP_WIN_SKIP)) {
                    unsigned long long key;
                    uint32_t gap;
                    make_key<WIDE>(b0, b1, g & kmask, k, key, gap);
                    uint32_t h = hash64(key) & (kLdsSlots - 1);
                    bool placed = false;
                    for (int probe = 0; probe <= kMaxProbe && !placed; probe++) {
                        unsigned long long old = atomicCAS(&s_key[h], kNoKey, key);
                        if (old == kNoKey) old = key;
                        if (slot_holds<WIDE>(old, key, s_gap, h, gap)) { atomicAdd(&s_cnt[h], 1u); atomicMin(&s_min[h], (uint32_t)A.patch_rows[e]); placed = true; }
                        h = (h + 1) & (kLdsSlots - 1);
                    }
                    if (!placed) global_insert<WIDE>(A, w, key, gap, 1u, (uint32_t)A.patch_rows[e]);
                }
            }
        }
        __syncthreads();
    }
    stamp(3);
    flush();
    __syncthreads();
    stamp(4);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// hist2_kernel [r6] — the same (window, slice of rows) decomposition, the same tables, a different row loop.
//
// What bound hist_kernel was never found in a unit being busy (rocprofv3 at 10^6 rows: vector ALU 24 %, LDS array 21 %, scalar 33 %,
// 6.4 TB/s through the vector memory path): a wave spent ~2100 cycles per 64 rows WAITING — every step ended in returning LDS atomics
// (64-bit compare-and-swap, a divergent probe walk behind it, add, min) for the ~27 lanes whose k-mer is not the consensus, and four
// waves per SIMD cannot hide that.  Folding more groups, batching the flush, flushing mid-slice: all within noise (profiles/
// r06_hist_experiments.txt).  So the row loop no longer ends in a returning atomic at all:
//   * a wave fixes a REFERENCE k-mer (the first k-mer a fifth of a step's lanes share — the window's consensus) in scalar registers;
//   * lanes that carry it are counted with one ballot and a scalar add (57 % of the rows of the bench alignment);
//   * lanes that differ from it in exactly ONE position (30 %) bump a dense counter [position][symbol] of their wave in LDS with a
//     non-returning add and min — 8 k counters instead of a hash table for the ~54 single-substitution variants that make up most
//     of the non-consensus rows;
//   * the rest (12 %) append (k-mer, row) to a ring of their wave in LDS (position = scalar tail + lane rank: no atomic), and whenever
//     the ring holds 64 entries the wave inserts them into the workgroup's hash table with ALL lanes busy — the returning
//     compare-and-swap chain is paid once per 64 such rows instead of once per step.
// At the end of the slice a wave's reference count, dense counters and ring leftovers go through the same insert; the hash table is
// flushed into the window's table in HBM as before (and mid-slice when it fills up).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int SLOTS, bool WIDE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4))) void hist2_kernel(const HistArgs A) {
    // the ring holds what may wait (63) plus what one / two row indices append before the next drain: the wide-key form drains after every
    // row index (128 entries: 36 KB of LDS per workgroup, four per CU), the narrow one after every second
    constexpr int kMaxProbe = 24, NW = kBlock / 64, QCAP = WIDE ? 128 : 256, ND = 256;
    __shared__ unsigned long long s_key[SLOTS];
    __shared__ uint32_t s_cnt[SLOTS];
    __shared__ uint32_t s_min[SLOTS];
    __shared__ uint32_t s_gap[WIDE ? SLOTS : 1];
    __shared__ uint32_t d_cnt[NW][ND], d_min[NW][ND];         // per wave: [position * 8 + symbol] of the single-difference k-mers
    __shared__ unsigned long long q_key[NW][QCAP];            // per wave: ring of the other k-mers waiting for a dense insert
    __shared__ uint32_t q_row[NW][QCAP];
    __shared__ uint32_t q_gap[WIDE ? NW : 1][QCAP];
    __shared__ int s_used, s_flag;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int wl = q % A.win_per_xcd, slice = q / A.win_per_xcd;
    const int w = xcd * A.win_per_xcd + wl;
    if (w >= A.n_win || slice >= A.n_slices) return;
    const int k = A.k;
    const uint32_t kmask = (1u << k) - 1u;
    const int p = A.p0 + w;
    const size_t np = (size_t)A.M.n_pad;
    const int r0 = slice * A.rows_per_block;
    const int r1 = min(r0 + A.rows_per_block, A.M.n_pad);
    auto stamp = [&](int i) { if (A.prof && threadIdx.x == 0) A.prof[(size_t)blockIdx.x * 8 + i] = clock64(); };
    stamp(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < SLOTS; i += kBlock) {
        s_key[i] = kNoKey; s_cnt[i] = 0; s_min[i] = kEmpty;
        if (WIDE) s_gap[i] = kNoGap;
    }
    for (int i = lane; i < ND; i += 64) { d_cnt[wave][i] = 0; d_min[wave][i] = kEmpty; }
    if (threadIdx.x == 0) { s_used = 0; s_flag = 0; }
    __syncthreads();
    stamp(1);
    const uint32_t *P = A.M.planes + ((size_t)(p >> 5) * 4) * np;
    constexpr int RPT = 4;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(P), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(A.M.rlen), 0, 0x7FFFFFFF, 0x00020000);
    const int plane_bytes = (int)(np * 4);
    struct Rows { u32x4 w[8], len; };
    auto fetch = [&](Rows &R, int r4) {
        if (r4 < r1) {
#pragma unroll
            for (int j = 0; j < 8; j++) R.w[j] = __builtin_amdgcn_raw_buffer_load_b128(prs, r4 * 4, j * plane_bytes, 0);
            R.len = __builtin_amdgcn_raw_buffer_load_b128(lrs, r4 * 4, 0, 0);
        }
    };
    // one (k-mer, count, first row) of every lane with `go` into the workgroup's table (or, when that is full around its hash, into
    // the window's table in HBM); returns 1 when the lane claimed a slot
    auto insert = [&](bool go, unsigned long long key, uint32_t gap, uint32_t cnt, uint32_t row) -> uint32_t {
        uint32_t claimed = 0;
        if (go) {
            uint32_t hh = hash64(key) & (SLOTS - 1);
            unsigned long long old = atomicCAS(&s_key[hh], kNoKey, key);
            if (old == kNoKey) { old = key; claimed = 1; }
            bool here = slot_holds<WIDE>(old, key, s_gap, hh, gap);
            for (int probe = 0; !here && probe < kMaxProbe; probe++) {
                hh = (hh + 1) & (SLOTS - 1);
                old = atomicCAS(&s_key[hh], kNoKey, key);
                if (old == kNoKey) { old = key; claimed = 1; }
                here = slot_holds<WIDE>(old, key, s_gap, hh, gap);
            }
            if (here) {
                atomicAdd(&s_cnt[hh], cnt);
                atomicMin(&s_min[hh], row);
            } else {
                global_insert<WIDE>(A, w, key, gap, cnt, row);
            }
        }
        return claimed;
    };
    auto note_claims = [&](uint32_t claimed) {                 // (uniform control flow: every lane of the wave comes by)
        const int n = (int)__popcll(__ballot(claimed != 0));
        if (lane == 0 && n) atomicAdd(&s_used, n);
    };
    // wave-uniform state (scalar registers)
    bool have_ref = false;
    uint32_t rb0 = 0, rb1 = 0, rg = 0;                        // the reference k-mer
    uint32_t n_ref = 0, first_ref = kEmpty;                   // rows that carry it, the first of them
    uint32_t q_head = 0, q_tail = 0;                          // the wave's ring: entries [q_head, q_tail) modulo QCAP
    auto drain = [&](uint32_t n) {                            // the first n <= 64 ring entries into the table
        const uint32_t pos = (q_head + (uint32_t)lane) & (QCAP - 1);
        const bool go = (uint32_t)lane < n;
        const unsigned long long key = q_key[wave][pos];
        const uint32_t row = q_row[wave][pos];
        const uint32_t gap = WIDE ? q_gap[WIDE ? wave : 0][pos] : 0u;
        note_claims(insert(go, key, gap, 1u, row));
        q_head += n;
    };
    auto hash_rows = [&](const Rows &R, int base) {
        const int r4 = base + (int)threadIdx.x * RPT;
        const u32x4 (&cw)[8] = R.w;
#ifdef MP_HIST_EXP_LOADONLY                                  // experiment: the loads alone (every word used once, nothing counted)
        {
            uint32_t x = R.len[0];
#pragma unroll
            for (int j = 0; j < 8; j++) x ^= R.w[j][0] ^ R.w[j][1] ^ R.w[j][2] ^ R.w[j][3];
            if (x == 0x12345678u && r4 < r1) n_ref++;
            return;
        }
#endif
        const u32x4 len4 = R.len;
        const bool mine = r4 < r1;
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            uint32_t b0, b1, g;
            const bool plain = fast_words(p, k, kmask, (int)len4[u], cw[0][u], cw[1][u], cw[2][u], cw[3][u], cw[4][u], cw[5][u], cw[6][u], cw[7][u], b0, b1, g);
            const bool ok = plain & mine & (r4 + u < A.M.n_rows);
            const uint32_t row = (uint32_t)(r4 + u);
            if (!have_ref) {                                   // (uniform) adopt the first k-mer that 12 lanes of a step share
                const unsigned long long okm = __ballot(ok);
                if (okm) {
                    const int lead = __ffsll((long long)okm) - 1;
                    const uint32_t l0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, lead), l1 = (uint32_t)__builtin_amdgcn_readlane((int)b1, lead),
                                   lg = (uint32_t)__builtin_amdgcn_readlane((int)g, lead);
                    const unsigned long long same = __ballot(ok && b0 == l0 && b1 == l1 && g == lg);
                    if (__popcll(same) >= 12) { have_ref = true; rb0 = l0; rb1 = l1; rg = lg; }
                }
            }
            const uint32_t d = (b0 ^ rb0) | (b1 ^ rb1) | (g ^ rg);
            const bool is_ref = ok && have_ref && d == 0;
            const bool is_one = ok && have_ref && d != 0 && (d & (d - 1u)) == 0;
            const unsigned long long m_ref = __ballot(is_ref);
            if (m_ref) {
                n_ref += (uint32_t)__popcll(m_ref);
                const uint32_t cand = (uint32_t)(base + ((wave * 64 + (__ffsll((long long)m_ref) - 1)) * RPT) + u);
                first_ref = cand < first_ref ? cand : first_ref;
            }
            if (is_one) {
                const int j = __ffs((int)d) - 1;
                const uint32_t sym = ((b0 >> j) & 1u) | (((b1 >> j) & 1u) << 1) | (((g >> j) & 1u) << 2);
                atomicAdd(&d_cnt[wave][j * 8 + (int)sym], 1u);
                atomicMin(&d_min[wave][j * 8 + (int)sym], row);
            }
            const bool other = ok && !is_ref && !is_one;
            const unsigned long long m_oth = __ballot(other);
            if (m_oth) {
                if (other) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_oth >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_oth, 0u));
                    const uint32_t pos = (q_tail + rank) & (QCAP - 1);
                    unsigned long long key;
                    uint32_t gap;
                    make_key<WIDE>(b0, b1, g, k, key, gap);
                    q_key[wave][pos] = key;
                    q_row[wave][pos] = row;
                    if (WIDE) q_gap[WIDE ? wave : 0][pos] = gap;
                }
                q_tail += (uint32_t)__popcll(m_oth);
            }
            if (WIDE || (u & 1)) {
#pragma unroll 1
                while (q_tail - q_head >= 64u) drain(64u);
            }
#ifndef MP_HIST2_NO_SCHED_BARRIER
            __builtin_amdgcn_sched_barrier(0);                // one row index after the other: interleaving the four keeps four k-mers' worth of registers alive
#endif
        }
    };
#ifndef MP_HIST2_FLUSH_16THS
#define MP_HIST2_FLUSH_16THS 9
#endif
    constexpr int kFlushAt = SLOTS * MP_HIST2_FLUSH_16THS / 16;
#ifdef MP_HIST_EXP_NOFLUSH                                   // experiment: nothing leaves the workgroup
    auto flush = [&]() { for (int i = threadIdx.x; i < SLOTS; i += kBlock) { s_key[i] = kNoKey; s_cnt[i] = 0; s_min[i] = kEmpty; } };
#else
    auto flush = [&]() { flush_table<SLOTS, WIDE>(A, w, s_key, s_cnt, s_min, s_gap); };
#endif
    auto maybe_flush = [&]() {
        __syncthreads();
        if (threadIdx.x == 0) s_flag = s_used > kFlushAt;
        __syncthreads();
        if (s_flag) {
            flush();
            if (threadIdx.x == 0) s_used = 0;
            __syncthreads();
        }
    };
#ifndef MP_HIST2_WIDE_SINGLE
#define MP_HIST2_WIDE_SINGLE 1        // the wide-key form takes one register set (its gap words cost it a wave per SIMD otherwise: 162 -> 128 VGPRs)
#endif
    if (WIDE && MP_HIST2_WIDE_SINGLE) {
        constexpr int kStep = kBlock * RPT;
        Rows Ra;
        const int t4 = (int)threadIdx.x * RPT;
        int it = 0;
        for (int base = r0; base < r1; base += kStep, it++) {
            fetch(Ra, base + t4);
            hash_rows(Ra, base);
            if ((it & 1) && base + kStep < r1) maybe_flush();
        }
    } else {
        constexpr int kStep = kBlock * RPT;
        Rows Ra, Rb;
        const int t4 = (int)threadIdx.x * RPT;
        fetch(Ra, r0 + t4);
        for (int base = r0; base < r1; base += 2 * kStep) {
            fetch(Rb, base + kStep + t4);
            hash_rows(Ra, base);
            if (base + kStep >= r1) break;
            fetch(Ra, base + 2 * kStep + t4);
            hash_rows(Rb, base + kStep);
            if (base + 2 * kStep < r1) maybe_flush();
        }
    }
    // the wave's leftovers: ring, reference count, dense counters — all through the same insert
    if (q_tail != q_head) drain(q_tail - q_head);
    {
        unsigned long long key;
        uint32_t gap;
        make_key<WIDE>(rb0, rb1, rg, k, key, gap);
        note_claims(insert(lane == 0 && n_ref != 0, key, gap, n_ref, first_ref));
        for (int i0 = 0; i0 < 8 * k; i0 += 64) {
            const int i = i0 + lane;
            const uint32_t c = i < 8 * k ? d_cnt[wave][i] : 0u;
            const uint32_t mn = i < 8 * k ? d_min[wave][i] : kEmpty;
            const int j = i >> 3;
            const uint32_t sym = (uint32_t)i & 7u, bit = 1u << (j & 31);
            const uint32_t e0 = (rb0 & ~bit) | ((sym & 1u) ? bit : 0u), e1 = (rb1 & ~bit) | ((sym & 2u) ? bit : 0u), eg = (rg & ~bit) | ((sym & 4u) ? bit : 0u);
            make_key<WIDE>(e0, e1, eg, k, key, gap);
            note_claims(insert(c != 0, key, gap, c, mn));
        }
    }
    __syncthreads();
    stamp(2);
    if (slice == 0 && A.patch_off) {
        const int e0 = A.patch_off[w], e1 = A.patch_off[w + 1];
        for (int eb = e0; eb < e1; eb += kBlock) {
            const int e = eb + threadIdx.x;
            bool go = false;
            unsigned long long key = 0;
            uint32_t gap = 0, row = 0;
            if (e < e1) {
                const uint32_t b0 = A.patch_words[3 * (size_t)e], b1 = A.patch_words[3 * (size_t)e + 1], g = A.patch_words[3 * (size_t)e + 2];
                if (!(g & MP_WIN_SKIP)) { make_key<WIDE>(b0, b1, g & kmask, k, key, gap); row = (uint32_t)A.patch_rows[e]; go = true; }
            }
            (void)insert(go, key, gap, 1u, row);
        }
        __syncthreads();
    }
    stamp(3);
    flush();
    __syncthreads();
    stamp(4);
}


struct CompactArgs {
    const unsigned long long *g_key;
    const uint32_t *g_cnt, *g_min;
    const uint32_t *g_gap;        // gap words of flagged keys (k = 22..31), else null
    int32_t *g_idx;
    int g_slots, k;
    const int64_t *win_base;      // [W] first entry of the window's segment
    int32_t *win_cursor;          // [W]
    int parts;                    // [r6] pieces a window's table is walked in (one workgroup each: table_sums_kernel's pieces)
    const int32_t *part_base;     // [W][parts] occupied slots of the window in front of the piece (from table_sums_kernel's counts)
    uint32_t *b0, *b1, *g;
    int32_t *count, *first;
    long long cap;
};

// One pass over the tables for everything the host wants to know before it lays out the entries: the occupied slots of every window (a
// counter bumped at every claim would be ~2 x 10^6 atomics on 31 cache lines) and the entropy gate's sums (mp_set_entropy_gate):
// T = sum of the counts, S = sum of c log2 c — the entropy of the window's k-mer distribution is log2 T - S / T whatever the order of the
// terms.  [r6] Rounds 2-5 ran two kernels of ONE workgroup per window, one slot per thread and trip (count_kernel, gate_kernel: 0.32 +
// 3.3 ms at 10^6 rows, two read-backs); now a window's table is cut into `parts` pieces (grid = windows x parts), a thread has eight
// slots in flight, and the per-piece partial sums (int count, two doubles) are added up on the host: one launch, one read-back.
struct TableSums { double t, s; int32_t used, pad; };
__global__ __launch_bounds__(kBlock) void table_sums_kernel(const unsigned long long *__restrict__ g_key, const uint32_t *__restrict__ g_cnt, int g_slots,
                                                            int parts, int want_sums, TableSums *__restrict__ out) {
    constexpr int U = 8;
    __shared__ double s_t[kBlock / 64], s_s[kBlock / 64];
    __shared__ int s_n[kBlock / 64];
    const int w = blockIdx.x / parts, part = blockIdx.x % parts;
    const int per = g_slots / parts;                          // g_slots and parts are powers of two, per >= kBlock
    const size_t base = (size_t)w * g_slots + (size_t)part * per;
    double t = 0, sum = 0;
    int n = 0;
    for (int i0 = threadIdx.x; i0 < per; i0 += U * kBlock) {
        unsigned long long key[U];
#pragma unroll
        for (int u = 0; u < U; u++) key[u] = i0 + u * kBlock < per ? g_key[base + i0 + u * kBlock] : kNoKey;
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (key[u] == kNoKey) continue;
            n++;
            if (want_sums) {
#if MP_HIST_CM64
                const double c = (double)(uint32_t)(reinterpret_cast<const unsigned long long *>(g_cnt)[base + i0 + u * kBlock] >> 32);
#else
                const double c = (double)g_cnt[base + i0 + u * kBlock];
#endif
                t += c;
                sum += c * log2(c);
            }
        }
    }
    for (int sft = 32; sft >= 1; sft >>= 1) { t += __shfl_xor(t, sft); sum += __shfl_xor(sum, sft); n += __shfl_xor(n, sft); }
    if ((threadIdx.x & 63) == 0) { s_t[threadIdx.x >> 6] = t; s_s[threadIdx.x >> 6] = sum; s_n[threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tt = 0, ss = 0;
        int nn = 0;
        for (int i = 0; i < kBlock / 64; i++) { tt += s_t[i]; ss += s_s[i]; nn += s_n[i]; }
        out[blockIdx.x] = TableSums{tt, ss, nn, 0};
    }
}

// occupied slots -> entries of the window's segment (order inside a window is unspecified).  One workgroup per window
// walks the window's table with a running count in LDS: no global atomics (a returning atomic per wave on ~1000 hot
// addresses made the first version 0.7 ms at 131072 x 1000 for 128 MB of reads).
__global__ __launch_bounds__(kBlock) void compact_kernel(const CompactArgs A) {
    constexpr int U = 8, NWV = kBlock / 64;                   // a thread takes U slots per pass: U independent loads in flight, one barrier pair per pass
    __shared__ int s_part[U * NWV];
    __shared__ int s_base;
    // [r6] one workgroup per PIECE of a window's table (the pieces table_sums_kernel counted: the occupied slots in front of a piece are known),
    // not per window: at 10^6 rows the ~430 windows the gate leaves were 430 workgroups walking 1 MB each, 32 passes of two barriers — 0.43 ms
    // for 450 MB at 5 % of the vector ALUs' time
    const int w = blockIdx.x / A.parts, part = blockIdx.x % A.parts, per = A.g_slots / A.parts;
    if (A.win_base[w] < 0) return;                            // a window the entropy gate rejected on the device: no entries leave it
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = A.part_base[blockIdx.x];
    __syncthreads();
    const uint32_t kmask = (1u << A.k) - 1u;
    const int i_end = (part + 1) * per;
    for (int i0 = part * per; i0 < i_end; i0 += U * kBlock) {     // per is a power of two >= U * kBlock, or the whole table (parts = 1)
        unsigned long long key[U], m[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * kBlock + threadIdx.x;
            key[u] = i < i_end ? A.g_key[(size_t)w * A.g_slots + i] : kNoKey;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            m[u] = __ballot(key[u] != kNoKey);
            if (lane == 0) s_part[u * NWV + wave] = (int)__popcll(m[u]);
        }
        __syncthreads();
        int before = s_base, run = 0, mine[U];
#pragma unroll
        for (int q = 0; q < U * NWV; q++) {                  // slot order: pass-local index u * kBlock + wave * 64 + lane
            if (q % NWV == 0) mine[q / NWV] = 0;
            if (q % NWV == wave) mine[q / NWV] = run;
            run += s_part[q];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (key[u] == kNoKey) continue;
            const size_t s = (size_t)w * A.g_slots + i0 + u * kBlock + threadIdx.x;
            const int idx = before + mine[u] + (int)__popcll(m[u] & ((1ull << lane) - 1ull));
            A.g_idx[s] = idx;
            const long long e = A.win_base[w] + idx;
            if (e < A.cap) {
                A.b0[e] = (uint32_t)key[u] & kmask;
                A.b1[e] = (uint32_t)(key[u] >> A.k) & kmask;
                A.g[e] = A.g_gap ? ((key[u] & kGapFlag) ? A.g_gap[s] : 0u) : (uint32_t)(key[u] >> (2 * A.k)) & kmask;
#if MP_HIST_CM64
                const unsigned long long cm = reinterpret_cast<const unsigned long long *>(A.g_cnt)[s];
                A.count[e] = (int32_t)(uint32_t)(cm >> 32);
                A.first[e] = (int32_t)~(uint32_t)cm;
#else
                A.count[e] = (int32_t)A.g_cnt[s];
                A.first[e] = (int32_t)A.g_min[s];
#endif
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base = before + run;
        __syncthreads();
    }
}

// per-row labels (index of the row's entry inside its window, -1 = not in the histogram): JSON side files only
__global__ __launch_bounds__(kBlock) void label_kernel(const MsaArgs M, int p0, int k, const unsigned long long *__restrict__ g_key,
                                                       const uint32_t *__restrict__ g_gap, const int32_t *__restrict__ g_idx, int g_slots,
                                                       int32_t *__restrict__ labels) {
    const int per_win = M.n_pad / kBlock;
    const int w = blockIdx.x / per_win;
    const int r = (blockIdx.x % per_win) * kBlock + threadIdx.x;
    if (r >= M.n_rows) return;
    const uint32_t kmask = (1u << k) - 1u;
    uint32_t b0, b1, g;
    FlyView(M, p0 + w, k, kmask).load(r, b0, b1, g);
    int32_t lab = -1;
    if (!(g & MP_WIN_SKIP)) {
        unsigned long long key;
        uint32_t gap;
        if (g_gap) make_key<true>(b0, b1, g & kmask, k, key, gap);
        else make_key<false>(b0, b1, g & kmask, k, key, gap);
        const uint32_t mask = (uint32_t)g_slots - 1u;
        uint32_t h = hash64(key) & mask;
        for (int probe = 0; probe < g_slots; probe++) {
            const unsigned long long o = g_key[(size_t)w * g_slots + h];
            if (o == key && (!g_gap || !(key & kGapFlag) || g_gap[(size_t)w * g_slots + h] == gap)) { lab = g_idx[(size_t)w * g_slots + h]; break; }
            if (o == kNoKey) break;
            h = (h + 1) & mask;
        }
    }
    labels[(size_t)w * M.n_pad + r] = lab;
}

// ---------------------------------------------------------------------------------------------- k >= 22
template <typename W>
struct UniqueOut {
    W *b0, *b1, *g;
    int32_t *count, *first;
    long long cap;
    unsigned long long *total;   // entries allocated so far (may exceed cap: caller checks)
    int64_t *win_base;           // [W]
    int32_t *win_count;          // [W]
    int32_t *labels;             // [W][Npad] or nullptr
    int32_t *overflow;           // [W] set to 1 when the table did not fit
};

__device__ inline uint32_t shfl_word(uint32_t x, int lane) { return __shfl(x, lane); }
__device__ inline uint64_t shfl_word(uint64_t x, int lane) {
    return (uint64_t)(uint32_t)__shfl((uint32_t)x, lane) | ((uint64_t)(uint32_t)__shfl((uint32_t)(x >> 32), lane) << 32);
}

// One block per window.  Rows stream through in lanes; equal keys inside a wave are folded with
// ballots first, then one lane per distinct key updates the table.  A slot stores the row of a representative; key
// comparison re-derives the representative's window words from the planes.
// TABLE_IN_LDS = false: same algorithm on a global-memory table (windows with more distinct k-mers
// than the LDS table holds).
template <bool TABLE_IN_LDS, typename W>
__global__ __launch_bounds__(kBlock) void unique_kernel(const MsaArgs M, int p0, int k, const int32_t *__restrict__ win_list, int slots,
                                                        int limit, uint32_t *__restrict__ gtable, UniqueOut<W> out) {
    constexpr W kSkip = WordTraits<W>::kSkip;
    __shared__ uint32_t s_rep[TABLE_IN_LDS ? kHashSlots : 1];
    __shared__ uint32_t s_cnt[TABLE_IN_LDS ? kHashSlots : 1];
    __shared__ uint32_t s_min[TABLE_IN_LDS ? kHashSlots : 1];
    __shared__ int s_used, s_over, s_nout;
    __shared__ unsigned long long s_base;
    const int n_rows = M.n_rows, n_pad = M.n_pad;
    const int w = win_list ? win_list[blockIdx.x] : blockIdx.x;
    uint32_t *rep, *cnt, *mn;
    if (TABLE_IN_LDS) { rep = s_rep; cnt = s_cnt; mn = s_min; }
    else { rep = gtable + (size_t)blockIdx.x * 3 * slots; cnt = rep + slots; mn = cnt + slots; }
    const uint32_t mask = slots - 1;
    for (int i = threadIdx.x; i < slots; i += kBlock) { rep[i] = kEmpty; cnt[i] = 0; mn[i] = kEmpty; }
    if (threadIdx.x == 0) { s_used = 0; s_over = 0; s_nout = 0; }
    __syncthreads();
    const size_t np = (size_t)n_pad;
    const FlyViewT<W> V(M, p0 + w, k, kmask_of<W>(k));
    const int lane = threadIdx.x & 63;
    for (int base = 0; base < n_pad; base += kBlock) {
        int r = base + threadIdx.x;
        W b0 = 0, b1 = 0, g = kSkip;
        if (r < n_rows) V.load(r, b0, b1, g);
        bool todo = !(g & kSkip);
        unsigned long long pending = __ballot(todo);
        while (pending) {
            int lead = __ffsll((long long)pending) - 1;
            const W k0 = shfl_word(b0, lead), k1 = shfl_word(b1, lead), k2 = shfl_word(g, lead);
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
                        W o0, o1, o2;
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
            W o0, o1, o2;
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
        W b0, b1, g;
        V.load(r, b0, b1, g);
        int32_t lab = -1;
        if (!(g & kSkip)) {
            uint32_t h = hash3(b0, b1, g) & mask;
            for (int probe = 0; probe < slots; probe++) {
                uint32_t rr = rep[h];
                if (rr == kEmpty) break;
                W o0, o1, o2;
                V.load((int)rr, o0, o1, o2);
                if (o0 == b0 && o1 == b1 && o2 == g) { lab = (int32_t)cnt[h]; break; }
                h = (h + 1) & mask;
            }
        }
        out.labels[(size_t)w * np + r] = lab;
    }
}

int alloc_entries(mp_ctx *c, int64_t cap) {
    int rc;
    c->u_cap = cap;
    if ((rc = dev_alloc(c, &c->u_b0, (size_t)cap * wsz(c)))) return rc;
    if ((rc = dev_alloc(c, &c->u_b1, (size_t)cap * wsz(c)))) return rc;
    if ((rc = dev_alloc(c, &c->u_g, (size_t)cap * wsz(c)))) return rc;
    if ((rc = dev_alloc(c, &c->u_count, (size_t)cap))) return rc;
    if ((rc = dev_alloc(c, &c->u_first, (size_t)cap))) return rc;
    return MP_OK;
}

// k <= 31: LDS-combined global hash tables straight from the planes (k <= 21: the key is the 3k bits; k = 22..31: 2k bits + gap word)
int unique_packed(mp_ctx *c, int64_t cap, int32_t want_labels, int64_t *n_entries) {
    const size_t W = (size_t)c->n_win, np = (size_t)c->n_pad;
    const bool wide_key = !c->p64;
    int rc;
    Lap lap(c->stream);
    if ((rc = dev_alloc(c, &c->u_over, W))) return rc;
    if ((rc = dev_alloc(c, &c->u_wcount, W))) return rc;
    if ((rc = dev_alloc(c, &c->u_wbase, W))) return rc;
    int32_t *d_cursor = nullptr;
    // a table holds every distinct k-mer of a window; first try: twice the rows, capped at max(16384, rows / 16) slots (the deepest
    // windows of the 10^6-row bench alignment hold ~14000 distinct k-mers; [r6] rounds 2-5 capped at rows / 8: every pass over the
    // tables — fill, sums, compaction — is proportional to the slots, 2.6 GB of them at 10^6 rows)
    int slots = kBlock, cap_slots = 16384;
    while (cap_slots < c->n_rows / 16) cap_slots <<= 1;
    while (slots < 2 * c->n_rows + 64 && slots < cap_slots) slots <<= 1;
    if (const char *e = getenv("MP_HIST_SLOTS")) { int s = atoi(e); if (s >= kBlock && (s & (s - 1)) == 0) slots = s; }
    const bool gate = c->gate_threshold > 0 && !want_labels && !getenv("MP_NO_DEVICE_GATE");
    std::vector<int32_t> used(W), over(W);
    std::vector<double> sums(2 * W, 0.0);
    // E per window for the gate (below): an exception row with more than v gaps is one gap_sequence entry, any other counts once per
    // expansion (V20:689-707).  Host work that needs nothing from the device: it runs while the histogram kernel does.
    std::vector<double> extra;
    int extra_rc = MP_OK;
    auto count_extra = [&]() {
        extra.assign(W, 0.0);
        if ((extra_rc = ex_fetch(c))) return;                   // (the records may still be on their way: windows.hip)
        static const int kSetSize[16] = {1, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4};
        const size_t n_ex = c->ex_host.size();
        const int T = (int)std::max<size_t>(1, std::min<size_t>(8, n_ex / 8192));
        std::vector<std::vector<double>> part((size_t)T);
        const int k = c->k, v = c->v;
        run_on_threads(T, [&](int t) {
            std::vector<double> &mine = part[(size_t)t];
            mine.assign(W, 0.0);
            for (size_t i = n_ex * (size_t)t / (size_t)T, i1 = n_ex * (size_t)(t + 1) / (size_t)T; i < i1; i++) {
                const ExRec &x = c->ex_host[i];
                int gaps = 0;
                double n_exp = 1;
                for (int j = 0; j < k; j++) {
                    const uint32_t code = (uint32_t)((x.q[j >> 4] >> (4 * (j & 15))) & 15u);
                    gaps += code == 0;
                    n_exp *= kSetSize[code];
                }
                if (x.win >= 0 && (size_t)x.win < W) mine[(size_t)x.win] += gaps > v ? 1.0 : n_exp;
            }
        });
        for (int t = 0; t < T; t++)
            for (size_t w = 0; w < W; w++) extra[w] += part[(size_t)t][w];
    };
    int parts_used = 1;
    std::vector<int32_t> piece_used;                          // [W][parts_used] occupied slots per piece of the final tables
    for (int attempt = 0; attempt < 8; attempt++) {
        const size_t n = W * (size_t)slots;
        c->g_slots = slots;
        if ((rc = dev_alloc(c, &c->g_key, n))) return rc;
        if ((rc = dev_alloc(c, &c->g_cnt, MP_HIST_CM64 ? 2 * n : n))) return rc;       // (MP_HIST_CM64: one u64 per slot, no first-row array)
        if ((rc = dev_alloc(c, &c->g_min, MP_HIST_CM64 ? 1 : n))) return rc;
        if ((rc = dev_alloc(c, &c->g_idx, n))) return rc;
        if (wide_key && (rc = dev_alloc(c, &c->g_gap, n))) return rc;
        const FillSeg init[6] = {{c->g_key, sizeof(unsigned long long) * n, 0xFFFFFFFFu}, {c->g_cnt, sizeof(uint32_t) * (MP_HIST_CM64 ? 2 * n : n), 0u}, {c->g_min, sizeof(uint32_t) * (MP_HIST_CM64 ? 1 : n), 0xFFFFFFFFu},
                                 {c->u_wcount, sizeof(int32_t) * W, 0u}, {c->u_over, sizeof(int32_t) * W, 0u}, {c->g_gap, sizeof(uint32_t) * n, 0xFFFFFFFFu}};
        lap("unique: table alloc");
        if ((rc = fill_segments(c, init, wide_key ? 6 : 5))) return rc;
        lap("unique: table fill");
        HistArgs A{msa_args(c), c->p0, c->k, c->n_win, 0, 0, 0, c->g_key, c->g_cnt, c->g_min, c->g_gap, slots, c->u_over,
                   c->n_patch ? c->patch_off : (const int32_t *)nullptr, c->patch_rows, c->patch_words, nullptr};
        // enough workgroups to fill 256 CUs several times over, slices of at least 4096 rows
        int n_slices = (int)std::max<size_t>(1, std::min<size_t>((np + 4095) / 4096, (8192 + W - 1) / W));
        // ... and slices of at most 24576 rows: an XCD's resident workgroups walk a band of ~120 consecutive windows of ONE slice, i.e. the
        // plane words of four or five 32-column chunks of that slice — 16 bytes per row and chunk, which have to stay in the XCD's 4 MB of
        // L2 to be read from HBM once instead of once per window (10^6 rows: 9 slices of 116 k rows 9.5 ms, 48 slices 5.8 ms; 131072
        // rows: 9 slices of 14.6 k rows 0.71 ms, more slices only add table flushes — tools/r05_hist.sh)
        n_slices = (int)std::max<size_t>((size_t)n_slices, (np + 24575) / 24576);
        if (const char *e = getenv("MP_HIST_SLICES")) n_slices = std::max(1, atoi(e));
        A.rows_per_block = (int)(((np + n_slices - 1) / n_slices + kBlock - 1) / kBlock * kBlock);
        A.n_slices = (int)((np + A.rows_per_block - 1) / A.rows_per_block);
        A.win_per_xcd = (int)((W + 7) / 8);
        const unsigned blocks = 8u * (unsigned)A.win_per_xcd * (unsigned)A.n_slices;
        {
            const char *prof_path = getenv("MP_HIST_PROF");       // debugging: phase stamps of every workgroup written to that file
            if (prof_path) {
                HIPCK(c, hipMalloc((void **)&A.prof, (size_t)blocks * 64));
                HIPCK(c, hipMemsetAsync(A.prof, 0, (size_t)blocks * 64, c->stream));
            }
            int folds = 2;                                        // MP_HIST_FOLDS: A/B of the wave-level folding (0 = consensus only)
            if (const char *e = getenv("MP_HIST_FOLDS")) folds = atoi(e);
            if (!(getenv("MP_HIST_V1") && getenv("MP_HIST_V1")[0] == '1')) {                          // [r6] reference count + dense single-difference counters + dense inserts
                if (wide_key) hipLaunchKernelGGL((hist2_kernel<1024, true>), dim3(blocks), dim3(kBlock), 0, c->stream, A);
                else hipLaunchKernelGGL((hist2_kernel<1024, false>), dim3(blocks), dim3(kBlock), 0, c->stream, A);
            } else if (wide_key) {
                if (folds <= 1) hipLaunchKernelGGL((hist_kernel<2048, true, 1>), dim3(blocks), dim3(kBlock), 0, c->stream, A);
                else hipLaunchKernelGGL((hist_kernel<2048, true, 2>), dim3(blocks), dim3(kBlock), 0, c->stream, A);
            } else {
                if (folds <= 1) hipLaunchKernelGGL((hist_kernel<2048, false, 1>), dim3(blocks), dim3(kBlock), 0, c->stream, A);
                else if (folds == 2) hipLaunchKernelGGL((hist_kernel<2048, false, 2>), dim3(blocks), dim3(kBlock), 0, c->stream, A);
                else hipLaunchKernelGGL((hist_kernel<2048, false, 3>), dim3(blocks), dim3(kBlock), 0, c->stream, A);
            }
            if (prof_path) {
                std::vector<unsigned long long> h((size_t)blocks * 8);
                HIPCK(c, hipStreamSynchronize(c->stream));
                HIPCK(c, hipMemcpy(h.data(), A.prof, h.size() * 8, hipMemcpyDeviceToHost));
                (void)hipFree(A.prof);
                A.prof = nullptr;
                if (FILE *f = fopen(prof_path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
            }
        }
        HIPCK(c, hipGetLastError());
        // occupied slots and the gate's sums in one pass over the tables, one read-back
        int parts = 1;
        while (parts < 16 && slots / (parts * 2) >= 8 * kBlock) parts <<= 1;
        TableSums *d_sums = nullptr;
        if ((rc = dev_alloc(c, &d_sums, W * (size_t)parts))) return rc;
        hipLaunchKernelGGL(table_sums_kernel, dim3((unsigned)(W * (size_t)parts)), dim3(kBlock), 0, c->stream, (const unsigned long long *)c->g_key,
                           (const uint32_t *)c->g_cnt, slots, parts, gate ? 1 : 0, d_sums);
        std::vector<TableSums> hs(W * (size_t)parts);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(hs.data(), d_sums, sizeof(TableSums) * hs.size(), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(over.data(), c->u_over, sizeof(int32_t) * W, hipMemcpyDeviceToHost, c->stream);
        if (gate && extra.empty()) count_extra();              // (host work beside the kernels)
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        dev_free(c, &d_sums, W * (size_t)parts);
        if (e != hipSuccess) return fail(c, MP_ERR_DEVICE, "histogram tables: %s", hipGetErrorString(e));
        if (extra_rc) return extra_rc;
        lap("unique: histogram + sums + d2h");
        bool any_over = false;
        for (size_t w = 0; w < W; w++) {
            int32_t u = 0;
            double t = 0, sum = 0;
            for (int q = 0; q < parts; q++) { const TableSums &x = hs[w * (size_t)parts + (size_t)q]; u += x.used; t += x.t; sum += x.s; }
            used[w] = u; sums[2 * w] = t; sums[2 * w + 1] = sum;
            any_over |= over[w] != 0 || u > slots - slots / 8;
        }
        parts_used = parts;
        piece_used.resize(hs.size());
        for (size_t i = 0; i < hs.size(); i++) piece_used[i] = hs[i].used;
        if (!any_over) break;
        // a window has (nearly) as many distinct k-mers as slots: start over with tables 8x the size (rare: random input)
        dev_free(c, &c->g_key, n); dev_free(c, &c->g_cnt, MP_HIST_CM64 ? 2 * n : n); dev_free(c, &c->g_min, MP_HIST_CM64 ? 1 : n); dev_free(c, &c->g_idx, n); dev_free(c, &c->g_gap, n);
        if (attempt == 7 || (size_t)slots * 8 > ((size_t)1 << 28)) return fail(c, MP_ERR_NOMEM, "histogram tables do not converge");
        slots *= 8;
    }
    // The entropy gate on the device (armed by mp_set_entropy_gate, no labels wanted): V20:723 rejects a window whose "Entropy of total",
    // rounded to two decimals, exceeds the threshold — and more than half of the windows of a deep alignment end there, after the host has
    // decoded, merged and ordered all their entries (they are the windows with the MOST entries).  The tables hold what the host would
    // sum, except the rows with an IUPAC code: X such rows add E >= X unit masses (one per expansion; V20:689-707).  The host's tBit
    // (V20:602-614) divides every count by N = cover_number + gap_sequence_number — ROWS, N = T + X — while the masses sum to
    // M = T + E >= N: with q the normalised distribution of the masses and r = M / N, tBit = -sum (c / N) log2(c / N) = r (H(q) - log2 r).
    // (a) H(q) is within theta log2(T + E) + h2(theta), theta = E / (T + E), of the table's entropy H = log2 T - S / T (q = (1 - theta) p
    // + theta e: total-variation distance <= theta; Fannes-Audenaert), 1e-6 covers the floating-point order.  (b) [r6, advisor] r is in
    // [1, (T + E) / T]; f(r) = r (H - log2 r) is concave in r, so over that interval it is at least min(f(1), f(r_max)), and it rises
    // with H.  Hence tBit >= min(Hlow, r_max (Hlow - log2 r_max)) with Hlow = H - bound; rounding moves tBit by at most 0.005.  A window
    // whose lower bound clears threshold + 0.005 is rejected here, for certain (tests/test_gate_bound.py compares with the exact host
    // value on IUPAC-heavy windows); every other window goes to the host as before, which decides exactly.  A rejected window's entries
    // are not compacted, not read back, not planned.
    c->h_wskip.clear();
    if (gate) {
        c->h_wskip.assign(W, 0);
        size_t n_skip = 0;
        for (size_t w = 0; w < W; w++) {
            const double T = sums[2 * w], S = sums[2 * w + 1], E = extra[w];
            if (!(T > 0)) continue;
            const double H = std::log2(T) - S / T, theta = E / (T + E);
            const double h2 = theta > 0 && theta < 1 ? -theta * std::log2(theta) - (1 - theta) * std::log2(1 - theta) : 0.0;
            const double bound = theta * std::log2(T + E) + h2 + 1e-6;
            const double Hq = H - bound, r_max = (T + E) / T;
            const double low = std::min(Hq, r_max * (Hq - std::log2(r_max)));
            if (low > c->gate_threshold + 0.005) { c->h_wskip[w] = 1; used[w] = 0; n_skip++; }
        }
        if (getenv("MP_TRACE")) fprintf(stderr, "[mprime] unique: entropy gate on the device rejected %zu of %zu windows\n", n_skip, W);
        lap("unique: entropy gate (host part)");
    }
    c->h_wbase.resize(W); c->h_wcount.resize(W);
    int64_t total = 0;
    std::vector<int64_t> dev_base(W);
    for (size_t w = 0; w < W; w++) {
        c->h_wbase[w] = total; c->h_wcount[w] = used[w]; total += used[w];
        dev_base[w] = !c->h_wskip.empty() && c->h_wskip[w] ? -1 : c->h_wbase[w];
    }
    if (n_entries) *n_entries = total;
    if (total > cap) { c->u_n = 0; return fail(c, MP_ERR_CAPACITY, "unique table needs %lld entries", (long long)total); }
    if ((rc = alloc_entries(c, std::max<int64_t>(total, 1)))) return rc;
    if ((rc = dev_alloc(c, &d_cursor, W))) return rc;
    HIPCK(c, hipMemcpyAsync(c->u_wbase, dev_base.data(), sizeof(int64_t) * W, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemsetAsync(d_cursor, 0, sizeof(int32_t) * W, c->stream));
    std::vector<int32_t> piece_base(piece_used.size());
    for (size_t w = 0; w < W; w++) {
        int32_t run = 0;
        for (int q = 0; q < parts_used; q++) { piece_base[w * (size_t)parts_used + (size_t)q] = run; run += piece_used[w * (size_t)parts_used + (size_t)q]; }
    }
    int32_t *d_pbase = nullptr;
    if ((rc = dev_alloc(c, &d_pbase, piece_base.size()))) return rc;
    HIPCK(c, hipMemcpyAsync(d_pbase, piece_base.data(), sizeof(int32_t) * piece_base.size(), hipMemcpyHostToDevice, c->stream));
    CompactArgs CA{c->g_key, c->g_cnt, c->g_min, c->g_gap, c->g_idx, c->g_slots, c->k, c->u_wbase, d_cursor, parts_used, d_pbase,
                   c->u_b0, c->u_b1, c->u_g, c->u_count, c->u_first, (long long)c->u_cap};
    hipLaunchKernelGGL(compact_kernel, dim3((unsigned)(W * (size_t)parts_used)), dim3(kBlock), 0, c->stream, CA);
    HIPCK(c, hipGetLastError());
    if (want_labels) {
        if ((rc = dev_alloc(c, &c->labels, W * np))) return rc;
        hipLaunchKernelGGL(label_kernel, dim3((unsigned)((np / kBlock) * W)), dim3(kBlock), 0, c->stream, msa_args(c), c->p0, c->k,
                           (const unsigned long long *)c->g_key, (const uint32_t *)c->g_gap, (const int32_t *)c->g_idx, c->g_slots, c->labels);
        HIPCK(c, hipGetLastError());
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    lap("unique: entries alloc+compact");
    dev_free(c, &d_cursor, W);
    dev_free(c, &d_pbase, piece_base.size());
    c->u_n = total;
    return MP_OK;
}

// k >= 22: one workgroup per window, representative-row table (W = the window word type: 64-bit for k > 31)
template <typename W>
int unique_rep_rows(mp_ctx *c, int64_t cap, int32_t want_labels, int64_t *n_entries) {
    const size_t n_w = (size_t)c->n_win, np = (size_t)c->n_pad;
    int rc;
    if ((rc = alloc_entries(c, cap))) return rc;
    if ((rc = dev_alloc(c, &c->u_over, n_w))) return rc;
    if ((rc = dev_alloc(c, &c->u_wcount, n_w))) return rc;
    if ((rc = dev_alloc(c, &c->u_wbase, n_w))) return rc;
    if ((rc = dev_alloc(c, &c->u_total, 1))) return rc;
    if (want_labels && (rc = dev_alloc(c, &c->labels, n_w * np))) return rc;
    HIPCK(c, hipMemsetAsync(c->u_total, 0, sizeof(unsigned long long), c->stream));
    UniqueOut<W> uo{reinterpret_cast<W *>(c->u_b0), reinterpret_cast<W *>(c->u_b1), reinterpret_cast<W *>(c->u_g), c->u_count, c->u_first, (long long)cap,
                    c->u_total, c->u_wbase, c->u_wcount, c->labels, c->u_over};
    const MsaArgs M = msa_args(c);
    hipLaunchKernelGGL((unique_kernel<true, W>), dim3((unsigned)n_w), dim3(kBlock), 0, c->stream, M, c->p0, c->k, (const int32_t *)nullptr,
                       kHashSlots, kHashLimit, (uint32_t *)nullptr, uo);
    HIPCK(c, hipGetLastError());
    std::vector<int32_t> over(n_w);
    HIPCK(c, hipMemcpyAsync(over.data(), c->u_over, sizeof(int32_t) * n_w, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    std::vector<int32_t> big;
    for (size_t w = 0; w < n_w; w++) if (over[w]) big.push_back((int32_t)w);
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
            hipLaunchKernelGGL((unique_kernel<false, W>), dim3((unsigned)nb), dim3(kBlock), 0, c->stream, M, c->p0, c->k,
                               (const int32_t *)d_list, slots, slots - 32, gtable, uo);
            HIPCK(c, hipGetLastError());
            HIPCK(c, hipStreamSynchronize(c->stream));
        }
        dev_free(c, &gtable, batch * 3 * (size_t)slots);
        dev_free(c, &d_list, batch);
    }
    unsigned long long total = 0;
    c->h_wbase.resize(n_w); c->h_wcount.resize(n_w);
    HIPCK(c, hipMemcpy(&total, c->u_total, sizeof(total), hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(c->h_wbase.data(), c->u_wbase, sizeof(int64_t) * n_w, hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(c->h_wcount.data(), c->u_wcount, sizeof(int32_t) * n_w, hipMemcpyDeviceToHost));
    if (n_entries) *n_entries = (int64_t)total;
    if ((long long)total > cap) { c->u_n = 0; return fail(c, MP_ERR_CAPACITY, "unique table needs %llu entries", total); }
    c->u_n = (long long)total;
    return MP_OK;
}

}  // namespace

namespace {
// [r6] A band of histogram entries into the registered staging area of the host BY A KERNEL (stores over the link): its five pieces —
// three word arrays, counts, first rows — in one launch.  The copy engine charges ~50-70 us per hipMemcpyAsync whatever its size; 14 bands
// x 5 arrays were 70 copies of ~0.7 MB, 5 ms for 49 MB that cross the link in ~1.5.
struct D2hSeg { const uint32_t *src; uint32_t *dst; unsigned long long n; };       // n 32-bit words
struct D2hArgs { D2hSeg seg[5]; };
__global__ __launch_bounds__(kBlock) void d2h_band_kernel(const D2hArgs A) {
    const unsigned long long stride = (unsigned long long)gridDim.x * kBlock;
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const uint32_t *__restrict__ src = A.seg[q].src;
        uint32_t *__restrict__ dst = A.seg[q].dst;
        for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < A.seg[q].n; i += stride) dst[i] = src[i];
    }
}

__global__ __launch_bounds__(kBlock) void gather_labels_kernel(const int32_t *__restrict__ labels, const int32_t *__restrict__ windows, int n,
                                                               int n_rows, int n_pad, int32_t *__restrict__ out) {
    const long long total = (long long)n * n_rows;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int w = (int)(i / n_rows), r = (int)(i % n_rows);
        out[i] = labels[(size_t)windows[w] * n_pad + r];
    }
}
}  // namespace

extern "C" {

int mp_window_unique(mp_ctx *c, int64_t cap, int32_t want_labels, int64_t *n_entries) {
    if (!c) return MP_ERR_ARG;
    if (!c->excl) return fail(c, MP_ERR_ARG, "no windows built");
    if (cap <= 0) return fail(c, MP_ERR_ARG, "cap_entries must be positive");
    HIPCK(c, hipSetDevice(c->dev));
    free_unique(c);
    c->h_wskip.clear();
    if (!c->wide && !getenv("MP_HIST_REP_ROWS")) return unique_packed(c, cap, want_labels, n_entries);     // k <= 31
    return c->wide ? unique_rep_rows<uint64_t>(c, cap, want_labels, n_entries) : unique_rep_rows<uint32_t>(c, cap, want_labels, n_entries);
}

int mp_set_entropy_gate(mp_ctx *c, double threshold) {
    if (!c) return MP_ERR_ARG;
    if (!(threshold >= 0)) return fail(c, MP_ERR_ARG, "mp_set_entropy_gate: the threshold must be >= 0 (0 switches the gate off)");
    c->gate_threshold = threshold;
    return MP_OK;
}

int mp_entropy_gate_result(mp_ctx *c, int32_t *n_rejected, uint8_t *rejected) {
    if (!c) return MP_ERR_ARG;
    int32_t n = 0;
    for (size_t w = 0; w < c->h_wskip.size(); w++) { n += c->h_wskip[w]; if (rejected) rejected[w] = c->h_wskip[w]; }
    if (rejected && c->h_wskip.empty()) memset(rejected, 0, (size_t)c->n_win);
    if (n_rejected) *n_rejected = n;
    return MP_OK;
}

int mp_get_unique(mp_ctx *c, int64_t *win_off, void *words_out, int32_t *count, int32_t *first_row) {
    if (!c) return MP_ERR_ARG;
    if (c->h_wbase.empty()) return fail(c, MP_ERR_ARG, "mp_window_unique has not run");
    HIPCK(c, hipSetDevice(c->dev));
    size_t n = (size_t)c->u_n, W = (size_t)c->n_win;
    // segments are laid out in window order when every window's base is the running sum (the packed path)
    bool in_order = true;
    {
        int64_t o = 0;
        for (size_t w = 0; w < W; w++) { if (c->h_wbase[w] != o) { in_order = false; break; } o += c->h_wcount[w]; }
    }
    const size_t wb = 4 * wsz(c);                       // bytes per window word
    uint8_t *words = static_cast<uint8_t *>(words_out);
    if (in_order) {
        Lap lap(c->stream);
        if (n) {
            prefault_host(words, 3 * wb * n); prefault_host(count, 4 * n); prefault_host(first_row, 4 * n);
            lap("get_unique: prefault");
            HIPCK(c, hipMemcpyAsync(words, c->u_b0, wb * n, hipMemcpyDeviceToHost, c->stream));
            HIPCK(c, hipMemcpyAsync(words + wb * n, c->u_b1, wb * n, hipMemcpyDeviceToHost, c->stream));
            HIPCK(c, hipMemcpyAsync(words + 2 * wb * n, c->u_g, wb * n, hipMemcpyDeviceToHost, c->stream));
            HIPCK(c, hipMemcpyAsync(count, c->u_count, 4 * n, hipMemcpyDeviceToHost, c->stream));
            HIPCK(c, hipMemcpyAsync(first_row, c->u_first, 4 * n, hipMemcpyDeviceToHost, c->stream));
            HIPCK(c, hipStreamSynchronize(c->stream));
        }
        if (lap.on) fprintf(stderr, "[mprime] get_unique: %zu entries, %.1f MB\n", n, (double)(3 * wb + 8) * (double)n / 1e6);
        lap("get_unique: d2h");
        int64_t o = 0;
        for (size_t w = 0; w < W; w++) { win_off[w] = o; o += c->h_wcount[w]; }
        win_off[W] = o;
        return MP_OK;
    }
    std::vector<uint8_t> b0(wb * (n + 1)), b1(wb * (n + 1)), g(wb * (n + 1));
    std::vector<int32_t> cn(n + 1), fr(n + 1);
    if (n) {
        HIPCK(c, hipMemcpy(b0.data(), c->u_b0, wb * n, hipMemcpyDeviceToHost));
        HIPCK(c, hipMemcpy(b1.data(), c->u_b1, wb * n, hipMemcpyDeviceToHost));
        HIPCK(c, hipMemcpy(g.data(), c->u_g, wb * n, hipMemcpyDeviceToHost));
        HIPCK(c, hipMemcpy(cn.data(), c->u_count, 4 * n, hipMemcpyDeviceToHost));
        HIPCK(c, hipMemcpy(fr.data(), c->u_first, 4 * n, hipMemcpyDeviceToHost));
    }
    // the kernel reserved each window's segment with one atomic; lay the segments out in window order
    int64_t o = 0;
    for (size_t w = 0; w < W; w++) {
        win_off[w] = o;
        size_t src = (size_t)c->h_wbase[w], m = (size_t)c->h_wcount[w];
        memcpy(words + wb * (size_t)o, b0.data() + wb * src, wb * m);
        memcpy(words + wb * (n + (size_t)o), b1.data() + wb * src, wb * m);
        memcpy(words + wb * (2 * n + (size_t)o), g.data() + wb * src, wb * m);
        for (size_t i = 0; i < m; i++) {
            count[(size_t)o + i] = cn[src + i];
            first_row[(size_t)o + i] = fr[src + i];
        }
        o += (int64_t)m;
    }
    win_off[W] = o;
    return MP_OK;
}

// mprime_host.h: the planning stage fed band by band while the entries come off the device
int mp_plan_create_streamed(mp_ctx *c, const mp_plan_params *params, int64_t row_base, int64_t n_exc, const int32_t *x_window,
                            const int64_t *x_row, const uint8_t *x_codes, const int64_t *freq, const int64_t *nn, int64_t *n_entries,
                            mp_plan **out) {
    if (!c || !params || !out) return MP_ERR_ARG;
    *out = nullptr;
    if (c->h_wbase.empty()) return fail(c, MP_ERR_ARG, "mp_window_unique has not run");
    if (params->n_windows != c->n_win || params->k != c->k) return fail(c, MP_ERR_ARG, "mp_plan_create_streamed: the parameters are not this context's windows");
    const bool trace = getenv("MP_TRACE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    HIPCK(c, hipSetDevice(c->dev));
    const size_t n = (size_t)c->u_n, W = (size_t)c->n_win, wb = 4 * wsz(c);
    if (n_entries) *n_entries = (int64_t)n;
    std::vector<int64_t> e_off(W + 1);
    bool in_order = true;
    {
        int64_t o = 0;
        for (size_t w = 0; w < W; w++) { e_off[w] = o; if (c->h_wbase[w] != o) in_order = false; o += c->h_wcount[w]; }
        e_off[W] = o;
    }
    // the context's staging area: words [3][n], counts [n], first rows [n]; grown when a larger table comes along, its pages faulted in
    // on several threads the first time (common.hpp prefault_host), reused as it stands afterwards
    const size_t words_bytes = (3 * wb * (n + 1) + 63) / 64 * 64, need = words_bytes + 2 * 4 * (n + 1);
    if (c->h_stage_bytes < need) {
        if (c->h_stage_pinned) (void)hipHostUnregister(c->h_stage);
        c->h_stage_pinned = false;
        host_unmap(c->h_stage, c->h_stage_bytes);
        c->h_stage_bytes = 0;
        const size_t room = (need + need / 8 + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
        c->h_stage = static_cast<uint8_t *>(host_map(room));
        if (!c->h_stage) return fail(c, MP_ERR_NOMEM, "mp_plan_create_streamed: out of host memory (%zu bytes)", need);
        c->h_stage_bytes = room;
        prefault_host(c->h_stage, room);
        // registered for the context's life when it is large enough to matter: the bands below are then DMA copies that take no host thread
        // away from the planning (common.hpp on why this is the only registration the library makes)
        if (room >= ((size_t)8 << 20) && !getenv("MP_NO_PIN") && hipHostRegister(c->h_stage, room, hipHostRegisterDefault) == hipSuccess) c->h_stage_pinned = true;
        else (void)hipGetLastError();
    }
    struct Span { uint8_t *p; uint8_t *get() const { return p; } };
    struct SpanI { int32_t *p; int32_t *get() const { return p; } int32_t &operator[](size_t i) const { return p[i]; } };
    const Span words{c->h_stage};
    const SpanI count{reinterpret_cast<int32_t *>(c->h_stage + words_bytes)}, first{reinterpret_cast<int32_t *>(c->h_stage + words_bytes) + (n + 1)};
    if (!in_order && c->stats_pending_f) {             // (no band copies to hide the statistics behind on this route)
        int rc = mp_window_stats_end(c, const_cast<int64_t *>(freq), const_cast<int64_t *>(nn));
        if (rc) return rc;
    }
    if (!in_order) {
        // the k >= 32 histograms reserve their segments in completion order: one blocking read-back that lays them out by window
        int rc = mp_get_unique(c, e_off.data(), words.get(), count.get(), first.get());
        if (rc) return rc;
        return mp_plan_create_segments(params, e_off.data(), words.get(), count.get(), first.get(), row_base, n_exc, x_window, x_row, x_codes, freq, nn, out);
    }
    mp_ready_gate ready;
    std::atomic<int> copy_error{0};
    if (trace) fprintf(stderr, "[mprime] plan_streamed: staging area ready at %.3f ms\n", ms_since());
    std::thread copier([&]() {
        hipError_t e = hipSetDevice(c->dev);
        bool stats_owed = c->stats_pending_f != 0;     // [r6] mp_window_stats_begin is pending: its counters land in freq / nn before a planner may read them
        // bands of whole windows: two small ones so that the planners start early, then a sixth (a twelfth when the area is registered)
        // of the entries each — a band is five copies and every copy through the runtime's staging buffers has ~50 us of its own
        // (24 bands took 6 ms for 58 MB that cross in 1.1 ms as one piece)
        const size_t target = std::max<size_t>(n / (c->h_stage_pinned ? 12 : 6), 65536);
        // [r6] with a registered staging area every copy is a DMA the runtime only queues: ALL bands' copies are queued up front, an event
        // behind each band, and this thread then waits for the events in turn — the copy engine runs back to back (rounds 4-5 queued a band,
        // waited for it, queued the next: ~0.1 ms of idle engine between the 14 bands).  Unregistered (staged) copies keep that order: each
        // of them blocks in the runtime anyway.
        struct Band { size_t w1, a, m; hipEvent_t ev; };
        std::vector<Band> bands;
        for (size_t w0 = 0; w0 < W;) {
            size_t w1 = w0 + 1;
            const size_t want = bands.size() < 2 ? target / 8 : target;
            while (w1 < W && (size_t)(e_off[w1] - e_off[w0]) < want) w1++;
            bands.push_back(Band{w1, (size_t)e_off[w0], (size_t)(e_off[w1] - e_off[w0]), nullptr});
            w0 = w1;
        }
        // the staging area as the device sees it (registered memory is mapped): the band kernel stores straight into it
        uint8_t *dev_stage = nullptr;
        // (measured equal to the copy engine within the noise of a box — 10^6 rows, 49 MB: planning 5.1-6.9 ms against 4.9-5.6 — so it is opt-in:
        // MP_PLAN_D2H_KERNEL=1; what slows the read-back at that depth is the host side, 96 planning threads on the same memory)
        if (c->h_stage_pinned && getenv("MP_PLAN_D2H_KERNEL") && hipHostGetDevicePointer((void **)&dev_stage, c->h_stage, 0) != hipSuccess) { (void)hipGetLastError(); dev_stage = nullptr; }
        auto queue_band = [&](Band &B) {
            const size_t a = B.a, m = B.m;
            if (!m || e != hipSuccess) return;
            if (dev_stage) {
                const size_t ww = wb / 4;                                  // 32-bit units per window word
                auto at = [&](const void *host) { return reinterpret_cast<uint32_t *>(dev_stage + (static_cast<const uint8_t *>(host) - c->h_stage)); };
                D2hArgs A;
                A.seg[0] = D2hSeg{reinterpret_cast<const uint32_t *>(c->u_b0) + ww * a, at(words.get() + wb * a), (unsigned long long)(ww * m)};
                A.seg[1] = D2hSeg{reinterpret_cast<const uint32_t *>(c->u_b1) + ww * a, at(words.get() + wb * (n + a)), (unsigned long long)(ww * m)};
                A.seg[2] = D2hSeg{reinterpret_cast<const uint32_t *>(c->u_g) + ww * a, at(words.get() + wb * (2 * n + a)), (unsigned long long)(ww * m)};
                A.seg[3] = D2hSeg{reinterpret_cast<const uint32_t *>(c->u_count) + a, at(count.get() + a), (unsigned long long)m};
                A.seg[4] = D2hSeg{reinterpret_cast<const uint32_t *>(c->u_first) + a, at(first.get() + a), (unsigned long long)m};
                const unsigned blocks = (unsigned)std::min<size_t>((ww * m + kBlock - 1) / kBlock, 2048);
                hipLaunchKernelGGL(d2h_band_kernel, dim3(std::max(1u, blocks)), dim3(kBlock), 0, c->stream, A);
                e = hipGetLastError();
                return;
            }
            e = hipMemcpyAsync(words.get() + wb * a, reinterpret_cast<const uint8_t *>(c->u_b0) + wb * a, wb * m, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(words.get() + wb * (n + a), reinterpret_cast<const uint8_t *>(c->u_b1) + wb * a, wb * m, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(words.get() + wb * (2 * n + a), reinterpret_cast<const uint8_t *>(c->u_g) + wb * a, wb * m, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(count.get() + a, c->u_count + a, 4 * m, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(first.get() + a, c->u_first + a, 4 * m, hipMemcpyDeviceToHost, c->stream);
        };
        const bool ahead = c->h_stage_pinned && !getenv("MP_PLAN_NO_AHEAD");
        if (ahead)
            for (Band &B : bands) {
                queue_band(B);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&B.ev, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(B.ev, c->stream);
            }
        int band = 0;
        for (Band &B : bands) {
            if (ahead) { if (e == hipSuccess && B.ev) e = hipEventSynchronize(B.ev); }
            else { queue_band(B); if (e == hipSuccess) e = hipStreamSynchronize(c->stream); }
            const size_t a = B.a, m = B.m, w1 = B.w1;
            if (e != hipSuccess) {                       // the planners must not wait for ever: hand them zeroed entries, the result is thrown away
                copy_error.store((int)e);
                memset(words.get() + wb * a, 0, wb * m); memset(words.get() + wb * (n + a), 0, wb * m); memset(words.get() + wb * (2 * n + a), 0, wb * m);
                for (size_t i = a; i < a + m; i++) { count[i] = 1; first[i] = 0; }
            }
            if (stats_owed) {
                stats_owed = false;
                if (mp_window_stats_end(c, const_cast<int64_t *>(freq), const_cast<int64_t *>(nn)) != MP_OK) copy_error.store((int)hipErrorUnknown);
                if (trace) fprintf(stderr, "[mprime] plan_streamed: window statistics there at %.3f ms\n", ms_since());
            }
            ready.raise((int)w1);
            if (trace && (band < 3 || w1 == W)) fprintf(stderr, "[mprime] plan_streamed: band %d (windows < %zu) there at %.3f ms\n", band, w1, ms_since());
            band++;
        }
        if (e != hipSuccess) (void)hipStreamSynchronize(c->stream);      // (queued copies must not outlive the buffers)
        for (Band &B : bands) if (B.ev) (void)hipEventDestroy(B.ev);
        if (trace) fprintf(stderr, "[mprime] plan_streamed: %zu entries, %.1f MB in %zu bands\n", n, (double)(3 * wb + 8) * (double)n / 1e6, bands.size());
    });
    int rc = mp_plan_create_segments_ready(params, e_off.data(), words.get(), count.get(), first.get(), row_base, n_exc, x_window, x_row, x_codes, freq, nn,
                                           &ready, c->h_wskip.empty() ? nullptr : c->h_wskip.data(), out);
    if (trace) fprintf(stderr, "[mprime] plan_streamed: planning done at %.3f ms\n", ms_since());
    copier.join();
    if (c->stats_pending_f) (void)mp_window_stats_end(c, const_cast<int64_t *>(freq), const_cast<int64_t *>(nn));      // (no window at all: nobody asked)
    if (trace) fprintf(stderr, "[mprime] plan_streamed: returning at %.3f ms\n", ms_since());
    if (copy_error.load()) {
        if (*out) { mp_plan_destroy(*out); *out = nullptr; }
        return fail(c, MP_ERR_DEVICE, "mp_plan_create_streamed: %s", hipGetErrorString((hipError_t)copy_error.load()));
    }
    return rc;
}

int mp_get_labels(mp_ctx *c, int32_t w, int32_t *labels) {
    if (!c) return MP_ERR_ARG;
    if (!c->labels) return fail(c, MP_ERR_ARG, "labels were not requested");
    if (w < 0 || w >= c->n_win) return fail(c, MP_ERR_ARG, "bad window");
    HIPCK(c, hipSetDevice(c->dev));
    HIPCK(c, hipMemcpy(labels, c->labels + (size_t)w * c->n_pad, sizeof(int32_t) * (size_t)c->n_rows, hipMemcpyDeviceToHost));
    return MP_OK;
}
int mp_get_labels_many(mp_ctx *c, int32_t n, const int32_t *windows, int32_t *labels) {
    if (!c) return MP_ERR_ARG;
    if (!c->labels) return fail(c, MP_ERR_ARG, "labels were not requested");
    if (n < 0 || (n && (!windows || !labels))) return fail(c, MP_ERR_ARG, "mp_get_labels_many: bad arguments");
    for (int32_t i = 0; i < n; i++)
        if (windows[i] < 0 || windows[i] >= c->n_win) return fail(c, MP_ERR_ARG, "bad window");
    HIPCK(c, hipSetDevice(c->dev));
    if (n == 0) return MP_OK;
    // the rows of the asked windows are gathered into one buffer on the device: one copy back instead of one per window
    int32_t *d_win = nullptr, *d_out = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &d_win, (size_t)n))) return rc;
    if ((rc = dev_alloc(c, &d_out, (size_t)n * c->n_rows))) { dev_free(c, &d_win, (size_t)n); return rc; }
    hipError_t e = hipMemcpyAsync(d_win, windows, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const long long total = (long long)n * c->n_rows;
        hipLaunchKernelGGL(gather_labels_kernel, dim3((unsigned)std::min<long long>((total + kBlock - 1) / kBlock, 65535 * 16)), dim3(kBlock), 0,
                           c->stream, (const int32_t *)c->labels, (const int32_t *)d_win, n, c->n_rows, c->n_pad, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(labels, d_out, sizeof(int32_t) * (size_t)n * c->n_rows, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(c, &d_win, (size_t)n);
    dev_free(c, &d_out, (size_t)n * c->n_rows);
    if (e != hipSuccess) return fail(c, MP_ERR_DEVICE, "mp_get_labels_many: %s", hipGetErrorString(e));
    return MP_OK;
}

}  // extern "C"
