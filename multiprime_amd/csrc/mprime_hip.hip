// mprime_hip.hip — hand-written HIP kernels (gfx950 / MI355X, wave64) and the C ABI of
// include/mprime.h for the hot path of multiPrime's core step.
//
// Data layout in HBM:
//   planes  [n_chunks][4][Npad] u32   one-hot base-set bit planes (A,C,G,T membership) of 32
//                                     alignment columns per word, sequences along the fastest
//                                     axis, so a wave reads 64 consecutive sequences per load
//   cum     [n_chunks+1][Npad] u32    residues (non-gap symbols) left of each 32-column chunk
//   ung     [N][ustride] u32          gap-free residue codes, 8 nibbles per word, row-major
//                                     (only touched by the edge-gap repair path)
//   win     the k-mer of every (window, sequence) after repair, 3 bits per symbol:
//             k <= 21: [W][Npad] u64       b0 | b1 << k | g << 2k, bit 63 = not in the universe
//             k >= 22: [W][3][Npad] u32    b0,b1 (2-bit base) and g (gap flag) words
//   uniq    per-window histogram entries (words, count, first row), labels [W][Npad]
//
// Kernels: pack_kernel, row_scan_kernel, ungap_kernel (mp_load_msa), build_windows_kernel
// (V20:666-687), unique_kernel (V20:689-711, LDS hash table per window), eval_kernel
// (V20:1103-1130 + 229-233; the candidate x sequence evaluation the benchmark measures).
// No MFMA: this is bit-mask work bounded by HBM / integer ALU.  gfx950 only.

#include "../../include/mprime.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;
constexpr int kHashSlots = 4096;          // LDS hash table slots per window (unique_kernel)
constexpr int kHashLimit = 3584;          // load limit before the window is handed to the global-table path
constexpr int kEvalCC = 8;                // candidates evaluated per block pass

struct ExRec { int32_t win, row; uint64_t lo, hi; };          // exception k-mer, 16+12 nibbles
struct EvalItem { int32_t win, cand0; };                        // one block's work: window + first padded candidate

// ----------------------------------------------------------------------------------------------
// (1) alignment -> planes
// ----------------------------------------------------------------------------------------------
__constant__ uint8_t c_code_lut[256];

// V20:453: upper-case, keep ACGTRYMKSWHBVD (as 4-bit base sets), everything else (N included) -> '-' = 0
static void host_code_lut(uint8_t *lut) {
    memset(lut, 0, 256);
    const char *sym = "ACGTRYMKSWHBVD";
    const uint8_t code[] = {1, 2, 4, 8, 5, 10, 3, 12, 6, 9, 11, 14, 7, 13};
    for (int i = 0; sym[i]; i++) {
        lut[(uint8_t)sym[i]] = code[i];
        lut[(uint8_t)(sym[i] + 32)] = code[i];
    }
}

// thread = (row, 32-column chunk); lanes run along rows so that the plane stores coalesce
__global__ __launch_bounds__(kBlock) void pack_kernel(const uint8_t *__restrict__ bytes, const int64_t *__restrict__ row_off,
                                                      int n_rows, int n_pad, int n_chunks, uint32_t *__restrict__ planes) {
    int r = blockIdx.x * kBlock + threadIdx.x;
    int c = blockIdx.y;
    if (r >= n_pad) return;
    uint32_t mA = 0, mC = 0, mG = 0, mT = 0;
    if (r < n_rows) {
        int64_t o = row_off[r];
        int64_t len = row_off[r + 1] - o;
        int64_t col0 = (int64_t)c * 32;
        const uint8_t *p = bytes + o + col0;
        int n = (int)(len - col0 < 32 ? (len - col0 < 0 ? 0 : len - col0) : 32);
        for (int j = 0; j < n; j++) {
            uint32_t code = c_code_lut[p[j]];
            mA |= (code & 1u) << j;
            mC |= ((code >> 1) & 1u) << j;
            mG |= ((code >> 2) & 1u) << j;
            mT |= ((code >> 3) & 1u) << j;
        }
    }
    size_t base = ((size_t)c * 4) * n_pad + r;
    planes[base] = mA;
    planes[base + n_pad] = mC;
    planes[base + 2 * (size_t)n_pad] = mG;
    planes[base + 3 * (size_t)n_pad] = mT;
}

// thread = row: prefix count of residues per chunk, leading-gap length and right-stripped length
// (V20:625-627)
__global__ __launch_bounds__(kBlock) void row_scan_kernel(const uint32_t *__restrict__ planes, const int64_t *__restrict__ row_off,
                                                          int n_rows, int n_pad, int n_chunks, uint32_t *__restrict__ cum,
                                                          int32_t *__restrict__ lead, int32_t *__restrict__ rstrip,
                                                          int32_t *__restrict__ rlen) {
    int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_pad) return;
    uint32_t run = 0;
    int first = -1, last = 0;
    for (int c = 0; c < n_chunks; c++) {
        size_t base = ((size_t)c * 4) * n_pad + r;
        uint32_t ng = planes[base] | planes[base + n_pad] | planes[base + 2 * (size_t)n_pad] | planes[base + 3 * (size_t)n_pad];
        cum[(size_t)c * n_pad + r] = run;
        run += __popc(ng);
        if (ng) {
            if (first < 0) first = c * 32 + (__ffs(ng) - 1);
            last = c * 32 + 32 - __clz(ng);
        }
    }
    cum[(size_t)n_chunks * n_pad + r] = run;
    if (r < n_rows) {
        int len = (int)(row_off[r + 1] - row_off[r]);
        lead[r] = first < 0 ? len : first;
        rstrip[r] = last;
        rlen[r] = len;
    }
}

// thread = (row, chunk): append this chunk's residues to the row's gap-free code string
__global__ __launch_bounds__(kBlock) void ungap_kernel(const uint32_t *__restrict__ planes, const uint32_t *__restrict__ cum,
                                                       int n_rows, int n_pad, int ustride, uint32_t *__restrict__ ung) {
    int r = blockIdx.x * kBlock + threadIdx.x;
    int c = blockIdx.y;
    if (r >= n_rows) return;
    size_t base = ((size_t)c * 4) * n_pad + r;
    uint32_t mA = planes[base], mC = planes[base + n_pad], mG = planes[base + 2 * (size_t)n_pad], mT = planes[base + 3 * (size_t)n_pad];
    uint32_t ng = mA | mC | mG | mT;
    if (!ng) return;
    uint32_t pos = cum[(size_t)c * n_pad + r];
    uint32_t *dst = ung + (size_t)r * ustride;
    uint32_t word = 0;
    uint32_t widx = pos >> 3;
    while (ng) {
        int j = __ffs(ng) - 1;
        ng &= ng - 1;
        uint32_t code = ((mA >> j) & 1u) | (((mC >> j) & 1u) << 1) | (((mG >> j) & 1u) << 2) | (((mT >> j) & 1u) << 3);
        if ((pos >> 3) != widx) {
            atomicOr(dst + widx, word);
            word = 0;
            widx = pos >> 3;
        }
        word |= code << ((pos & 7) * 4);
        pos++;
    }
    atomicOr(dst + widx, word);
}

// Column planes for the bit-sliced evaluation: cols[col][3][Npad/64] u64, bit r%64 of word r/64 =
// sequence r; planes b0, b1 (2-bit base, 0 where gap) and g (gap / beyond the row's end).  IUPAC
// symbols produce arbitrary b0/b1 here: windows that touch one are always routed to the
// general path (patch list), never to the bit-sliced pass.
__global__ __launch_bounds__(kBlock) void colplane_kernel(const uint32_t *__restrict__ planes, int n_pad, int n_chunks,
                                                          unsigned long long *__restrict__ cols) {
    const int r = blockIdx.x * kBlock + threadIdx.x;     // n_pad is a multiple of kBlock: every lane is live
    const int c = blockIdx.y;
    const size_t np = (size_t)n_pad, nw = np / 64;
    const size_t base = ((size_t)c * 4) * np + r;
    const uint32_t mA = planes[base], mC = planes[base + np], mG = planes[base + 2 * np], mT = planes[base + 3 * np];
    const uint32_t b0 = mC | mT, b1 = mG | mT, ng = mA | mC | mG | mT;
    const int lane = threadIdx.x & 63;
    for (int j = 0; j < 32; j++) {
        unsigned long long x0 = __ballot((b0 >> j) & 1u), x1 = __ballot((b1 >> j) & 1u), xg = __ballot(!((ng >> j) & 1u));
        if (lane == 0) {
            unsigned long long *dst = cols + ((size_t)(c * 32 + j) * 3) * nw + (size_t)(r >> 6);
            dst[0] = x0; dst[nw] = x1; dst[2 * nw] = xg;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// (2) window k-mers with edge-gap repair (V20:666-687)
// ----------------------------------------------------------------------------------------------
struct Nib {          // up to 32 symbol codes, one nibble each
    uint64_t lo, hi;
    __device__ uint32_t get(int j) const { return (uint32_t)((j < 16 ? lo >> (4 * j) : hi >> (4 * (j - 16))) & 15u); }
    __device__ void set(int j, uint32_t v) {
        if (j < 16) lo = (lo & ~(15ull << (4 * j))) | ((uint64_t)v << (4 * j));
        else hi = (hi & ~(15ull << (4 * (j - 16)))) | ((uint64_t)v << (4 * (j - 16)));
    }
    __device__ void shift_up(int n) {      // move every nibble n positions towards the 3' end
        int s = 4 * n;
        if (s == 0) return;
        if (s >= 64) { hi = lo << (s - 64); lo = 0; }
        else { hi = (hi << s) | (lo >> (64 - s)); lo <<= s; }
    }
};

__device__ inline uint32_t ung_get(const uint32_t *__restrict__ ung_row, uint32_t t) {
    return (ung_row[t >> 3] >> ((t & 7) * 4)) & 15u;
}

// The general path: rows whose window starts or ends in a gap, holds an IUPAC code, or runs past
// the end of a ragged row.  Follows get_primers line by line.  Returns 0 = store words,
// 1 = exception (IUPAC code present, `buf` returned), 2 = fewer than k residues (V20:683-687).
__device__ int repair_window(uint32_t wA, uint32_t wC, uint32_t wG, uint32_t wT, int k, int p, int len,
                             uint32_t c_left, uint32_t total, const uint32_t *__restrict__ ung_row,
                             uint32_t &b0, uint32_t &b1, uint32_t &g, Nib &buf) {
    uint32_t kmask = (k == 32) ? 0xFFFFFFFFu : ((1u << k) - 1u);
    int m = len - p;
    m = m < 0 ? 0 : (m > k ? k : m);
    uint32_t ng = (wA | wC | wG | wT) & kmask;
    buf.lo = buf.hi = 0;
    for (int j = 0; j < m; j++) {
        uint32_t code = ((wA >> j) & 1u) | (((wC >> j) & 1u) << 1) | (((wG >> j) & 1u) << 2) | (((wT >> j) & 1u) << 3);
        buf.set(j, code);
    }
    int n = m;
    bool all_gap = (m == k) && ng == 0;                       // V20:668
    if (!all_gap && n > 0) {
        if (buf.get(0) == 0) {                                // V20:671 sequence.startswith("-")
            int run = 0;
            while (run < n && buf.get(run) == 0) run++;
            if (c_left >= (uint32_t)run)                      // V20:675
                for (int t = 0; t < run; t++) buf.set(t, ung_get(ung_row, c_left - run + t));
        }
        if (buf.get(n - 1) == 0) {                            // V20:677 sequence.endswith("-")
            int run = 0;
            while (run < n && buf.get(n - 1 - run) == 0) run++;
            uint32_t c_after = c_left + __popc(ng);          // residues in s[0 : p+k]
            if (total - c_after >= (uint32_t)run)             // V20:681
                for (int t = 0; t < run; t++) buf.set(n - run + t, ung_get(ung_row, c_after + t));
        }
    }
    if (n < k) {                                              // V20:683
        int need = k - n;
        if (c_left < (uint32_t)need) return 2;
        buf.shift_up(need);
        for (int t = 0; t < need; t++) buf.set(t, ung_get(ung_row, c_left - need + t));
        n = k;
    }
    b0 = b1 = g = 0;
    bool iupac = false;
    for (int j = 0; j < k; j++) {
        uint32_t code = buf.get(j);
        if (code == 0) g |= 1u << j;
        else if (code & (code - 1)) iupac = true;
        else {
            uint32_t bi = __ffs(code) - 1;
            b0 |= (bi & 1u) << j;
            b1 |= (bi >> 1) << j;
        }
    }
    return iupac ? 1 : 0;
}

// ----------------------------------------------------------------------------------------------
// window words in HBM: one packed u64 per (window, sequence) when 3k <= 63, else three u32 planes
// ----------------------------------------------------------------------------------------------
template <bool P64>
struct WinView;

template <>
struct WinView<false> {
    const uint32_t *W0, *W1, *W2;
    __device__ WinView(const void *base, int w, size_t np, int, uint32_t)
        : W0((const uint32_t *)base + (size_t)w * 3 * np), W1(W0 + np), W2(W1 + np) {}
    __device__ inline void load(int r, uint32_t &b0, uint32_t &b1, uint32_t &g) const { b0 = W0[r]; b1 = W1[r]; g = W2[r]; }
    struct Raw4 { uint4 a, b, c; };
    __device__ inline Raw4 load4(int r) const {
        Raw4 q;
        q.a = *reinterpret_cast<const uint4 *>(W0 + r);
        q.b = *reinterpret_cast<const uint4 *>(W1 + r);
        q.c = *reinterpret_cast<const uint4 *>(W2 + r);
        return q;
    }
    __device__ inline void unpack(const Raw4 &q, int i, uint32_t &b0, uint32_t &b1, uint32_t &g) const {
        b0 = i == 0 ? q.a.x : i == 1 ? q.a.y : i == 2 ? q.a.z : q.a.w;
        b1 = i == 0 ? q.b.x : i == 1 ? q.b.y : i == 2 ? q.b.z : q.b.w;
        g = i == 0 ? q.c.x : i == 1 ? q.c.y : i == 2 ? q.c.z : q.c.w;
    }
    __device__ static inline void store(void *base, int w, size_t np, int r, uint32_t b0, uint32_t b1, uint32_t g, int, uint32_t) {
        uint32_t *W = (uint32_t *)base + (size_t)w * 3 * np + r;
        W[0] = b0; W[np] = b1; W[2 * np] = g;
    }
};

template <>
struct WinView<true> {
    const uint64_t *Wp;
    int k;
    uint32_t kmask;
    __device__ WinView(const void *base, int w, size_t np, int k_, uint32_t kmask_)
        : Wp((const uint64_t *)base + (size_t)w * np), k(k_), kmask(kmask_) {}
    __device__ inline void split(uint64_t x, uint32_t &b0, uint32_t &b1, uint32_t &g) const {
        b0 = (uint32_t)x & kmask;
        b1 = (uint32_t)(x >> k) & kmask;
        g = ((uint32_t)(x >> (2 * k)) & kmask) | ((uint32_t)(x >> 32) & MP_WIN_SKIP);
    }
    __device__ inline void load(int r, uint32_t &b0, uint32_t &b1, uint32_t &g) const { split(Wp[r], b0, b1, g); }
    struct Raw4 { uint4 a, b; };
    __device__ inline Raw4 load4(int r) const {
        Raw4 q;
        q.a = *reinterpret_cast<const uint4 *>(Wp + r);
        q.b = *reinterpret_cast<const uint4 *>(Wp + r + 2);
        return q;
    }
    __device__ inline void unpack(const Raw4 &q, int i, uint32_t &b0, uint32_t &b1, uint32_t &g) const {
        uint32_t lo = i == 0 ? q.a.x : i == 1 ? q.a.z : i == 2 ? q.b.x : q.b.z;
        uint32_t hi = i == 0 ? q.a.y : i == 1 ? q.a.w : i == 2 ? q.b.y : q.b.w;
        split(((uint64_t)hi << 32) | lo, b0, b1, g);
    }
    __device__ static inline void store(void *base, int w, size_t np, int r, uint32_t b0, uint32_t b1, uint32_t g, int k, uint32_t kmask) {
        uint64_t x = (uint64_t)b0 | ((uint64_t)b1 << k) | ((uint64_t)(g & kmask) << (2 * k)) |
                     ((uint64_t)(g & MP_WIN_SKIP) << 32);
        ((uint64_t *)base)[(size_t)w * np + r] = x;
    }
};

// thread = row, block = 256 rows x a tile of consecutive windows; the 32-column plane words slide
// in registers, so every plane word is read once per tile.
// Rows whose k-mer is NOT the plain column slice (edge-gap repair, IUPAC, ragged end) are flagged per
// (window, 64-row word) in `excl` and collected, per window, in a compact patch list of window words:
// the bit-sliced evaluation skips them, the row-per-lane evaluation handles exactly them.
// pass 0 writes the window words, the flags and the per-window patch counts; pass 1 (same
// computation) fills the patch list once the host has turned the counts into offsets.
struct PatchOut {
    int pass;
    unsigned long long *excl;     // [W][Npad/64]
    int32_t *count;               // [W]
    const int32_t *off;           // [W+1]   (pass 1)
    int32_t *cursor;              // [W]     (pass 1)
    uint32_t *words;              // [n][3]  (pass 1)
};

template <bool P64>
__global__ __launch_bounds__(kBlock) void build_windows_kernel(
    const uint32_t *__restrict__ planes, const uint32_t *__restrict__ cum, const uint32_t *__restrict__ ung,
    const int32_t *__restrict__ rlen, int n_rows, int n_pad, int n_chunks, int ustride, int p0, int n_win, int tile,
    int k, void *__restrict__ win, ExRec *__restrict__ ex, int ex_cap, int *__restrict__ ex_count,
    int *__restrict__ err, PatchOut po) {
    int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_pad) return;
    int w0 = blockIdx.y * tile;
    int w1 = w0 + tile < n_win ? w0 + tile : n_win;
    const uint32_t kmask = (1u << k) - 1u;
    const size_t np = (size_t)n_pad;
    if (r >= n_rows) {                       // padding rows never take part
        if (po.pass == 0)
            for (int w = w0; w < w1; w++) WinView<P64>::store(win, w, np, r, 0, 0, MP_WIN_SKIP | kmask, k, kmask);
        return;
    }
    const int len = rlen[r];
    const uint32_t total = cum[(size_t)n_chunks * np + r];
    const uint32_t *ung_row = ung + (size_t)r * ustride;
    int cur = -1;
    uint32_t loA = 0, loC = 0, loG = 0, loT = 0, hiA = 0, hiC = 0, hiG = 0, hiT = 0;
    for (int w = w0; w < w1; w++) {
        int p = p0 + w;
        int c = p >> 5, o = p & 31;
        if (c != cur) {
            size_t base = ((size_t)c * 4) * np + r;
            if (c == cur + 1 && cur >= 0) { loA = hiA; loC = hiC; loG = hiG; loT = hiT; }
            else { loA = planes[base]; loC = planes[base + np]; loG = planes[base + 2 * np]; loT = planes[base + 3 * np]; }
            size_t nb = base + 4 * np;           // chunk c+1 exists: n_chunks is padded by two
            hiA = planes[nb]; hiC = planes[nb + np]; hiG = planes[nb + 2 * np]; hiT = planes[nb + 3 * np];
            cur = c;
        }
        uint32_t wA = __funnelshift_r(loA, hiA, o) & kmask;
        uint32_t wC = __funnelshift_r(loC, hiC, o) & kmask;
        uint32_t wG = __funnelshift_r(loG, hiG, o) & kmask;
        uint32_t wT = __funnelshift_r(loT, hiT, o) & kmask;
        uint32_t o1 = wA | wC, a1 = wA & wC, o2 = wG | wT, a2 = wG & wT;
        uint32_t ng = o1 | o2;
        uint32_t multi = a1 | a2 | (o1 & o2);
        uint32_t gw = ~ng & kmask;
        uint32_t b0, b1, g;
        bool fast = (p + k <= len) && multi == 0 && (gw == kmask || ((gw & 1u) == 0 && (gw >> (k - 1)) == 0));
        if (fast) {
            b0 = wC | wT; b1 = wG | wT; g = gw;
        } else {
            uint32_t ng_lo = loA | loC | loG | loT;
            uint32_t c_left = cum[(size_t)c * np + r] + __popc(ng_lo & ((1u << o) - 1u));
            Nib buf;
            int rc = repair_window(wA, wC, wG, wT, k, p, len, c_left, total, ung_row, b0, b1, g, buf);
            if (rc == 1) {
                if (po.pass == 0) {
                    int idx = atomicAdd(ex_count, 1);
                    if (idx < ex_cap) { ex[idx].win = w; ex[idx].row = r; ex[idx].lo = buf.lo; ex[idx].hi = buf.hi; }
                }
                b0 = 0; b1 = 0; g = MP_WIN_SKIP | kmask;
            } else if (rc == 2) {
                atomicMax(err, 1);
                err[1] = w; err[2] = r;
                b0 = 0; b1 = 0; g = MP_WIN_SKIP | kmask;
            }
        }
        {
            // flags and patch list; the lanes of a wave that are still here are all real rows
            const unsigned long long live = __ballot(true);
            const unsigned long long flg = __ballot(!fast);
            const bool keep = !fast && !(g & MP_WIN_SKIP);
            const unsigned long long kp = __ballot(keep);
            const int lane = threadIdx.x & 63;
            const int leader = __ffsll((long long)live) - 1;
            if (po.pass == 0) {
                if (lane == leader) {
                    po.excl[(size_t)w * (np / 64) + (size_t)(r >> 6)] = flg;
                    if (kp) atomicAdd(&po.count[w], (int)__popcll(kp));
                }
            } else if (kp) {
                int base = 0;
                if (lane == leader) base = atomicAdd(&po.cursor[w], (int)__popcll(kp));
                base = __shfl(base, leader);
                if (keep) {
                    int slot = po.off[w] + base + (int)__popcll(kp & ((1ull << lane) - 1ull));
                    po.words[3 * (size_t)slot] = b0; po.words[3 * (size_t)slot + 1] = b1; po.words[3 * (size_t)slot + 2] = g;
                }
            }
        }
        if (po.pass == 0) WinView<P64>::store(win, w, np, r, b0, b1, g, k, kmask);
    }
}

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

// ----------------------------------------------------------------------------------------------
// (4) candidate x sequence evaluation (V20:1103-1130, Y_distance V20:229-233)
// ----------------------------------------------------------------------------------------------
// A candidate is held as four k-bit words nX = positions whose symbol does NOT contain base X.
// For a sequence k-mer (b0,b1,g) the mismatch word is  g | select(nA,nC,nG,nT by (b1,b0))  — three
// v_bfi_b32 and one v_or_b32.  With D = set bits of mm:
//   perfect = (mm == 0)
//   F_raw   = |D| <= v  and  mm & strictF == 0      (R_raw likewise)
// F_raw includes the perfect rows, so the kernel counts F_raw and subtracts `perfect` once per
// block (F_mis = F_raw - perfect), which saves the |D| != 0 test per evaluation.
// VMODE 1 (v == 1, the pipeline default): |D| <= 1  <=>  mm & (mm-1) == 0, so
//   F_raw <=> mm & ((mm-1) | strictF) == 0  — one v_add, one v_bitop3, one compare, no popcount.
__device__ inline uint32_t bfi(uint32_t s, uint32_t a, uint32_t b) { return (s & a) | (~s & b); }

struct EvalArgs {
    const void *win;
    int n_pad, k;
    const EvalItem *items;
    const uint4 *cand_n;        // [padded cand] nA,nC,nG,nT
    const int32_t *cand_out;    // [padded cand] index into out or -1
    const int32_t *extra_off;   // [W+1] or nullptr
    const uint32_t *extra_words;
    uint32_t sF, sR;
    int v;
    uint32_t kmask;
    int rows_per_split;
    unsigned long long *out;
};

// COUNT 0: per-lane VGPR accumulators (v_cmp + v_addc); COUNT 1: wave ballots counted on the
// scalar unit (v_cmp -> s_bcnt1_i32_b64 -> s_add), accumulators live in SGPRs.
template <int CC, int COUNT>
struct EvalAcc {
    uint32_t p[CC], f[CC], r[CC];
    __device__ inline void clear() {
#pragma unroll
        for (int c = 0; c < CC; c++) p[c] = f[c] = r[c] = 0;
    }
    __device__ inline void add(int c, bool pp, bool ff, bool rr) {
        if (COUNT == 0) {
            p[c] += pp; f[c] += ff; r[c] += rr;
        } else {
            p[c] += (uint32_t)__popcll(__ballot(pp));
            f[c] += (uint32_t)__popcll(__ballot(ff));
            r[c] += (uint32_t)__popcll(__ballot(rr));
        }
    }
};

// FORM 0: bfi select, operand placement left to the compiler (candidate words end up in SGPRs and
//         the one-SGPR-per-VALU constant-bus rule splits every v_bfi_b32 in two);
// FORM 1: candidate words pinned in VGPRs: three v_bfi_b32 + one v_or_b32 per evaluation;
// FORM 2: per-row one-hot words eqX (4 ops per row, shared by the candidates) and a chain of four
//         v_and_or_b32 with the candidate words as the single SGPR operand.
template <int CC, int VMODE, int COUNT, int FORM>
__device__ inline void eval_row(uint32_t b0, uint32_t b1, uint32_t g, const EvalArgs &A,
                                const uint32_t (&nA)[CC], const uint32_t (&nC)[CC], const uint32_t (&nG)[CC],
                                const uint32_t (&nT)[CC], EvalAcc<CC, COUNT> &acc) {
    uint32_t gk = g & A.kmask;
    // rows outside the universe (SKIP slots and k-mers with more than v gaps, V20:689) get an
    // all-ones mismatch word: 32 mismatches, counted nowhere
    if ((int)__popc(gk) > A.v) gk = 0xFFFFFFFFu;
    uint32_t eA = 0, eC = 0, eG = 0, eT = 0;
    if (FORM == 2) { eA = ~(b0 | b1 | gk); eC = b0 & ~b1; eG = b1 & ~b0; eT = b0 & b1; }
#pragma unroll
    for (int c = 0; c < CC; c++) {
        uint32_t mm;
        if (FORM == 2) mm = (eT & nT[c]) | ((eG & nG[c]) | ((eC & nC[c]) | ((eA & nA[c]) | gk)));
        else mm = bfi(b1, bfi(b0, nT[c], nG[c]), bfi(b0, nC[c], nA[c])) | gk;
        bool pp = mm == 0, ff, rr;
        if (VMODE == 0) {
            ff = rr = pp;
        } else if (VMODE == 1) {
            uint32_t t;
            if (FORM == 0) t = mm - 1u;
            else asm("v_add_u32_e32 %0, -1, %1" : "=v"(t) : "v"(mm));   // no carry-out: keeps `mm == 0` a plain v_cmp
            ff = (mm & (t | A.sF)) == 0;
            rr = (mm & (t | A.sR)) == 0;
        } else {
            bool le = (int)__popc(mm) <= A.v;
            ff = le && (mm & A.sF) == 0;
            rr = le && (mm & A.sR) == 0;
        }
        acc.add(c, pp, ff, rr);
    }
}

template <int CC, int VMODE, int COUNT, bool PREFETCH, int FORM, bool P64>
__global__ __launch_bounds__(kBlock) void eval_kernel(const EvalArgs A) {
    __shared__ uint32_t s_acc[3 * CC];
    const EvalItem it = A.items[blockIdx.x];
    uint32_t nA[CC], nC[CC], nG[CC], nT[CC];
    EvalAcc<CC, COUNT> acc;
    acc.clear();
#pragma unroll
    for (int c = 0; c < CC; c++) {
        uint4 q = A.cand_n[it.cand0 + c];
        nA[c] = q.x; nC[c] = q.y; nG[c] = q.z; nT[c] = q.w;
        if (FORM == 1) {
            asm volatile("" : "+v"(nA[c]));
            asm volatile("" : "+v"(nC[c]));
            asm volatile("" : "+v"(nG[c]));
            asm volatile("" : "+v"(nT[c]));
        }
    }
    if (threadIdx.x < 3 * CC) s_acc[threadIdx.x] = 0;
    const size_t np = (size_t)A.n_pad;
    const WinView<P64> V(A.win, it.win, np, A.k, A.kmask);
    typedef typename WinView<P64>::Raw4 Raw4;
    const int r0 = blockIdx.y * A.rows_per_split;
    const int r1 = r0 + A.rows_per_split < A.n_pad ? r0 + A.rows_per_split : A.n_pad;
    // 4 consecutive sequences per lane and iteration, 16-byte loads (n_pad % 4 == 0)
    int r = r0 + threadIdx.x * 4;
    Raw4 cur;
    if (PREFETCH && r < r1) cur = V.load4(r);
#pragma unroll 1
    while (r < r1) {
        const int rn = r + kBlock * 4;
        Raw4 now;
        if (PREFETCH) {
            now = cur;
            if (rn < r1) cur = V.load4(rn);          // next group in flight while this one computes
        } else {
            now = V.load4(r);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t b0, b1, g;
            V.unpack(now, i, b0, b1, g);
            eval_row<CC, VMODE, COUNT, FORM>(b0, b1, g, A, nA, nC, nG, nT, acc);
        }
        r = rn;
    }
    if (blockIdx.y == 0 && A.extra_off) {      // host-expanded IUPAC rows of this window
        const int e0 = A.extra_off[it.win], e1 = A.extra_off[it.win + 1];
        for (int eb = e0; eb < e1; eb += kBlock) {   // uniform trip count: COUNT 1 ballots need every lane
            int e = eb + threadIdx.x;
            uint32_t b0 = 0, b1 = 0, g = 0xFFFFFFFFu;
            if (e < e1) { b0 = A.extra_words[3 * e]; b1 = A.extra_words[3 * e + 1]; g = A.extra_words[3 * e + 2]; }
            eval_row<CC, VMODE, COUNT, FORM>(b0, b1, g, A, nA, nC, nG, nT, acc);
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CC; c++) {
        uint32_t x = acc.p[c], y = acc.f[c], z = acc.r[c];
        if (COUNT == 0) {
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) {
                x += __shfl_xor(x, s);
                y += __shfl_xor(y, s);
                z += __shfl_xor(z, s);
            }
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&s_acc[3 * c], x);
            atomicAdd(&s_acc[3 * c + 1], y - x);      // F_mis = F_raw - perfect
            atomicAdd(&s_acc[3 * c + 2], z - x);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * CC) {
        int oc = A.cand_out[it.cand0 + threadIdx.x / 3];
        uint32_t val = s_acc[threadIdx.x];
        if (oc >= 0 && val) atomicAdd(&A.out[(size_t)oc * 3 + threadIdx.x % 3], (unsigned long long)val);
    }
}


// ----------------------------------------------------------------------------------------------
// (4b) bit-sliced evaluation: 64 sequences per register word
// ----------------------------------------------------------------------------------------------
// For the sequences whose k-mer at window w is the plain column slice (everything the patch list
// does not hold) the symbol at window position j is column p0+w+j of the alignment, so the
// evaluation can run on the COLUMN planes: a thread owns G words of 64 sequences; for every
// position it loads the three plane words once, and for every candidate the mismatch word of 64
// sequences is ONE v_bitop3 of (g,b1,b0) whose truth table is fixed by the candidate's symbol
// (a wave-uniform 16-way dispatch).  Mismatch counts are bit-sliced saturating counters
// (t1 = ">= 1", t2 = ">= 2", t3 = ">= 3": LV = v+1 levels), the strict-position sets are two more
// words, and the three coverage counters are popcounts at the end.  ~2.5 VALU per evaluation
// instead of ~13, and the inputs (N*L*3/8 bytes) stay in L2 / Infinity Cache.
struct EvalBitsArgs {
    const unsigned long long *cols;    // [n_cols][3][nw]
    const unsigned long long *excl;    // [W][nw]
    int nw, p0, k, v;
    const EvalItem *items;
    const uint32_t *cand_symT;         // [item][32] u32: nibble c of word j = symbol of candidate c at position j
    const int32_t *cand_out;
    uint32_t sF, sR;
    unsigned long long *out;
    int ny, ny_pad;                    // row slices per item; ny_pad = ny rounded up to a multiple of 8 (XCDs)
};

// truth table of v_bitop3_b32 D = f(S0,S1,S2): bit (S0<<2 | S1<<1 | S2) of the immediate
constexpr int bs_lut_mismatch(int sym) {      // inputs (g, b1, b0): gap, or base not in the symbol's set
    int t = 0;
    for (int idx = 0; idx < 8; idx++) {
        int g = idx >> 2, base = idx & 3;
        if (g || !((sym >> base) & 1)) t |= 1 << idx;
    }
    return t;
}
constexpr int kLutOrAnd = 0xF8;               // S0 | (S1 & S2)
constexpr int kLutAndNotNot = 0x10;           // S0 & ~S1 & ~S2

template <int SYM, int GW>
__device__ inline void bs_mismatch(const uint32_t (&b0)[GW], const uint32_t (&b1)[GW], const uint32_t (&g)[GW], uint32_t (&m)[GW]) {
    constexpr int lut = bs_lut_mismatch(SYM);
#pragma unroll
    for (int i = 0; i < GW; i++) m[i] = __builtin_amdgcn_bitop3_b32(g[i], b1[i], b0[i], lut);
}

// CP candidates per pass over the k positions (8 / CP passes), GW 32-bit words (32 sequences each)
// per thread, LV = v + 1 saturating counter levels.
template <int CP, int LV, int GW, bool PREFETCH>
__global__ __launch_bounds__(kBlock) void eval_bits_kernel(const EvalBitsArgs A) {
    constexpr int CC = 8;
    __shared__ uint32_t s_acc[3 * CC];
    // XCD-aware block mapping: workgroup b runs on XCD b % 8 (observed dispatch order), so all blocks of
    // one row slice land on the same XCD and consecutive windows re-read their 17 shared columns from
    // that XCD's L2 (a slice of the planes is 1/ny of N*L*3/8 bytes)
    const int slice = blockIdx.x % A.ny_pad, item = blockIdx.x / A.ny_pad;
    if (slice >= A.ny) return;
    const EvalItem it = A.items[item];
    if (threadIdx.x < 3 * CC) s_acc[threadIdx.x] = 0;
    const size_t nw32 = (size_t)A.nw * 2;             // 32-bit words per plane row
    const int word0 = (slice * kBlock + threadIdx.x) * GW;
    const bool live = word0 < (int)nw32;              // nw32 % GW == 0 (n_pad % 256 == 0, GW <= 8)
    const uint32_t *cols = reinterpret_cast<const uint32_t *>(A.cols);
    uint32_t accP[CC], accF[CC], accR[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) accP[c] = accF[c] = accR[c] = 0;
    if (live) {
        uint32_t valid[GW];
        bool have_valid = false;
#pragma unroll 1
        for (int pass = 0; pass < CC / CP; pass++) {
            uint32_t t1[CP][GW], t2[CP][GW], t3[CP][GW], sf[CP][GW], sr[CP][GW];
            uint32_t g1[GW], g2[GW], g3[GW];
#pragma unroll
            for (int i = 0; i < GW; i++) {
                g1[i] = g2[i] = g3[i] = 0;
#pragma unroll
                for (int c = 0; c < CP; c++) t1[c][i] = t2[c][i] = t3[c][i] = sf[c][i] = sr[c][i] = 0;
            }
            uint32_t n0[GW], n1[GW], ng[GW];               // next position's planes, in flight during this one
            if (PREFETCH) {
                const uint32_t *P = cols + ((size_t)(A.p0 + it.win) * 3) * nw32 + word0;
#pragma unroll
                for (int i = 0; i < GW; i++) { n0[i] = P[i]; n1[i] = P[nw32 + i]; ng[i] = P[2 * nw32 + i]; }
            }
#pragma unroll 1
            for (int j = 0; j < A.k; j++) {
                uint32_t b0[GW], b1[GW], g[GW];
                if (PREFETCH) {
#pragma unroll
                    for (int i = 0; i < GW; i++) { b0[i] = n0[i]; b1[i] = n1[i]; g[i] = ng[i]; }
                    if (j + 1 < A.k) {
                        const uint32_t *P = cols + ((size_t)(A.p0 + it.win + j + 1) * 3) * nw32 + word0;
#pragma unroll
                        for (int i = 0; i < GW; i++) { n0[i] = P[i]; n1[i] = P[nw32 + i]; ng[i] = P[2 * nw32 + i]; }
                    }
                } else {
                    const uint32_t *P = cols + ((size_t)(A.p0 + it.win + j) * 3) * nw32 + word0;
#pragma unroll
                    for (int i = 0; i < GW; i++) { b0[i] = P[i]; b1[i] = P[nw32 + i]; g[i] = P[2 * nw32 + i]; }
                }
                if (!have_valid) {
#pragma unroll
                    for (int i = 0; i < GW; i++) {            // gaps per k-mer, saturating (V20:689 needs "> v")
                        if (LV >= 3) g3[i] = __builtin_amdgcn_bitop3_b32(g3[i], g2[i], g[i], kLutOrAnd);
                        if (LV >= 2) g2[i] = __builtin_amdgcn_bitop3_b32(g2[i], g1[i], g[i], kLutOrAnd);
                        g1[i] |= g[i];
                    }
                }
                const uint32_t sw = __builtin_amdgcn_readfirstlane(A.cand_symT[(size_t)item * 32 + j]) >> (4 * CP * pass);
                // the mismatch word of every possible candidate symbol at this position (15 x GW v_bitop3,
                // shared by all candidates); a candidate then picks its word by a wave-uniform register index
                uint32_t tab[16][GW];
#pragma unroll
                for (int i = 0; i < GW; i++) tab[0][i] = 0xFFFFFFFFu;
                bs_mismatch<1, GW>(b0, b1, g, tab[1]); bs_mismatch<2, GW>(b0, b1, g, tab[2]); bs_mismatch<3, GW>(b0, b1, g, tab[3]);
                bs_mismatch<4, GW>(b0, b1, g, tab[4]); bs_mismatch<5, GW>(b0, b1, g, tab[5]); bs_mismatch<6, GW>(b0, b1, g, tab[6]);
                bs_mismatch<7, GW>(b0, b1, g, tab[7]); bs_mismatch<8, GW>(b0, b1, g, tab[8]); bs_mismatch<9, GW>(b0, b1, g, tab[9]);
                bs_mismatch<10, GW>(b0, b1, g, tab[10]); bs_mismatch<11, GW>(b0, b1, g, tab[11]); bs_mismatch<12, GW>(b0, b1, g, tab[12]);
                bs_mismatch<13, GW>(b0, b1, g, tab[13]); bs_mismatch<14, GW>(b0, b1, g, tab[14]); bs_mismatch<15, GW>(b0, b1, g, tab[15]);
                uint32_t m[CP][GW];
#pragma unroll
                for (int c = 0; c < CP; c++) {
                    const uint32_t sy = (sw >> (4 * c)) & 15u;
#pragma unroll
                    for (int i = 0; i < GW; i++) m[c][i] = tab[sy][i];
#pragma unroll
                    for (int i = 0; i < GW; i++) {
                        if (LV >= 3) t3[c][i] = __builtin_amdgcn_bitop3_b32(t3[c][i], t2[c][i], m[c][i], kLutOrAnd);
                        if (LV >= 2) t2[c][i] = __builtin_amdgcn_bitop3_b32(t2[c][i], t1[c][i], m[c][i], kLutOrAnd);
                        t1[c][i] |= m[c][i];
                    }
                }
                if (__builtin_amdgcn_readfirstlane((A.sF >> j) & 1u)) {
#pragma unroll
                    for (int c = 0; c < CP; c++)
#pragma unroll
                        for (int i = 0; i < GW; i++) sf[c][i] |= m[c][i];
                }
                if (__builtin_amdgcn_readfirstlane((A.sR >> j) & 1u)) {
#pragma unroll
                    for (int c = 0; c < CP; c++)
#pragma unroll
                        for (int i = 0; i < GW; i++) sr[c][i] |= m[c][i];
                }
            }
            if (!have_valid) {
                const uint32_t *E = reinterpret_cast<const uint32_t *>(A.excl) + (size_t)it.win * nw32 + word0;
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    const uint32_t gapbad = LV == 1 ? g1[i] : (LV == 2 ? g2[i] : g3[i]);
                    valid[i] = ~(E[i] | gapbad);
                }
                have_valid = true;
            }
#pragma unroll
            for (int c = 0; c < CP; c++) {
                uint32_t p = 0, f = 0, r = 0;
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    const uint32_t far = LV == 1 ? t1[c][i] : (LV == 2 ? t2[c][i] : t3[c][i]);
                    p += __popc(valid[i] & ~t1[c][i]);
                    f += __popc(__builtin_amdgcn_bitop3_b32(valid[i], far, sf[c][i], kLutAndNotNot));
                    r += __popc(__builtin_amdgcn_bitop3_b32(valid[i], far, sr[c][i], kLutAndNotNot));
                }
                // static index into the accumulators: the pass loop is not unrolled, so select by pass
#pragma unroll
                for (int q = 0; q < CC / CP; q++)
                    if (pass == q) { accP[q * CP + c] += p; accF[q * CP + c] += f; accR[q * CP + c] += r; }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CC; c++) {
        uint32_t x = accP[c], y = accF[c], z = accR[c];
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            x += __shfl_xor(x, sft);
            y += __shfl_xor(y, sft);
            z += __shfl_xor(z, sft);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&s_acc[3 * c], x);
            atomicAdd(&s_acc[3 * c + 1], y - x);      // F_mis = F_raw - perfect
            atomicAdd(&s_acc[3 * c + 2], z - x);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * CC) {
        int oc = A.cand_out[it.cand0 + threadIdx.x / 3];
        uint32_t val = s_acc[threadIdx.x];
        if (oc >= 0 && val) atomicAdd(&A.out[(size_t)oc * 3 + threadIdx.x % 3], (unsigned long long)val);
    }
}

// Row-per-lane evaluation of two compact per-window lists of window words: the patch list (rows with
// edge-gap repair / ragged ends, built on the device) and the host-expanded IUPAC rows.
struct EvalListArgs {
    const EvalItem *items;
    const uint4 *cand_n;
    const int32_t *cand_out;
    const int32_t *off_a;
    const uint32_t *words_a;
    const int32_t *off_b;       // may be nullptr
    const uint32_t *words_b;
    uint32_t sF, sR;
    int v;
    uint32_t kmask;
    unsigned long long *out;
};

template <int CC, int VMODE>
__global__ __launch_bounds__(kBlock) void eval_list_kernel(const EvalListArgs L) {
    __shared__ uint32_t s_acc[3 * CC];
    const EvalItem it = L.items[blockIdx.x];
    uint32_t nA[CC], nC[CC], nG[CC], nT[CC];
    EvalAcc<CC, 1> acc;
    acc.clear();
#pragma unroll
    for (int c = 0; c < CC; c++) {
        uint4 q = L.cand_n[it.cand0 + c];
        nA[c] = q.x; nC[c] = q.y; nG[c] = q.z; nT[c] = q.w;
    }
    if (threadIdx.x < 3 * CC) s_acc[threadIdx.x] = 0;
    EvalArgs A;
    A.sF = L.sF; A.sR = L.sR; A.v = L.v; A.kmask = L.kmask;
    for (int which = 0; which < 2; which++) {
        const int32_t *off = which ? L.off_b : L.off_a;
        const uint32_t *words = which ? L.words_b : L.words_a;
        if (!off) continue;
        const int e0 = off[it.win], e1 = off[it.win + 1];
        for (int eb = e0 + blockIdx.y * kBlock; eb < e1; eb += gridDim.y * kBlock) {     // uniform per wave
            int e = eb + threadIdx.x;
            uint32_t b0 = 0, b1 = 0, g = 0xFFFFFFFFu;
            if (e < e1) { b0 = words[3 * (size_t)e]; b1 = words[3 * (size_t)e + 1]; g = words[3 * (size_t)e + 2]; }
            eval_row<CC, VMODE, 1, 2>(b0, b1, g, A, nA, nC, nG, nT, acc);
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CC; c++) {
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&s_acc[3 * c], acc.p[c]);
            atomicAdd(&s_acc[3 * c + 1], acc.f[c] - acc.p[c]);
            atomicAdd(&s_acc[3 * c + 2], acc.r[c] - acc.p[c]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * CC) {
        int oc = L.cand_out[it.cand0 + threadIdx.x / 3];
        uint32_t val = s_acc[threadIdx.x];
        if (oc >= 0 && val) atomicAdd(&L.out[(size_t)oc * 3 + threadIdx.x % 3], (unsigned long long)val);
    }
}

typedef void (*EvalBitsFn)(const EvalBitsArgs);
typedef void (*EvalListFn)(const EvalListArgs);

typedef void (*EvalFn)(const EvalArgs);
struct EvalVariant { const char *name; EvalFn fn[2][3]; };     // fn[P64][VMODE]
#define EVAL_VARIANT(name, COUNT, PREFETCH, FORM)                                                         \
    { name, { { eval_kernel<kEvalCC, 0, COUNT, PREFETCH, FORM, false>, eval_kernel<kEvalCC, 1, COUNT, PREFETCH, FORM, false>, \
                eval_kernel<kEvalCC, 2, COUNT, PREFETCH, FORM, false> },                                   \
              { eval_kernel<kEvalCC, 0, COUNT, PREFETCH, FORM, true>, eval_kernel<kEvalCC, 1, COUNT, PREFETCH, FORM, true>,   \
                eval_kernel<kEvalCC, 2, COUNT, PREFETCH, FORM, true> } } }
// variant 0 is the default; the others exist to be measured (tools/variant_bench.py, MP_EVAL_VARIANT)
const EvalVariant kEvalVariants[] = {
    EVAL_VARIANT("ballot+prefetch/onehot", 1, true, 2),      // default: fastest measured (profiles/r01_variants.txt)
    EVAL_VARIANT("ballot+prefetch/bfi-vgpr", 1, true, 1),
    EVAL_VARIANT("ballot+prefetch/bfi-sgpr", 1, true, 0),
    EVAL_VARIANT("lane-acc+prefetch/onehot", 0, true, 2),
    EVAL_VARIANT("ballot/onehot", 1, false, 2),
};
constexpr int kNumEvalVariants = (int)(sizeof(kEvalVariants) / sizeof(kEvalVariants[0]));


// ----------------------------------------------------------------------------------------------
// (5) 3'-end dimer scan (finDimer_V4.py:191-224 "FD", get_Maxprimerset_V1.3.py:193-215 "MS")
// ----------------------------------------------------------------------------------------------
// thread = one (x, y) primer pair.  Primers are <= 32 symbols: the symbol codes sit in a 2 x u64
// nibble buffer, a concrete expansion is a 2-bit-packed u64, "RC(end) occurs in p at idx" is one
// shift-mask-compare per offset, GC content one popcount.  The two floating-point decisions come
// in as an exact byte table (Loss >= threshold) and as double constants that are only ADDED, in
// the reference's order, with __dadd_rn (no contraction, no multiply): the same doubles as CPython.
__constant__ uint8_t c_msize[16];
__constant__ uint8_t c_member[16][4];

struct DimerArgs {
    const uint8_t *codes;
    const int32_t *off;
    int n, mode, n_new;
    const uint8_t *loss_hit;
    const double *dg;
    double dg_limit;
    long long cap;
    int32_t *hits;
    unsigned long long *n_hits;
};

__device__ inline uint32_t dm_degeneracy(const Nib &c, int start, int len) {
    uint32_t d = 1;
    for (int p = 0; p < len; p++) d *= c_msize[c.get(start + p)];
    return d;
}

// expansion number idx (itertools.product order, last position fastest) as 2 bits per base, position 0 lowest
__device__ inline uint64_t dm_expand(const Nib &c, int start, int len, uint32_t idx) {
    uint64_t x = 0;
    for (int p = len - 1; p >= 0; p--) {
        uint32_t code = c.get(start + p);
        uint32_t sz = c_msize[code];
        uint32_t ch = idx % sz;
        idx /= sz;
        x |= (uint64_t)c_member[code][ch] << (2 * p);
    }
    return x;
}

__device__ inline double dm_delta_g(uint64_t e, int l, const double *__restrict__ dg) {
    double g = 0.0;
    uint32_t prev = (uint32_t)e & 3u;
    for (int t = 1; t < l; t++) {
        uint32_t cur = (uint32_t)(e >> (2 * t)) & 3u;
        g = __dadd_rn(g, dg[cur * 4 + prev]);                        // FD:176-178
        prev = cur;
    }
    uint32_t first = (uint32_t)e & 3u, last = (uint32_t)(e >> (2 * (l - 1))) & 3u;
    int ta = l >= 2 && ((uint32_t)(e >> (2 * (l - 2))) & 3u) == 3u && last == 0u;    // end[-2:] == "TA", FD:179
    g = __dadd_rn(g, dg[16 + (first * 4 + last) * 2 + ta]);          // FD:181-183
    g = __dadd_rn(g, -dg[48 + l]);                                   // FD:185
    bool sym = (l & 1) == 0;                                         // FD:115-125
    for (int t = 0; sym && t < l / 2; t++)
        sym = ((uint32_t)(e >> (2 * t)) & 3u) == (3u - ((uint32_t)(e >> (2 * (l / 2 + t))) & 3u));
    if (sym) g = __dadd_rn(g, dg[48 + MP_DIMER_MAX_LEN + 1]);        // FD:186-187
    return g;
}

// one ordered pair x -> y: first passing (end length, end expansion, y expansion); returns true on a hit
__device__ inline bool dimer_pair_scan(const uint8_t *__restrict__ codes, const int32_t *__restrict__ off, int x, int y,
                                       int mode, const uint8_t *__restrict__ loss_hit, const double *__restrict__ dg,
                                       double dg_limit, int32_t (&rec)[4]) {
    const int lx = off[x + 1] - off[x], ly = off[y + 1] - off[y];
    Nib cx, cy;
    cx.lo = cx.hi = cy.lo = cy.hi = 0;
    for (int p = 0; p < lx; p++) cx.set(p, codes[off[x] + p]);
    for (int p = 0; p < ly; p++) cy.set(p, codes[off[y] + p]);
    const uint32_t dy = dm_degeneracy(cy, 0, ly);
    int l_hi, l_lo;
    if (mode == 0) { l_hi = lx < 18 ? lx : 18; l_lo = lx < 5 ? lx : 5; }         // FD:162-169
    else { l_hi = lx - 1; l_lo = 5; }                                             // MS:149-154
    for (int l = l_hi; l >= l_lo; l--) {
        if (l <= 0 || l > ly) continue;                                           // cannot occur in a shorter primer
        const uint32_t de = dm_degeneracy(cx, lx - l, l);
        const uint64_t mask = l == 32 ? ~0ull : ((1ull << (2 * l)) - 1ull);
        for (uint32_t ei = 0; ei < de; ei++) {
            const uint64_t e = dm_expand(cx, lx - l, l, ei);
            uint64_t rc = 0;                                                      // reverse complement: 3 - base, reversed
            for (int t = 0; t < l; t++) rc |= (uint64_t)(3u - ((uint32_t)(e >> (2 * (l - 1 - t))) & 3u)) << (2 * t);
            const int gc = __popcll((e ^ (e >> 1)) & 0x5555555555555555ull & mask);   // C = 01, G = 10
            for (uint32_t pi = 0; pi < dy; pi++) {
                const uint64_t p = dm_expand(cy, 0, ly, pi);
                int idx = -1;
                for (int s0 = 0; s0 + l <= ly; s0++)
                    if (((p >> (2 * s0)) & mask) == rc) { idx = s0; break; }      // str.find: first occurrence
                if (idx < 0) continue;
                const int d2 = ly - l - idx;
                bool hit = loss_hit[((size_t)l * (MP_DIMER_MAX_LEN + 1) + gc) * 64 + d2] != 0;
                if (!hit && d2 == 0) hit = dm_delta_g(e, l, dg) < dg_limit;
                if (hit) { rec[0] = l; rec[1] = (int32_t)ei; rec[2] = (int32_t)pi; rec[3] = idx; return true; }
            }
        }
    }
    return false;
}

__global__ __launch_bounds__(kBlock) void dimer_kernel(const DimerArgs A) {
    const int x = blockIdx.x;
    const int y = blockIdx.y * kBlock + threadIdx.x;
    if (y >= A.n) return;
    if (A.mode == 0 ? (y < x) : (x >= A.n_new && y >= A.n_new)) return;
    int32_t rec[4];
    if (dimer_pair_scan(A.codes, A.off, x, y, A.mode, A.loss_hit, A.dg, A.dg_limit, rec)) {
        unsigned long long h = atomicAdd(A.n_hits, 1ull);
        if ((long long)h < A.cap) {
            int32_t *r = A.hits + 6 * h;
            r[0] = x; r[1] = y; r[2] = rec[0]; r[3] = rec[1]; r[4] = rec[2]; r[5] = rec[3];
        }
    }
}

// explicit ordered pairs (get_multiPrime_V8.py:419-438): one thread per pair, any passing combination
__global__ __launch_bounds__(kBlock) void dimer_pairs_kernel(const uint8_t *__restrict__ codes, const int32_t *__restrict__ off,
                                                             long long n_pairs, const int32_t *__restrict__ pairs,
                                                             const uint8_t *__restrict__ loss_hit, const double *__restrict__ dg,
                                                             double dg_limit, uint8_t *__restrict__ flags) {
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_pairs) return;
    int32_t rec[4];
    flags[p] = dimer_pair_scan(codes, off, pairs[2 * p], pairs[2 * p + 1], 0, loss_hit, dg, dg_limit, rec) ? 1 : 0;
}

// popcount(A[i] | B[j]) per pair: one wave per pair, lanes stride over the set's words (get_multiPrime_V8.py:560-569)
__global__ __launch_bounds__(kBlock) void pair_coverage_kernel(const unsigned long long *__restrict__ a,
                                                               const unsigned long long *__restrict__ b, int n_words,
                                                               long long n_pairs, const int32_t *__restrict__ pairs,
                                                               int32_t *__restrict__ out) {
    const long long p = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (p >= n_pairs) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long *A = a + (size_t)pairs[2 * p] * n_words, *B = b + (size_t)pairs[2 * p + 1] * n_words;
    int cnt = 0;
    for (int w = lane; w < n_words; w += 64) cnt += __popcll(A[w] | B[w]);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) cnt += __shfl_xor(cnt, s);
    if (lane == 0) out[p] = cnt;
}

}  // namespace

// ================================================================================================
// host side of the C ABI
// ================================================================================================
struct mp_ctx {
    char err[512] = {0};
    int dev = 0;
    hipStream_t stream = nullptr;
    int64_t bytes = 0;
    // alignment
    int n_rows = 0, n_pad = 0, n_chunks = 0, max_len = 0, ustride = 0;
    uint32_t *planes = nullptr, *cum = nullptr, *ung = nullptr;
    unsigned long long *cols = nullptr;      // [n_chunks*32][3][n_pad/64]
    int32_t *lead = nullptr, *rstrip = nullptr, *rlen = nullptr;
    // windows
    int p0 = 0, n_win = 0, k = 0, v = 0;
    void *win = nullptr;
    unsigned long long *excl = nullptr;      // [W][n_pad/64]
    int32_t *patch_count = nullptr, *patch_off = nullptr, *patch_cursor = nullptr;
    uint32_t *patch_words = nullptr;
    int n_patch = 0, max_patch = 0;
    bool p64 = false;
    size_t win_bytes = 0;
    ExRec *ex = nullptr;
    int ex_cap = 0;
    int *ex_count = nullptr, *err_flag = nullptr;
    std::vector<ExRec> ex_host;
    int32_t *extra_off = nullptr;
    uint32_t *extra_words = nullptr;
    int n_extra = 0;
    // unique
    long long u_cap = 0, u_n = 0;
    uint32_t *u_b0 = nullptr, *u_b1 = nullptr, *u_g = nullptr;
    int32_t *u_count = nullptr, *u_first = nullptr, *labels = nullptr, *u_over = nullptr, *u_wcount = nullptr;
    int64_t *u_wbase = nullptr;
    unsigned long long *u_total = nullptr;
    std::vector<int64_t> h_wbase;
    std::vector<int32_t> h_wcount;
    // eval staging
    int n_cand = 0, n_items = 0, n_padded = 0;
    EvalItem *items = nullptr;
    uint4 *cand_n = nullptr;
    uint32_t *cand_symT = nullptr;
    int32_t *cand_out = nullptr;
    uint32_t sF = 0, sR = 0;
    unsigned long long *tmp_out = nullptr;
    int tmp_out_n = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_busy, ev_free;
    double ev_ms = 0;
    int ev_n = 0;
    int eval_variant = 0;
};

namespace {

int fail(mp_ctx *c, int code, const char *fmt, ...) {
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

template <typename T>
int dev_alloc(mp_ctx *c, T **p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e != hipSuccess) return fail(c, MP_ERR_NOMEM, "hipMalloc(%zu bytes): %s", n * sizeof(T), hipGetErrorString(e));
    c->bytes += (int64_t)(n * sizeof(T));
    return MP_OK;
}

template <typename T>
void dev_free(mp_ctx *c, T **p, size_t n) {
    if (*p) {
        (void)hipFree(*p);
        c->bytes -= (int64_t)((n ? n : 1) * sizeof(T));
        *p = nullptr;
    }
}

void free_eval(mp_ctx *c) {
    dev_free(c, &c->items, (size_t)c->n_items);
    dev_free(c, &c->cand_n, (size_t)c->n_padded);
    dev_free(c, &c->cand_out, (size_t)c->n_padded);
    dev_free(c, &c->cand_symT, (size_t)c->n_items * 32);
    c->n_items = c->n_padded = c->n_cand = 0;
}

void free_unique(mp_ctx *c) {
    size_t cap = (size_t)c->u_cap, W = (size_t)c->n_win;
    dev_free(c, &c->u_b0, cap); dev_free(c, &c->u_b1, cap); dev_free(c, &c->u_g, cap);
    dev_free(c, &c->u_count, cap); dev_free(c, &c->u_first, cap);
    dev_free(c, &c->labels, W * c->n_pad);
    dev_free(c, &c->u_over, W); dev_free(c, &c->u_wcount, W); dev_free(c, &c->u_wbase, W);
    dev_free(c, &c->u_total, 1);
    c->u_cap = c->u_n = 0;
    c->h_wbase.clear(); c->h_wcount.clear();
}

void free_windows(mp_ctx *c) {
    free_eval(c);
    free_unique(c);
    if (c->win) { (void)hipFree(c->win); c->bytes -= (int64_t)c->win_bytes; c->win = nullptr; c->win_bytes = 0; }
    dev_free(c, &c->excl, (size_t)c->n_win * (c->n_pad / 64));
    dev_free(c, &c->patch_count, (size_t)c->n_win);
    dev_free(c, &c->patch_off, (size_t)c->n_win + 1);
    dev_free(c, &c->patch_cursor, (size_t)c->n_win);
    dev_free(c, &c->patch_words, (size_t)3 * c->n_patch);
    c->n_patch = c->max_patch = 0;
    dev_free(c, &c->ex, (size_t)c->ex_cap);
    dev_free(c, &c->ex_count, 1);
    dev_free(c, &c->err_flag, 4);
    dev_free(c, &c->extra_off, (size_t)c->n_win + 1);
    dev_free(c, &c->extra_words, (size_t)3 * c->n_extra);
    c->ex_cap = 0; c->n_extra = 0; c->n_win = 0;
    c->ex_host.clear();
}

void free_msa(mp_ctx *c) {
    free_windows(c);
    size_t np = (size_t)c->n_pad;
    dev_free(c, &c->planes, (size_t)c->n_chunks * 4 * np);
    dev_free(c, &c->cols, (size_t)c->n_chunks * 32 * 3 * (np / 64));
    dev_free(c, &c->cum, ((size_t)c->n_chunks + 1) * np);
    dev_free(c, &c->ung, (size_t)c->n_rows * c->ustride);
    dev_free(c, &c->lead, np); dev_free(c, &c->rstrip, np); dev_free(c, &c->rlen, np);
    c->n_rows = c->n_pad = c->n_chunks = 0;
}

}  // namespace

extern "C" {

const char *mp_backend_name(void) { return "hip"; }
const char *mp_last_error(const mp_ctx *c) { return c ? c->err : "mp_create failed: no usable HIP device"; }

int mp_create(int device, mp_ctx **out) {
    if (!out) return MP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return MP_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MP_ERR_DEVICE;
    mp_ctx *c = new mp_ctx();
    c->dev = device;
    uint8_t lut[256];
    host_code_lut(lut);
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_code_lut), lut, 256) != hipSuccess) { delete c; return MP_ERR_DEVICE; }
    {   // member order of every IUPAC symbol as the reference lists it (V20:105-107, FD:46-48), bases as A0 C1 G2 T3
        static const char *members[16] = {"", "A", "C", "AC", "G", "AG", "GC", "GAC", "T", "AT", "CT", "ATC", "GT", "GAT", "GTC", "ATGC"};
        uint8_t msize[16], member[16][4];
        for (int m = 0; m < 16; m++) {
            msize[m] = (uint8_t)strlen(members[m]);
            for (int t = 0; t < 4; t++) {
                char ch = t < msize[m] ? members[m][t] : 'A';
                member[m][t] = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
            }
        }
        if (hipMemcpyToSymbol(HIP_SYMBOL(c_msize), msize, 16) != hipSuccess ||
            hipMemcpyToSymbol(HIP_SYMBOL(c_member), member, 64) != hipSuccess) { delete c; return MP_ERR_DEVICE; }
    }
    *out = c;
    return MP_OK;
}

void mp_destroy(mp_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->dev);
    (void)hipDeviceSynchronize();
    free_msa(c);
    dev_free(c, &c->tmp_out, (size_t)c->tmp_out_n);
    for (auto &p : c->ev_busy) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &p : c->ev_free) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    delete c;
}

int mp_set_stream(mp_ctx *c, void *s) {
    if (!c) return MP_ERR_ARG;
    c->stream = (hipStream_t)s;
    return MP_OK;
}

int mp_device_bytes(mp_ctx *c, int64_t *b) {
    if (!c || !b) return MP_ERR_ARG;
    *b = c->bytes;
    return MP_OK;
}

// ---------------------------------------------------------------------------------------------
int mp_load_msa(mp_ctx *c, const uint8_t *bytes, const int64_t *row_off, int32_t n_rows) {
    if (!c) return MP_ERR_ARG;
    if (!bytes || !row_off || n_rows <= 0) return fail(c, MP_ERR_ARG, "mp_load_msa: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    free_msa(c);
    int64_t max_len = 0;
    for (int r = 0; r < n_rows; r++) {
        int64_t l = row_off[r + 1] - row_off[r];
        if (l < 0 || l > 0x3fffffff) return fail(c, MP_ERR_ARG, "row %d has bad length", r);
        max_len = std::max(max_len, l);
    }
    c->n_rows = n_rows;
    c->n_pad = (n_rows + kBlock - 1) / kBlock * kBlock;
    c->max_len = (int)max_len;
    c->n_chunks = (int)((max_len + 31) / 32) + 2;
    c->ustride = (int)(max_len / 8) + 2;
    size_t np = (size_t)c->n_pad;
    int64_t total = row_off[n_rows] - row_off[0];
    uint8_t *d_bytes = nullptr;
    int64_t *d_off = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &d_bytes, (size_t)total + 64))) return rc;
    if ((rc = dev_alloc(c, &d_off, (size_t)n_rows + 1))) return rc;
    if ((rc = dev_alloc(c, &c->planes, (size_t)c->n_chunks * 4 * np))) return rc;
    if ((rc = dev_alloc(c, &c->cols, (size_t)c->n_chunks * 32 * 3 * (np / 64)))) return rc;
    if ((rc = dev_alloc(c, &c->cum, ((size_t)c->n_chunks + 1) * np))) return rc;
    if ((rc = dev_alloc(c, &c->ung, (size_t)n_rows * c->ustride))) return rc;
    if ((rc = dev_alloc(c, &c->lead, np))) return rc;
    if ((rc = dev_alloc(c, &c->rstrip, np))) return rc;
    if ((rc = dev_alloc(c, &c->rlen, np))) return rc;
    std::vector<int64_t> off0(n_rows + 1);
    for (int r = 0; r <= n_rows; r++) off0[r] = row_off[r] - row_off[0];
    HIPCK(c, hipMemcpyAsync(d_bytes, bytes + row_off[0], (size_t)total, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_off, off0.data(), sizeof(int64_t) * (n_rows + 1), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemsetAsync(c->ung, 0, sizeof(uint32_t) * (size_t)n_rows * c->ustride, c->stream));
    HIPCK(c, hipMemsetAsync(c->rlen, 0, sizeof(int32_t) * np, c->stream));
    dim3 grid((unsigned)(c->n_pad / kBlock), (unsigned)c->n_chunks);
    hipLaunchKernelGGL(pack_kernel, grid, dim3(kBlock), 0, c->stream, d_bytes, d_off, n_rows, c->n_pad, c->n_chunks, c->planes);
    hipLaunchKernelGGL(row_scan_kernel, dim3(c->n_pad / kBlock), dim3(kBlock), 0, c->stream, c->planes, d_off, n_rows,
                       c->n_pad, c->n_chunks, c->cum, c->lead, c->rstrip, c->rlen);
    hipLaunchKernelGGL(ungap_kernel, grid, dim3(kBlock), 0, c->stream, c->planes, c->cum, n_rows, c->n_pad, c->ustride, c->ung);
    hipLaunchKernelGGL(colplane_kernel, grid, dim3(kBlock), 0, c->stream, c->planes, c->n_pad, c->n_chunks, c->cols);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipStreamSynchronize(c->stream));
    dev_free(c, &d_bytes, (size_t)total + 64);
    dev_free(c, &d_off, (size_t)n_rows + 1);
    return MP_OK;
}

int mp_row_attributes(mp_ctx *c, int32_t *lead, int32_t *rstrip, int32_t *rowlen) {
    if (!c) return MP_ERR_ARG;
    if (!c->planes) return fail(c, MP_ERR_ARG, "no alignment loaded");
    HIPCK(c, hipSetDevice(c->dev));
    size_t n = sizeof(int32_t) * (size_t)c->n_rows;
    if (lead) HIPCK(c, hipMemcpyAsync(lead, c->lead, n, hipMemcpyDeviceToHost, c->stream));
    if (rstrip) HIPCK(c, hipMemcpyAsync(rstrip, c->rstrip, n, hipMemcpyDeviceToHost, c->stream));
    if (rowlen) HIPCK(c, hipMemcpyAsync(rowlen, c->rlen, n, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

// ---------------------------------------------------------------------------------------------
int mp_build_windows(mp_ctx *c, int32_t p0, int32_t n_win, int32_t k, int32_t v, int32_t *n_exc) {
    if (!c) return MP_ERR_ARG;
    if (!c->planes) return fail(c, MP_ERR_ARG, "no alignment loaded");
    if (k < 2 || k > MP_MAX_K || n_win <= 0 || p0 < 0 || v < 0 || v >= k)
        return fail(c, MP_ERR_ARG, "bad window arguments (k=%d v=%d n_windows=%d)", k, v, n_win);
    if (p0 + n_win > c->max_len) return fail(c, MP_ERR_ARG, "windows run past the longest row");
    HIPCK(c, hipSetDevice(c->dev));
    free_windows(c);
    c->p0 = p0; c->n_win = n_win; c->k = k; c->v = v;
    size_t np = (size_t)c->n_pad;
    int rc;
    // one packed u64 per (window, sequence) when 3k bits + the flag fit, else three u32 planes
    c->p64 = 3 * k <= 63 && !getenv("MP_WIN_NO_PACK");
    {
        uint8_t *wp = nullptr;
        size_t nb = (size_t)n_win * np * (c->p64 ? 8 : 12);
        if ((rc = dev_alloc(c, &wp, nb))) return rc;
        c->win = wp;
        c->win_bytes = nb;
    }
    if ((rc = dev_alloc(c, &c->ex_count, 1))) return rc;
    if ((rc = dev_alloc(c, &c->err_flag, 4))) return rc;
    if ((rc = dev_alloc(c, &c->extra_off, (size_t)n_win + 1))) return rc;
    HIPCK(c, hipMemsetAsync(c->extra_off, 0, sizeof(int32_t) * ((size_t)n_win + 1), c->stream));
    const size_t nw = np / 64;
    if ((rc = dev_alloc(c, &c->excl, (size_t)n_win * nw))) return rc;
    if ((rc = dev_alloc(c, &c->patch_count, (size_t)n_win))) return rc;
    if ((rc = dev_alloc(c, &c->patch_off, (size_t)n_win + 1))) return rc;
    if ((rc = dev_alloc(c, &c->patch_cursor, (size_t)n_win))) return rc;
    int cap = 1 << 16;
    const int tile = 64;
    const dim3 grid((unsigned)(c->n_pad / kBlock), (unsigned)((n_win + tile - 1) / tile));
    auto launch = [&](const PatchOut &po, int ex_cap) {
        if (c->p64)
            hipLaunchKernelGGL(build_windows_kernel<true>, grid, dim3(kBlock), 0, c->stream, c->planes, c->cum, c->ung, c->rlen,
                               c->n_rows, c->n_pad, c->n_chunks, c->ustride, p0, n_win, tile, k, c->win, c->ex, ex_cap,
                               c->ex_count, c->err_flag, po);
        else
            hipLaunchKernelGGL(build_windows_kernel<false>, grid, dim3(kBlock), 0, c->stream, c->planes, c->cum, c->ung, c->rlen,
                               c->n_rows, c->n_pad, c->n_chunks, c->ustride, p0, n_win, tile, k, c->win, c->ex, ex_cap,
                               c->ex_count, c->err_flag, po);
    };
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = dev_alloc(c, &c->ex, (size_t)cap))) return rc;
        c->ex_cap = cap;
        HIPCK(c, hipMemsetAsync(c->ex_count, 0, sizeof(int), c->stream));
        HIPCK(c, hipMemsetAsync(c->err_flag, 0, 4 * sizeof(int), c->stream));
        HIPCK(c, hipMemsetAsync(c->excl, 0, sizeof(unsigned long long) * (size_t)n_win * nw, c->stream));
        HIPCK(c, hipMemsetAsync(c->patch_count, 0, sizeof(int32_t) * (size_t)n_win, c->stream));
        launch(PatchOut{0, c->excl, c->patch_count, nullptr, nullptr, nullptr}, cap);
        HIPCK(c, hipGetLastError());
        int cnt = 0, errv[4] = {0, 0, 0, 0};
        HIPCK(c, hipMemcpyAsync(&cnt, c->ex_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipMemcpyAsync(errv, c->err_flag, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
        if (errv[0])
            return fail(c, MP_ERR_SHORT_WINDOW, "row %d has fewer than %d residues at window %d", errv[2], k, p0 + errv[1]);
        if (cnt <= cap) {
            c->ex_host.resize((size_t)cnt);
            if (cnt) HIPCK(c, hipMemcpy(c->ex_host.data(), c->ex, sizeof(ExRec) * (size_t)cnt, hipMemcpyDeviceToHost));
            std::sort(c->ex_host.begin(), c->ex_host.end(),
                      [](const ExRec &a, const ExRec &b) { return a.win != b.win ? a.win < b.win : a.row < b.row; });
            if (n_exc) *n_exc = cnt;
            // patch list: counts -> offsets on the host, then the fill pass
            std::vector<int32_t> pc((size_t)n_win), po((size_t)n_win + 1, 0);
            HIPCK(c, hipMemcpy(pc.data(), c->patch_count, sizeof(int32_t) * (size_t)n_win, hipMemcpyDeviceToHost));
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
            HIPCK(c, hipMemcpy(c->patch_off, po.data(), sizeof(int32_t) * po.size(), hipMemcpyHostToDevice));
            if (tot) {
                if ((rc = dev_alloc(c, &c->patch_words, (size_t)3 * (size_t)tot))) return rc;
                HIPCK(c, hipMemsetAsync(c->patch_cursor, 0, sizeof(int32_t) * (size_t)n_win, c->stream));
                launch(PatchOut{1, c->excl, c->patch_count, c->patch_off, c->patch_cursor, c->patch_words}, 0);
                HIPCK(c, hipGetLastError());
                HIPCK(c, hipStreamSynchronize(c->stream));
            }
            return MP_OK;
        }
        dev_free(c, &c->ex, (size_t)cap);
        cap = cnt;
    }
    return fail(c, MP_ERR_DEVICE, "exception list did not converge");
}

int mp_get_exceptions(mp_ctx *c, int32_t cap, int32_t *ew, int32_t *er, uint8_t *codes) {
    if (!c) return MP_ERR_ARG;
    if (!c->win) return fail(c, MP_ERR_ARG, "no windows built");
    int n = (int)c->ex_host.size();
    if (cap < n) return fail(c, MP_ERR_CAPACITY, "exception buffer too small: need %d", n);
    for (int i = 0; i < n; i++) {
        const ExRec &e = c->ex_host[(size_t)i];
        ew[i] = e.win; er[i] = e.row;
        for (int j = 0; j < c->k; j++)
            codes[(size_t)i * c->k + j] = (uint8_t)((j < 16 ? e.lo >> (4 * j) : e.hi >> (4 * (j - 16))) & 15u);
    }
    return MP_OK;
}

int mp_set_extra_rows(mp_ctx *c, int32_t n, const int32_t *win, const uint32_t *words) {
    if (!c) return MP_ERR_ARG;
    if (!c->win) return fail(c, MP_ERR_ARG, "no windows built");
    HIPCK(c, hipSetDevice(c->dev));
    dev_free(c, &c->extra_words, (size_t)3 * c->n_extra);
    c->n_extra = 0;
    std::vector<int32_t> off((size_t)c->n_win + 1, 0);
    for (int i = 0; i < n; i++) {
        if (win[i] < 0 || win[i] >= c->n_win || (i && win[i] < win[i - 1]))
            return fail(c, MP_ERR_ARG, "extra rows must be sorted by window and in range");
        off[(size_t)win[i] + 1]++;
    }
    for (int w = 0; w < c->n_win; w++) off[(size_t)w + 1] += off[(size_t)w];
    HIPCK(c, hipMemcpy(c->extra_off, off.data(), sizeof(int32_t) * off.size(), hipMemcpyHostToDevice));
    if (n > 0) {
        int rc;
        if ((rc = dev_alloc(c, &c->extra_words, (size_t)3 * n))) return rc;
        c->n_extra = n;
        HIPCK(c, hipMemcpy(c->extra_words, words, sizeof(uint32_t) * 3 * (size_t)n, hipMemcpyHostToDevice));
    }
    return MP_OK;
}

int mp_get_window_words(mp_ctx *c, int32_t w, int32_t row0, int32_t n, uint32_t *out) {
    if (!c) return MP_ERR_ARG;
    if (!c->win) return fail(c, MP_ERR_ARG, "no windows built");
    if (w < 0 || w >= c->n_win || row0 < 0 || n < 0 || row0 + n > c->n_rows) return fail(c, MP_ERR_ARG, "bad range");
    HIPCK(c, hipSetDevice(c->dev));
    size_t np = (size_t)c->n_pad;
    if (c->p64) {
        std::vector<uint64_t> tmp((size_t)n + 1);
        HIPCK(c, hipMemcpyAsync(tmp.data(), (const uint64_t *)c->win + (size_t)w * np + row0, sizeof(uint64_t) * (size_t)n,
                                hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
        const uint32_t kmask = (1u << c->k) - 1u;
        for (int i = 0; i < n; i++) {
            uint64_t x = tmp[(size_t)i];
            out[i] = (uint32_t)x & kmask;
            out[(size_t)n + i] = (uint32_t)(x >> c->k) & kmask;
            out[2 * (size_t)n + i] = ((uint32_t)(x >> (2 * c->k)) & kmask) | ((uint32_t)(x >> 32) & MP_WIN_SKIP);
        }
        return MP_OK;
    }
    const uint32_t *W = (const uint32_t *)c->win;
    for (int p = 0; p < 3; p++)
        HIPCK(c, hipMemcpyAsync(out + (size_t)p * n, W + ((size_t)w * 3 + p) * np + row0, sizeof(uint32_t) * (size_t)n,
                                hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

// ---------------------------------------------------------------------------------------------
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

// ---------------------------------------------------------------------------------------------
int mp_eval_upload(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint32_t sF, uint32_t sR) {
    if (!c) return MP_ERR_ARG;
    if (!c->win) return fail(c, MP_ERR_ARG, "no windows built");
    if (n_cand < 0 || (n_cand && (!cw || !codes))) return fail(c, MP_ERR_ARG, "bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    free_eval(c);
    const int k = c->k;
    const uint32_t kmask = (1u << k) - 1u;
    std::vector<EvalItem> items;
    std::vector<uint4> cn;
    std::vector<int32_t> co;
    std::vector<uint32_t> symT;
    int i = 0;
    while (i < n_cand) {
        int w = cw[i];
        if (w < 0 || w >= c->n_win || (i && w < cw[i - 1])) return fail(c, MP_ERR_ARG, "candidate windows must be ascending and in range");
        int j = i;
        while (j < n_cand && cw[j] == w) j++;
        for (int b = i; b < j; b += kEvalCC) {
            items.push_back(EvalItem{w, (int32_t)cn.size()});
            symT.resize(items.size() * 32, 0u);
            for (int t = 0; t < kEvalCC; t++) {
                int ci = b + t;
                if (ci < j) {
                    uint32_t nA = 0, nC = 0, nG = 0, nT = 0;
                    for (int p = 0; p < k; p++) {
                        uint8_t m = codes[(size_t)ci * k + p];
                        if (!(m & 1)) nA |= 1u << p;
                        if (!(m & 2)) nC |= 1u << p;
                        if (!(m & 4)) nG |= 1u << p;
                        if (!(m & 8)) nT |= 1u << p;
                    }
                    cn.push_back(uint4{nA, nC, nG, nT});
                    co.push_back(ci);
                    for (int p = 0; p < k; p++)
                        symT[(items.size() - 1) * 32 + (size_t)p] |= (uint32_t)(codes[(size_t)ci * k + p] & 15u) << (4 * t);
                } else {
                    cn.push_back(uint4{kmask, kmask, kmask, kmask});
                    co.push_back(-1);
                }
            }
        }
        i = j;
    }
    c->n_cand = n_cand; c->sF = sF; c->sR = sR;
    c->n_items = (int)items.size();
    c->n_padded = (int)cn.size();
    if (c->n_items == 0) return MP_OK;
    int rc;
    if ((rc = dev_alloc(c, &c->items, items.size()))) return rc;
    if ((rc = dev_alloc(c, &c->cand_n, cn.size()))) return rc;
    if ((rc = dev_alloc(c, &c->cand_out, co.size()))) return rc;
    if ((rc = dev_alloc(c, &c->cand_symT, symT.size()))) return rc;
    HIPCK(c, hipMemcpy(c->cand_symT, symT.data(), sizeof(uint32_t) * symT.size(), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(c->items, items.data(), sizeof(EvalItem) * items.size(), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(c->cand_n, cn.data(), sizeof(uint4) * cn.size(), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(c->cand_out, co.data(), sizeof(int32_t) * co.size(), hipMemcpyHostToDevice));
    return MP_OK;
}

int mp_eval_launch(mp_ctx *c, int64_t *device_out) {
    if (!c) return MP_ERR_ARG;
    if (!c->win) return fail(c, MP_ERR_ARG, "no windows built");
    if (!device_out) return fail(c, MP_ERR_ARG, "null output");
    HIPCK(c, hipSetDevice(c->dev));
    if (c->n_cand == 0) return MP_OK;
    HIPCK(c, hipMemsetAsync(device_out, 0, sizeof(int64_t) * 3 * (size_t)c->n_cand, c->stream));
    // enough blocks to fill 256 CUs several times over, each with at least 1024 sequences
    int max_split = (c->n_pad + 1023) / 1024;
    int want = (4096 + c->n_items - 1) / c->n_items;
    int split = std::max(1, std::min(max_split, want));
    int rows = ((c->n_pad + split - 1) / split + 1023) / 1024 * 1024;
    split = (c->n_pad + rows - 1) / rows;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!c->ev_free.empty()) { ev = c->ev_free.back(); c->ev_free.pop_back(); }
    else { HIPCK(c, hipEventCreate(&ev.first)); HIPCK(c, hipEventCreate(&ev.second)); }
    HIPCK(c, hipEventRecord(ev.first, c->stream));
    const char *mode_env = getenv("MP_EVAL_MODE");
    const bool bits = c->v <= 2 && !(mode_env && !strcmp(mode_env, "rows"));
    const int vmode = c->v == 0 ? 0 : (c->v == 1 ? 1 : 2);     // predicate specialisation of the row-per-lane code
    if (bits) {
        // bit-sliced pass over the column planes + row-per-lane pass over the patch / IUPAC lists
        // kernel shape (MP_EVAL_BITS): 0 = 8 candidates x 2 words (64 sequences) per thread (default),
        // 1 = the same with the next position's planes prefetched, 2 = 8 x 1 word
        int shape = 0;
        if (const char *e = getenv("MP_EVAL_BITS")) { shape = atoi(e); if (shape < 0 || shape > 2) shape = 0; }
        static const int shape_gw[3] = {2, 2, 1};
        const int nw = c->n_pad / 64;
        const int GW = shape_gw[shape];
        const int ny = std::max(1, (2 * nw / GW + kBlock - 1) / kBlock);
        const int ny_pad = (ny + 7) / 8 * 8;
        EvalBitsArgs ba{c->cols, c->excl, nw, c->p0, c->k, c->v, c->items, c->cand_symT, c->cand_out, c->sF, c->sR,
                        (unsigned long long *)device_out, ny, ny_pad};
#define BITS_ROW(LV) {eval_bits_kernel<8, LV, 2, false>, eval_bits_kernel<8, LV, 2, true>, eval_bits_kernel<8, LV, 1, false>}
        static const EvalBitsFn bfn[3][3] = {BITS_ROW(1), BITS_ROW(2), BITS_ROW(3)};
#undef BITS_ROW
        hipLaunchKernelGGL(bfn[c->v][shape], dim3((unsigned)((size_t)c->n_items * ny_pad)), dim3(kBlock), 0, c->stream, ba);
        if (c->n_patch || c->n_extra) {
            EvalListArgs la{c->items, c->cand_n, c->cand_out, c->n_patch ? c->patch_off : (const int32_t *)nullptr, c->patch_words,
                            c->n_extra ? c->extra_off : (const int32_t *)nullptr, c->extra_words, c->sF, c->sR, c->v,
                            (1u << c->k) - 1u, (unsigned long long *)device_out};
            static const EvalListFn lfn[3] = {eval_list_kernel<kEvalCC, 0>, eval_list_kernel<kEvalCC, 1>, eval_list_kernel<kEvalCC, 2>};
            int ly = std::max(1, std::min(64, (c->max_patch + 2047) / 2048));
            hipLaunchKernelGGL(lfn[vmode], dim3((unsigned)c->n_items, (unsigned)ly), dim3(kBlock), 0, c->stream, la);
        }
    } else {
    EvalArgs ea{c->win, c->n_pad, c->k, c->items, c->cand_n, c->cand_out, c->n_extra ? c->extra_off : (const int32_t *)nullptr,
                c->extra_words, c->sF, c->sR, c->v, (1u << c->k) - 1u, rows, (unsigned long long *)device_out};
    int variant = c->eval_variant;
    if (const char *e = getenv("MP_EVAL_VARIANT")) variant = atoi(e);
    if (variant < 0 || variant >= kNumEvalVariants) variant = 0;
    hipLaunchKernelGGL(kEvalVariants[variant].fn[c->p64 ? 1 : 0][getenv("MP_EVAL_GENERIC_V") ? 2 : vmode], dim3((unsigned)c->n_items, (unsigned)split),
                       dim3(kBlock), 0, c->stream, ea);
    }
    HIPCK(c, hipEventRecord(ev.second, c->stream));
    c->ev_busy.push_back(ev);
    HIPCK(c, hipGetLastError());
    return MP_OK;
}

int mp_eval_timing(mp_ctx *c, int32_t reset, double *total_ms, int32_t *n_launches) {
    if (!c) return MP_ERR_ARG;
    HIPCK(c, hipSetDevice(c->dev));
    for (auto &p : c->ev_busy) {
        HIPCK(c, hipEventSynchronize(p.second));
        float ms = 0;
        HIPCK(c, hipEventElapsedTime(&ms, p.first, p.second));
        c->ev_ms += ms;
        c->ev_n++;
        c->ev_free.push_back(p);
    }
    c->ev_busy.clear();
    if (total_ms) *total_ms = c->ev_ms;
    if (n_launches) *n_launches = c->ev_n;
    if (reset) { c->ev_ms = 0; c->ev_n = 0; }
    return MP_OK;
}

int mp_eval_candidates(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint32_t sF, uint32_t sR,
                       int64_t *out) {
    if (!c) return MP_ERR_ARG;
    int rc = mp_eval_upload(c, n_cand, cw, codes, sF, sR);
    if (rc) return rc;
    if (n_cand == 0) return MP_OK;
    if (!out) return fail(c, MP_ERR_ARG, "null output");
    if (c->tmp_out_n < 3 * n_cand) {
        dev_free(c, &c->tmp_out, (size_t)c->tmp_out_n);
        c->tmp_out_n = 0;
        if ((rc = dev_alloc(c, &c->tmp_out, (size_t)3 * n_cand))) return rc;
        c->tmp_out_n = 3 * n_cand;
    }
    if ((rc = mp_eval_launch(c, (int64_t *)c->tmp_out))) return rc;
    HIPCK(c, hipMemcpyAsync(out, c->tmp_out, sizeof(int64_t) * 3 * (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

// ---------------------------------------------------------------------------------------------
int mp_dimer_scan(mp_ctx *c, int32_t n, const uint8_t *codes, const int32_t *off, int32_t mode, int32_t n_new,
                  const uint8_t *loss_hit, const double *dg, double dg_limit, int64_t cap, int32_t *hits, int64_t *n_hits) {
    if (!c) return MP_ERR_ARG;
    if (n < 0 || !codes || !off || !loss_hit || !dg || !n_hits || cap < 0 || (cap && !hits) || (mode != 0 && mode != 1))
        return fail(c, MP_ERR_ARG, "mp_dimer_scan: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    *n_hits = 0;
    if (n == 0) return MP_OK;
    static const int msize[16] = {0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4};
    for (int32_t i = 0; i < n; i++) {
        int len = off[i + 1] - off[i];
        if (len < 1 || len > MP_DIMER_MAX_LEN) return fail(c, MP_ERR_ARG, "primer %d has length %d (1..%d supported)", i, len, MP_DIMER_MAX_LEN);
        long long d = 1;
        for (int p = 0; p < len; p++) {
            uint8_t m = codes[off[i] + p];
            if (m == 0 || m > 15) return fail(c, MP_ERR_ARG, "primer %d holds a gap / unknown symbol", i);
            d *= msize[m];
            if (d > (1LL << 24)) return fail(c, MP_ERR_ARG, "primer %d has too many expansions", i);
        }
    }
    const size_t total = (size_t)off[n], tbl = (size_t)(MP_DIMER_MAX_LEN + 1) * (MP_DIMER_MAX_LEN + 1) * 64;
    const size_t ndg = 16 + 32 + MP_DIMER_MAX_LEN + 1 + 1;
    uint8_t *d_codes = nullptr, *d_loss = nullptr;
    int32_t *d_off = nullptr, *d_hits = nullptr;
    double *d_dg = nullptr;
    unsigned long long *d_n = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &d_codes, total))) return rc;
    if ((rc = dev_alloc(c, &d_off, (size_t)n + 1))) return rc;
    if ((rc = dev_alloc(c, &d_loss, tbl))) return rc;
    if ((rc = dev_alloc(c, &d_dg, ndg))) return rc;
    if ((rc = dev_alloc(c, &d_hits, (size_t)cap * 6))) return rc;
    if ((rc = dev_alloc(c, &d_n, 1))) return rc;
    HIPCK(c, hipMemcpyAsync(d_codes, codes, total, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_off, off, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_loss, loss_hit, tbl, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_dg, dg, sizeof(double) * ndg, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemsetAsync(d_n, 0, sizeof(unsigned long long), c->stream));
    DimerArgs da{d_codes, d_off, n, mode, n_new, d_loss, d_dg, dg_limit, (long long)cap, d_hits, d_n};
    hipLaunchKernelGGL(dimer_kernel, dim3((unsigned)n, (unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, da);
    HIPCK(c, hipGetLastError());
    unsigned long long nh = 0;
    HIPCK(c, hipMemcpyAsync(&nh, d_n, sizeof(nh), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    size_t ncopy = (size_t)std::min<unsigned long long>(nh, (unsigned long long)cap);
    if (ncopy) HIPCK(c, hipMemcpy(hits, d_hits, sizeof(int32_t) * 6 * ncopy, hipMemcpyDeviceToHost));
    *n_hits = (int64_t)nh;
    dev_free(c, &d_codes, total); dev_free(c, &d_off, (size_t)n + 1); dev_free(c, &d_loss, tbl);
    dev_free(c, &d_dg, ndg); dev_free(c, &d_hits, (size_t)cap * 6); dev_free(c, &d_n, 1);
    return MP_OK;
}

static int check_primers(mp_ctx *c, int32_t n, const uint8_t *codes, const int32_t *off) {
    static const int msize[16] = {0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4};
    for (int32_t i = 0; i < n; i++) {
        int len = off[i + 1] - off[i];
        if (len < 1 || len > MP_DIMER_MAX_LEN) return fail(c, MP_ERR_ARG, "primer %d has length %d (1..%d supported)", i, len, MP_DIMER_MAX_LEN);
        long long d = 1;
        for (int p = 0; p < len; p++) {
            uint8_t m = codes[off[i] + p];
            if (m == 0 || m > 15) return fail(c, MP_ERR_ARG, "primer %d holds a gap / unknown symbol", i);
            d *= msize[m];
            if (d > (1LL << 24)) return fail(c, MP_ERR_ARG, "primer %d has too many expansions", i);
        }
    }
    return MP_OK;
}

int mp_dimer_pairs(mp_ctx *c, int32_t n, const uint8_t *codes, const int32_t *off, int64_t n_pairs, const int32_t *pairs,
                   const uint8_t *loss_hit, const double *dg, double dg_limit, uint8_t *flags) {
    if (!c) return MP_ERR_ARG;
    if (n < 0 || !codes || !off || !loss_hit || !dg || n_pairs < 0 || (n_pairs && (!pairs || !flags)))
        return fail(c, MP_ERR_ARG, "mp_dimer_pairs: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    if (n_pairs == 0) return MP_OK;
    int rc;
    if ((rc = check_primers(c, n, codes, off))) return rc;
    for (int64_t p = 0; p < 2 * n_pairs; p++)
        if (pairs[p] < 0 || pairs[p] >= n) return fail(c, MP_ERR_ARG, "pair %lld out of range", (long long)(p / 2));
    const size_t total = (size_t)off[n], tbl = (size_t)(MP_DIMER_MAX_LEN + 1) * (MP_DIMER_MAX_LEN + 1) * 64;
    const size_t ndg = 16 + 32 + MP_DIMER_MAX_LEN + 1 + 1;
    uint8_t *d_codes = nullptr, *d_loss = nullptr, *d_flags = nullptr;
    int32_t *d_off = nullptr, *d_pairs = nullptr;
    double *d_dg = nullptr;
    if ((rc = dev_alloc(c, &d_codes, total))) return rc;
    if ((rc = dev_alloc(c, &d_off, (size_t)n + 1))) return rc;
    if ((rc = dev_alloc(c, &d_loss, tbl))) return rc;
    if ((rc = dev_alloc(c, &d_dg, ndg))) return rc;
    if ((rc = dev_alloc(c, &d_pairs, (size_t)2 * n_pairs))) return rc;
    if ((rc = dev_alloc(c, &d_flags, (size_t)n_pairs))) return rc;
    HIPCK(c, hipMemcpyAsync(d_codes, codes, total, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_off, off, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_loss, loss_hit, tbl, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_dg, dg, sizeof(double) * ndg, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_pairs, pairs, sizeof(int32_t) * 2 * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(dimer_pairs_kernel, dim3((unsigned)((n_pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, d_codes,
                       d_off, (long long)n_pairs, d_pairs, d_loss, d_dg, dg_limit, d_flags);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(flags, d_flags, (size_t)n_pairs, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    dev_free(c, &d_codes, total); dev_free(c, &d_off, (size_t)n + 1); dev_free(c, &d_loss, tbl); dev_free(c, &d_dg, ndg);
    dev_free(c, &d_pairs, (size_t)2 * n_pairs); dev_free(c, &d_flags, (size_t)n_pairs);
    return MP_OK;
}

int mp_pair_coverage(mp_ctx *c, int32_t n_sets, int32_t n_words, const uint64_t *a, const uint64_t *b, int64_t n_pairs,
                     const int32_t *pairs, int32_t *out) {
    if (!c) return MP_ERR_ARG;
    if (n_sets < 0 || n_words < 0 || n_pairs < 0 || (n_pairs && (!a || !b || !pairs || !out)))
        return fail(c, MP_ERR_ARG, "mp_pair_coverage: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    if (n_pairs == 0) return MP_OK;
    for (int64_t p = 0; p < 2 * n_pairs; p++)
        if (pairs[p] < 0 || pairs[p] >= n_sets) return fail(c, MP_ERR_ARG, "pair %lld out of range", (long long)(p / 2));
    const size_t nset = (size_t)n_sets * (size_t)n_words;
    unsigned long long *d_a = nullptr, *d_b = nullptr;
    int32_t *d_pairs = nullptr, *d_out = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &d_a, nset))) return rc;
    if ((rc = dev_alloc(c, &d_b, nset))) return rc;
    if ((rc = dev_alloc(c, &d_pairs, (size_t)2 * n_pairs))) return rc;
    if ((rc = dev_alloc(c, &d_out, (size_t)n_pairs))) return rc;
    HIPCK(c, hipMemcpyAsync(d_a, a, sizeof(uint64_t) * nset, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_b, b, sizeof(uint64_t) * nset, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_pairs, pairs, sizeof(int32_t) * 2 * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    const long long per_block = kBlock / 64;
    hipLaunchKernelGGL(pair_coverage_kernel, dim3((unsigned)((n_pairs + per_block - 1) / per_block)), dim3(kBlock), 0, c->stream,
                       d_a, d_b, n_words, (long long)n_pairs, d_pairs, d_out);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, d_out, sizeof(int32_t) * (size_t)n_pairs, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    dev_free(c, &d_a, nset); dev_free(c, &d_b, nset); dev_free(c, &d_pairs, (size_t)2 * n_pairs); dev_free(c, &d_out, (size_t)n_pairs);
    return MP_OK;
}

}  // extern "C"
