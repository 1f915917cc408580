// winwords.hpp — the k-mer of one (window, sequence) pair with edge-gap repair (get_primers, V20:666-687), derived from
// the bit planes on the fly.  Shared by the window scan (windows.hip), the histograms (unique.hip) and the row-per-lane
// evaluation / mask kernels (eval.hip): nothing stores window words any more.
//
// Word type W: uint32_t while k <= MP_NARROW_K (a window lies inside two 32-column chunks), uint64_t for primers of 32..63 bases
// (three chunks).  The narrow instantiations are what every round-1..3 kernel used; the wide ones run the same statements on 64-bit
// words (round 4, -l up to 63).
#pragma once

#include "common.hpp"

namespace mp {

struct MsaArgs {
    const uint32_t *planes;       // [n_chunks][4][n_pad]
    const uint32_t *cum;          // [n_chunks+1][n_pad]
    const uint32_t *ung;          // [ustride][n_pad]
    const int32_t *rlen;          // [n_pad]
    int n_rows, n_pad, n_chunks;
};

inline MsaArgs msa_args(const mp_ctx *c) { return MsaArgs{c->planes, c->cum, c->ung, c->rlen, c->n_rows, c->n_pad, c->n_chunks}; }

template <typename W> struct WordTraits;
template <> struct WordTraits<uint32_t> {
    static constexpr int kNibWords = 2, kChunks = 2;
    static constexpr uint32_t kSkip = MP_WIN_SKIP;
};
template <> struct WordTraits<uint64_t> {
    static constexpr int kNibWords = 4, kChunks = 3;
    static constexpr uint64_t kSkip = MP_WIN_SKIP64;
};
template <typename W> using NibOf = NibT<WordTraits<W>::kNibWords>;

__device__ inline int popcw(uint32_t x) { return __popc(x); }
__device__ inline int popcw(uint64_t x) { return __popcll((unsigned long long)x); }
template <typename W> __host__ __device__ inline W kmask_of(int k) { return k >= (int)(8 * sizeof(W)) ? ~(W)0 : (((W)1 << k) - (W)1); }

// the plane words of one row that cover a window: chunk c = p >> 5 and the one (narrow) or two (wide) after it, bases A,C,G,T
template <typename W> struct PlaneWords { uint32_t w[WordTraits<W>::kChunks][4]; };

template <typename W>
__device__ inline PlaneWords<W> load_plane_words(const uint32_t *P /* planes + (chunk * 4) * np + row */, size_t np) {
    PlaneWords<W> q;
#pragma unroll
    for (int c = 0; c < WordTraits<W>::kChunks; c++)
#pragma unroll
        for (int b = 0; b < 4; b++) q.w[c][b] = P[(size_t)(c * 4 + b) * np];
    return q;
}

// k columns of base b from offset o of the first chunk
__device__ inline uint32_t slice_of(const PlaneWords<uint32_t> &q, int b, int o, uint32_t kmask) { return __funnelshift_r(q.w[0][b], q.w[1][b], o) & kmask; }
__device__ inline uint64_t slice_of(const PlaneWords<uint64_t> &q, int b, int o, uint64_t kmask) {
    const uint64_t lo = (uint64_t)q.w[0][b] | ((uint64_t)q.w[1][b] << 32);
    return (o ? (lo >> o) | ((uint64_t)q.w[2][b] << (64 - o)) : lo) & kmask;
}

__device__ inline uint32_t ung_get(const MsaArgs &M, int r, uint32_t t) {
    return (M.ung[(size_t)(t >> 3) * (size_t)M.n_pad + r] >> ((t & 7) * 4)) & 15u;
}

// The general path: rows whose window starts or ends in a gap, holds an IUPAC code, or runs past
// the end of a ragged row.  Follows get_primers line by line.  Returns 0 = store words,
// 1 = exception (IUPAC code present, `buf` returned), 2 = fewer than k residues (V20:683-687).
template <typename W>
__device__ inline int repair_window(const MsaArgs &M, int r, W wA, W wC, W wG, W wT, int k, int p, int len,
                                    uint32_t c_left, uint32_t total, W &b0, W &b1, W &g, NibOf<W> &buf) {
    const W kmask = kmask_of<W>(k);
    int m = len - p;
    m = m < 0 ? 0 : (m > k ? k : m);
    W ng = (wA | wC | wG | wT) & kmask;
    buf.clear();
    for (int j = 0; j < m; j++) {
        uint32_t code = (uint32_t)((wA >> j) & 1u) | ((uint32_t)((wC >> j) & 1u) << 1) | ((uint32_t)((wG >> j) & 1u) << 2) | ((uint32_t)((wT >> j) & 1u) << 3);
        buf.set(j, code);
    }
    int n = m;
    bool all_gap = (m == k) && ng == 0;                       // V20:668
    if (!all_gap && n > 0) {
        if (buf.get(0) == 0) {                                // V20:671 sequence.startswith("-")
            int run = 0;
            while (run < n && buf.get(run) == 0) run++;
            if (c_left >= (uint32_t)run)                      // V20:675
                for (int t = 0; t < run; t++) buf.set(t, ung_get(M, r, c_left - run + t));
        }
        if (buf.get(n - 1) == 0) {                            // V20:677 sequence.endswith("-")
            int run = 0;
            while (run < n && buf.get(n - 1 - run) == 0) run++;
            uint32_t c_after = c_left + (uint32_t)popcw(ng);  // residues in s[0 : p+k]
            if (total - c_after >= (uint32_t)run)             // V20:681
                for (int t = 0; t < run; t++) buf.set(n - run + t, ung_get(M, r, c_after + t));
        }
    }
    if (n < k) {                                              // V20:683
        int need = k - n;
        if (c_left < (uint32_t)need) return 2;
        buf.shift_up(need);
        for (int t = 0; t < need; t++) buf.set(t, ung_get(M, r, c_left - need + t));
        n = k;
    }
    b0 = b1 = g = 0;
    bool iupac = false;
    for (int j = 0; j < k; j++) {
        uint32_t code = buf.get(j);
        if (code == 0) g |= (W)1 << j;
        else if (code & (code - 1)) iupac = true;
        else {
            uint32_t bi = __ffs(code) - 1;
            b0 |= (W)(bi & 1u) << j;
            b1 |= (W)(bi >> 1) << j;
        }
    }
    return iupac ? 1 : 0;
}

// The plain column slice from the four base slices of the window: returns true when the k-mer IS that slice (inside the row, no
// IUPAC code, no gap at either edge — or all gaps): (b0,b1,g) are then final.  false: the row needs slow_words().
template <typename W>
__device__ inline bool fast_from_slices(int p, int k, W kmask, int len, W wA, W wC, W wG, W wT, W &b0, W &b1, W &g) {
    const W o1 = wA | wC, a1 = wA & wC, o2 = wG | wT, a2 = wG & wT;
    const W ng = o1 | o2;
    const W multi = a1 | a2 | (o1 & o2);
    const W gw = ~ng & kmask;
    b0 = wC | wT; b1 = wG | wT; g = gw;
    return (p + k <= len) && multi == 0 && (gw == kmask || ((gw & 1u) == 0 && (gw >> (k - 1)) == 0));
}

// The narrow form on the eight plane words that cover the window (chunk c = p >> 5 and c + 1), as the hot kernels hold them.
__device__ inline bool fast_words(int p, int k, uint32_t kmask, int len, uint32_t loA, uint32_t loC, uint32_t loG, uint32_t loT,
                                  uint32_t hiA, uint32_t hiC, uint32_t hiG, uint32_t hiT, uint32_t &b0, uint32_t &b1, uint32_t &g) {
    const int o = p & 31;
    const uint32_t wA = __funnelshift_r(loA, hiA, o) & kmask;
    const uint32_t wC = __funnelshift_r(loC, hiC, o) & kmask;
    const uint32_t wG = __funnelshift_r(loG, hiG, o) & kmask;
    const uint32_t wT = __funnelshift_r(loT, hiT, o) & kmask;
    return fast_from_slices<uint32_t>(p, k, kmask, len, wA, wC, wG, wT, b0, b1, g);
}

template <typename W>
__device__ inline bool fast_words(int p, int k, W kmask, int len, const PlaneWords<W> &q, W &b0, W &b1, W &g) {
    const int o = p & 31;
    return fast_from_slices<W>(p, k, kmask, len, slice_of(q, 0, o, kmask), slice_of(q, 1, o, kmask), slice_of(q, 2, o, kmask),
                               slice_of(q, 3, o, kmask), b0, b1, g);
}

// The general path for a row fast_words() turned down (edge-gap repair, IUPAC code, ragged end).
// rc 0: (b0,b1,g) valid; 1: IUPAC exception (`buf` holds the symbol codes); 2: fewer than k residues.
template <typename W>
__device__ inline int slow_words(const MsaArgs &M, int r, int p, int k, W kmask, int len, const PlaneWords<W> &q, W &b0, W &b1, W &g,
                                 NibOf<W> &buf) {
    const int c = p >> 5, o = p & 31;
    const size_t np = (size_t)M.n_pad;
    const uint32_t ng_lo = q.w[0][0] | q.w[0][1] | q.w[0][2] | q.w[0][3];
    const uint32_t c_left = M.cum[(size_t)c * np + r] + __popc(ng_lo & ((1u << o) - 1u));
    const uint32_t total = M.cum[(size_t)M.n_chunks * np + r];
    return repair_window<W>(M, r, slice_of(q, 0, o, kmask), slice_of(q, 1, o, kmask), slice_of(q, 2, o, kmask), slice_of(q, 3, o, kmask), k, p, len,
                            c_left, total, b0, b1, g, buf);
}

// Window words of (window at absolute column p, row r), derived on the fly; rows past n_rows, IUPAC windows and
// too-short rows come back as SKIP slots (not part of any count), like the stored words of round 1 did.
template <typename W>
struct FlyViewT {
    MsaArgs M;
    int p, k;
    W kmask;
    __device__ FlyViewT(const MsaArgs &M_, int p_, int k_, W kmask_) : M(M_), p(p_), k(k_), kmask(kmask_) {}
    __device__ inline void load(int r, W &b0, W &b1, W &g) const {
        if (r >= M.n_rows) { b0 = b1 = 0; g = WordTraits<W>::kSkip | kmask; return; }
        const size_t np = (size_t)M.n_pad;
        const PlaneWords<W> q = load_plane_words<W>(M.planes + ((size_t)(p >> 5) * 4) * np + r, np);
        const int len = M.rlen[r];
        if (fast_words<W>(p, k, kmask, len, q, b0, b1, g)) return;
        NibOf<W> buf;
        if (slow_words<W>(M, r, p, k, kmask, len, q, b0, b1, g, buf)) { b0 = b1 = 0; g = WordTraits<W>::kSkip | kmask; }
    }
};
typedef FlyViewT<uint32_t> FlyView;

}  // namespace mp
