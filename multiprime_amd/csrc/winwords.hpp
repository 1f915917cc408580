// winwords.hpp — the k-mer of one (window, sequence) pair with edge-gap repair (get_primers, V20:666-687), derived from
// the bit planes on the fly.  Shared by the window scan (windows.hip), the histograms (unique.hip) and the row-per-lane
// evaluation / mask kernels (eval.hip): nothing stores window words any more.
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

__device__ inline uint32_t ung_get(const MsaArgs &M, int r, uint32_t t) {
    return (M.ung[(size_t)(t >> 3) * (size_t)M.n_pad + r] >> ((t & 7) * 4)) & 15u;
}

// The general path: rows whose window starts or ends in a gap, holds an IUPAC code, or runs past
// the end of a ragged row.  Follows get_primers line by line.  Returns 0 = store words,
// 1 = exception (IUPAC code present, `buf` returned), 2 = fewer than k residues (V20:683-687).
__device__ inline int repair_window(const MsaArgs &M, int r, uint32_t wA, uint32_t wC, uint32_t wG, uint32_t wT, int k, int p, int len,
                                    uint32_t c_left, uint32_t total, uint32_t &b0, uint32_t &b1, uint32_t &g, Nib &buf) {
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
                for (int t = 0; t < run; t++) buf.set(t, ung_get(M, r, c_left - run + t));
        }
        if (buf.get(n - 1) == 0) {                            // V20:677 sequence.endswith("-")
            int run = 0;
            while (run < n && buf.get(n - 1 - run) == 0) run++;
            uint32_t c_after = c_left + __popc(ng);          // residues in s[0 : p+k]
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

// The plain column slice: window words of row r at absolute column p from the eight plane words that cover it (chunk
// c = p >> 5 and c + 1).  Returns true when the k-mer IS that slice (inside the row, no IUPAC code, no gap at either
// edge — or all gaps): (b0,b1,g) are then final.  false: the row needs slow_words().
__device__ inline bool fast_words(int p, int k, uint32_t kmask, int len, uint32_t loA, uint32_t loC, uint32_t loG, uint32_t loT,
                                  uint32_t hiA, uint32_t hiC, uint32_t hiG, uint32_t hiT, uint32_t &b0, uint32_t &b1, uint32_t &g) {
    const int o = p & 31;
    const uint32_t wA = __funnelshift_r(loA, hiA, o) & kmask;
    const uint32_t wC = __funnelshift_r(loC, hiC, o) & kmask;
    const uint32_t wG = __funnelshift_r(loG, hiG, o) & kmask;
    const uint32_t wT = __funnelshift_r(loT, hiT, o) & kmask;
    const uint32_t o1 = wA | wC, a1 = wA & wC, o2 = wG | wT, a2 = wG & wT;
    const uint32_t ng = o1 | o2;
    const uint32_t multi = a1 | a2 | (o1 & o2);
    const uint32_t gw = ~ng & kmask;
    b0 = wC | wT; b1 = wG | wT; g = gw;
    return (p + k <= len) && multi == 0 && (gw == kmask || ((gw & 1u) == 0 && (gw >> (k - 1)) == 0));
}

// The general path for a row fast_words() turned down (edge-gap repair, IUPAC code, ragged end).
// rc 0: (b0,b1,g) valid; 1: IUPAC exception (`buf` holds the symbol codes); 2: fewer than k residues.
__device__ inline int slow_words(const MsaArgs &M, int r, int p, int k, uint32_t kmask, int len, uint32_t loA, uint32_t loC,
                                 uint32_t loG, uint32_t loT, uint32_t hiA, uint32_t hiC, uint32_t hiG, uint32_t hiT,
                                 uint32_t &b0, uint32_t &b1, uint32_t &g, Nib &buf) {
    const int c = p >> 5, o = p & 31;
    const uint32_t wA = __funnelshift_r(loA, hiA, o) & kmask;
    const uint32_t wC = __funnelshift_r(loC, hiC, o) & kmask;
    const uint32_t wG = __funnelshift_r(loG, hiG, o) & kmask;
    const uint32_t wT = __funnelshift_r(loT, hiT, o) & kmask;
    const size_t np = (size_t)M.n_pad;
    const uint32_t ng_lo = loA | loC | loG | loT;
    const uint32_t c_left = M.cum[(size_t)c * np + r] + __popc(ng_lo & ((1u << o) - 1u));
    const uint32_t total = M.cum[(size_t)M.n_chunks * np + r];
    return repair_window(M, r, wA, wC, wG, wT, k, p, len, c_left, total, b0, b1, g, buf);
}

__device__ inline int words_from_planes(const MsaArgs &M, int r, int p, int k, uint32_t kmask, int len, uint32_t loA, uint32_t loC,
                                        uint32_t loG, uint32_t loT, uint32_t hiA, uint32_t hiC, uint32_t hiG, uint32_t hiT,
                                        uint32_t &b0, uint32_t &b1, uint32_t &g, bool &fast, Nib &buf) {
    fast = fast_words(p, k, kmask, len, loA, loC, loG, loT, hiA, hiC, hiG, hiT, b0, b1, g);
    if (fast) return 0;
    return slow_words(M, r, p, k, kmask, len, loA, loC, loG, loT, hiA, hiC, hiG, hiT, b0, b1, g, buf);
}

// Window words of (window at absolute column p, row r), derived on the fly; rows past n_rows, IUPAC windows and
// too-short rows come back as MP_WIN_SKIP slots (not part of any count), like the stored words of round 1 did.
struct FlyView {
    MsaArgs M;
    int p, k;
    uint32_t kmask;
    __device__ FlyView(const MsaArgs &M_, int p_, int k_, uint32_t kmask_) : M(M_), p(p_), k(k_), kmask(kmask_) {}
    __device__ inline void load(int r, uint32_t &b0, uint32_t &b1, uint32_t &g) const {
        if (r >= M.n_rows) { b0 = b1 = 0; g = MP_WIN_SKIP | kmask; return; }
        const size_t np = (size_t)M.n_pad;
        const size_t base = ((size_t)(p >> 5) * 4) * np + r;
        const uint32_t *P = M.planes + base;
        bool fast;
        Nib buf;
        int rc = words_from_planes(M, r, p, k, kmask, M.rlen[r], P[0], P[np], P[2 * np], P[3 * np], P[4 * np], P[5 * np], P[6 * np],
                                   P[7 * np], b0, b1, g, fast, buf);
        if (rc) { b0 = b1 = 0; g = MP_WIN_SKIP | kmask; }
    }
};

}  // namespace mp
