// dimer.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of
// include/mprime.h.  3'-end dimer scans (finDimer, get_Maxprimerset, get_multiPrime) and pair coverage from sequence bitsets.
#include "common.hpp"

using namespace mp;

namespace {

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

template <typename Str>
__device__ inline double dm_delta_g(Str e, int l, const double *__restrict__ dg) {
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

// One (end length, end expansion, y expansion) combination of the ordered pair x -> y; on a hit *idx_out = where RC(end) starts.
__device__ inline bool dimer_combo(const Nib &cx, const Nib &cy, int lx, int ly, int l, uint32_t ei, uint32_t pi,
                                   const uint8_t *__restrict__ loss_hit, int l0, int gc_rows, const double *__restrict__ dg, double dg_limit,
                                   int &idx_out) {
    const uint64_t mask = l == 32 ? ~0ull : ((1ull << (2 * l)) - 1ull);
    const uint64_t e = dm_expand(cx, lx - l, l, ei);
    uint64_t rc = 0;                                                      // reverse complement: 3 - base, reversed
    for (int t = 0; t < l; t++) rc |= (uint64_t)(3u - ((uint32_t)(e >> (2 * (l - 1 - t))) & 3u)) << (2 * t);
    const uint64_t p = dm_expand(cy, 0, ly, pi);
    int idx = -1;
    for (int s0 = 0; s0 + l <= ly; s0++)
        if (((p >> (2 * s0)) & mask) == rc) { idx = s0; break; }          // str.find: first occurrence
    if (idx < 0) return false;
    const int gc = __popcll((e ^ (e >> 1)) & 0x5555555555555555ull & mask);   // C = 01, G = 10
    const int d2 = ly - l - idx;
    bool hit = loss_hit[((size_t)(l - l0) * gc_rows + gc) * 64 + d2] != 0;      // the table starts at end length l0, gc_rows rows per length
    if (!hit && d2 == 0) hit = dm_delta_g(e, l, dg) < dg_limit;
    idx_out = idx;
    return hit;
}

// The same search with G lanes per pair (G = 16 or 64, a sub-wave): the (end length, end expansion, y expansion) combinations
// are numbered in the reference's order — longest end first, then expansion of the end, then expansion of y — and taken G at a
// time, one per lane; the lowest lane with a hit IS the first passing combination.  A pair of plain primers has one
// combination per end length (14 in finDimer's mode): one step instead of a 14-step serial loop, and a degenerate pair walks
// its expansions G at a time.  Used when there are too few pairs to fill the chip with one thread each (the self-dimer test
// of the core step, the pair lists of the pairing stage, get_Maxprimerset's incremental scans).
template <int G>
__device__ inline bool dimer_pair_group(const uint8_t *__restrict__ codes, const int32_t *__restrict__ off, int x, int y, int mode,
                                        const uint8_t *__restrict__ loss_hit, int l0, int gc_rows, const double *__restrict__ dg,
                                        double dg_limit, int32_t (&rec)[4], int sub = 0, int split = 1) {
    const int lane = threadIdx.x & 63, gl = lane & (G - 1), g0 = lane & ~(G - 1);
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << g0;
    const int lx = off[x + 1] - off[x], ly = off[y + 1] - off[y];
    Nib cx, cy;
    cx.clear(); cy.clear();
    for (int p = 0; p < lx; p++) cx.set(p, codes[off[x] + p]);
    for (int p = 0; p < ly; p++) cy.set(p, codes[off[y] + p]);
    const uint32_t dy = dm_degeneracy(cy, 0, ly);
    int l_hi, l_lo;
    if (mode == 0) { l_hi = lx < 18 ? lx : 18; l_lo = lx < 5 ? lx : 5; }         // FD:162-169
    else { l_hi = lx - 1; l_lo = 5; }                                             // MS:149-154
    const int n_l = l_hi - l_lo + 1;                                              // <= 28 end lengths, j-th is l_hi - j
    // combinations per end length, two slots per lane (n_l <= 32 = 2 x 16)
    unsigned long long cnt[2] = {0, 0};
    for (int t = 0; t < 2; t++) {
        const int j = gl + t * G;
        const int l = l_hi - j;
        if (j < n_l && l > 0 && l <= ly) cnt[t] = (unsigned long long)dm_degeneracy(cx, lx - l, l) * dy;
    }
    unsigned long long total = 0;
    for (int j = 0; j < n_l; j++) total += __shfl(cnt[j / G], g0 + (j % G));
    // (sub, split): this group takes every split-th step of the walk, starting at step sub — a yes/no caller spreads one pair over
    // `split` groups and ORs their answers; the first passing combination is the lowest group's only when split = 1
    for (unsigned long long t0 = (unsigned long long)sub * G; t0 < total; t0 += (unsigned long long)split * G) {
        const unsigned long long id = t0 + gl;
        bool hit = false;
        int l = 0, idx = 0;
        uint32_t ei = 0, pi = 0;
        unsigned long long acc = 0;
        bool found = false;
        for (int j = 0; j < n_l; j++) {                                           // uniform trip count: every lane shuffles
            const unsigned long long cj = __shfl(cnt[j / G], g0 + (j % G));
            if (!found && id < acc + cj) {
                found = true;
                l = l_hi - j;
                const unsigned long long rem = id - acc;
                ei = (uint32_t)(rem / dy);
                pi = (uint32_t)(rem % dy);
            }
            acc += cj;
        }
        if (found) hit = dimer_combo(cx, cy, lx, ly, l, ei, pi, loss_hit, l0, gc_rows, dg, dg_limit, idx);
        const unsigned long long hb = __ballot(hit) & gmask;
        if (hb) {
            const int first = __ffsll((long long)hb) - 1;
            rec[0] = __shfl(l, first); rec[1] = (int32_t)__shfl(ei, first); rec[2] = (int32_t)__shfl(pi, first); rec[3] = __shfl(idx, first);
            return true;
        }
    }
    return false;
}

// Tables in LDS (north_star): the Loss decision bytes of the end lengths a launch can meet and the deltaG constants.
constexpr int kLossRow = (MP_DIMER_MAX_LEN + 1) * 64;                    // bytes per end length of the caller's table
constexpr int kLossL0 = 5, kLossRows = 27, kStageGc = 32;                // end lengths 5..31 x GC counts 0..31 are staged (54 KB); others read global
constexpr int kStageRow = kStageGc * 64;
constexpr int kNdg = 16 + 32 + MP_DIMER_MAX_LEN + 1 + 1;

// explicit ordered pairs or an all-pairs scan, G lanes per pair, tables staged in LDS; persistent workgroups stride over the pairs
// split > 1 (explicit pairs with flags only): `split` groups per pair, each on its own share of the pair's combinations; flags must
// be zero before the launch and a group writes only a hit (the self-dimer test of the core step has one pair per window primer —
// a few thousand at most — and a primer of degeneracy 64 has 57 000 combinations: one group would walk them alone for 0.14 ms)
template <int G>
__global__ __launch_bounds__(kBlock) void dimer_group_kernel(const DimerArgs A, long long n_pairs, const int32_t *__restrict__ pairs,
                                                             uint8_t *__restrict__ flags, int split) {
    __shared__ uint8_t s_loss[kLossRows * kStageRow];
    __shared__ double s_dg[kNdg];
    for (int i = threadIdx.x; i < kLossRows * kStageRow / 16; i += kBlock) {
        const int l = i / (kStageRow / 16), rest = i % (kStageRow / 16);                       // rest = 16-byte piece of [gc < 32][64]
        reinterpret_cast<uint4 *>(s_loss)[i] = reinterpret_cast<const uint4 *>(A.loss_hit + (size_t)(kLossL0 + l) * kLossRow)[rest];
    }
    for (int i = threadIdx.x; i < kNdg; i += kBlock) s_dg[i] = A.dg[i];
    __syncthreads();
    // pairs whose end lengths all lie in 5..31 use the staged rows; the rare others (primers shorter than 5, 32-mers) the global table
    const int per_block = kBlock / G;
    const int gl = threadIdx.x & (G - 1);
    for (long long vp = (long long)blockIdx.x * per_block + threadIdx.x / G; vp < n_pairs * split; vp += (long long)gridDim.x * per_block) {
        const long long p = vp / split;
        const int sub = (int)(vp % split);
        int x, y;
        if (pairs) { x = pairs[2 * p]; y = pairs[2 * p + 1]; }
        else {
            x = (int)(p / A.n); y = (int)(p % A.n);
            if (A.mode == 0 ? (y < x) : (x >= A.n_new && y >= A.n_new)) continue;
        }
        const int lx = A.off[x + 1] - A.off[x];
        const bool staged = lx <= 31 && (A.mode != 0 || lx >= 5);          // every end length of this pair lies in 5..31
        int32_t rec[4];
        const bool hit = staged ? dimer_pair_group<G>(A.codes, A.off, x, y, A.mode, s_loss, kLossL0, kStageGc, s_dg, A.dg_limit, rec, sub, split)
                                : dimer_pair_group<G>(A.codes, A.off, x, y, A.mode, A.loss_hit, 0, MP_DIMER_MAX_LEN + 1, A.dg, A.dg_limit, rec, sub, split);
        if (gl != 0) continue;
        if (flags) { if (split == 1) flags[p] = hit ? 1 : 0; else if (hit) flags[p] = 1; }
        else if (hit) {
            unsigned long long h = atomicAdd(A.n_hits, 1ull);
            if ((long long)h < A.cap) {
                int32_t *r = A.hits + 6 * h;
                r[0] = x; r[1] = y; r[2] = rec[0]; r[3] = rec[1]; r[4] = rec[2]; r[5] = rec[3];
            }
        }
    }
}

// ---- one thread per pair, for lists that fill the chip (the all-pairs scans of finDimer / get_Maxprimerset at database scale) and for
// every list that holds a primer of more than 32 bases (adaptor-tailed primers: 64-bit planes, 128-bit strings) ----
typedef unsigned __int128 u128;
template <bool LONG> struct PrimTraits;
template <> struct PrimTraits<false> {
    typedef uint32_t plane_t;          // one bit per position
    typedef uint64_t str_t;            // 2 bits per base
    static constexpr int kMax = 32, kNibWords = 2;
};
template <> struct PrimTraits<true> {
    typedef uint64_t plane_t;
    typedef u128 str_t;
    static constexpr int kMax = 64, kNibWords = 4;
};

// A primer as the pair test wants it (written once per call by prim_kernel): its symbol codes as nibbles, the base-set planes of the
// primer itself (y[b] bit i = base b allowed at position i) and of its reverse complement (r[b] bit i = base b allowed at position i
// of RC(primer)), its first expansion 2 bits per base, the positions holding more than one base, length, number of expansions.
template <bool LONG>
struct __align__(16) PrimRec {
    typedef typename PrimTraits<LONG>::plane_t plane_t;
    uint64_t nib[PrimTraits<LONG>::kNibWords];
    plane_t y[4];
    plane_t r[4];
    uint64_t base0[LONG ? 2 : 1];
    plane_t dmask;
    int32_t len;
    uint32_t deg;
    __device__ uint32_t code(int j) const { return (uint32_t)(nib[j >> 4] >> (4 * (j & 15))) & 15u; }
    __device__ typename PrimTraits<LONG>::str_t first() const {
        if constexpr (LONG) return (u128)base0[0] | ((u128)base0[1] << 64);
        else return base0[0];
    }
};

// member order of the symbols (the table dimer_init uploads), 2 bits per member, one byte per symbol code
constexpr uint64_t member_pack(int half) {
    const char *members[16] = {"", "A", "C", "AC", "G", "AG", "GC", "GAC", "T", "AT", "CT", "ATC", "GT", "GAT", "GTC", "ATGC"};
    uint64_t w = 0;
    for (int m = 0; m < 8; m++)
        for (int t = 0; t < 4 && members[half * 8 + m][t]; t++) {
            const char ch = members[half * 8 + m][t];
            w |= (uint64_t)(ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3) << (8 * m + 2 * t);
        }
    return w;
}
constexpr uint64_t kMemLo = member_pack(0), kMemHi = member_pack(1);
__device__ inline uint32_t member_of(uint32_t code, uint32_t r) { return (uint32_t)(((code & 8u) ? kMemHi : kMemLo) >> (8 * (code & 7u) + 2 * r)) & 3u; }

template <bool LONG>
__global__ __launch_bounds__(kBlock) void prim_kernel(const uint8_t *__restrict__ codes, const int32_t *__restrict__ off, int n,
                                                      PrimRec<LONG> *__restrict__ out) {
    typedef typename PrimTraits<LONG>::plane_t plane_t;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int len = off[i + 1] - off[i];
    PrimRec<LONG> o;
#pragma unroll
    for (int t = 0; t < PrimTraits<LONG>::kNibWords; t++) o.nib[t] = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) o.y[b] = o.r[b] = 0;
    o.base0[0] = 0;
    if constexpr (LONG) o.base0[1] = 0;
    plane_t dmask = 0;
    uint32_t deg = 1;
    for (int p = 0; p < len; p++) {
        const uint32_t c = codes[off[i] + p];
#pragma unroll
        for (int t = 0; t < PrimTraits<LONG>::kNibWords; t++)
            if ((p >> 4) == t) o.nib[t] |= (uint64_t)c << (4 * (p & 15));
        deg *= (uint32_t)__popc(c);
        dmask |= (plane_t)(__popc(c) > 1) << p;
        const uint64_t first = member_of(c, 0);
        if (p < 32) o.base0[0] |= first << (2 * p);
        else if constexpr (LONG) o.base0[1] |= first << (2 * (p - 32));
#pragma unroll
        for (int b = 0; b < 4; b++) {
            o.y[b] |= (plane_t)((c >> b) & 1u) << p;
            o.r[b] |= (plane_t)((c >> (3 - b)) & 1u) << (len - 1 - p);      // complement: A <-> T, C <-> G (bit b <-> bit 3 - b), reversed
        }
    }
    o.dmask = dmask; o.len = len; o.deg = deg;
    out[i] = o;
}

template <typename P> __device__ inline int low_bit(P m) { return sizeof(P) == 8 ? __ffsll((long long)m) - 1 : __ffs((int)m) - 1; }
template <typename P> __device__ inline int high_bit(P m) { return sizeof(P) == 8 ? 63 - __clzll((long long)m) : 31 - __clz((int)m); }

// expansions of positions [start, start + len): only the positions holding several bases are visited
template <bool LONG>
__device__ inline uint32_t fast_degeneracy(const PrimRec<LONG> &R, int start, int len) {
    typedef typename PrimTraits<LONG>::plane_t plane_t;
    plane_t m = (R.dmask >> start) & (len >= PrimTraits<LONG>::kMax ? ~(plane_t)0 : (((plane_t)1 << len) - 1));
    uint32_t d = 1;
    while (m) {
        const int p = low_bit(m);
        m &= m - 1;
        d *= (uint32_t)__popc(R.code(start + p));
    }
    return d;
}

// expansion number idx of positions [start, start + len) (itertools.product order: last position fastest), 2 bits per base
template <bool LONG>
__device__ inline typename PrimTraits<LONG>::str_t fast_expand(const PrimRec<LONG> &R, int start, int len, uint32_t idx) {
    typedef typename PrimTraits<LONG>::plane_t plane_t;
    typedef typename PrimTraits<LONG>::str_t str_t;
    const bool all = len >= PrimTraits<LONG>::kMax;
    str_t x = (R.first() >> (2 * start)) & (all ? ~(str_t)0 : (((str_t)1 << (2 * len)) - 1));
    plane_t m = (R.dmask >> start) & (all ? ~(plane_t)0 : (((plane_t)1 << len) - 1));
    while (m) {
        const int p = high_bit(m);
        m &= ~((plane_t)1 << p);
        const uint32_t code = R.code(start + p);
        const uint32_t sz = (uint32_t)__popc(code);                                  // 2, 3 or 4
        const uint32_t q = sz == 3 ? __umulhi(idx, 0xAAAAAAABu) >> 1 : idx >> (sz >> 1);
        const uint32_t r = idx - q * sz;
        idx = q;
        x = (x & ~((str_t)3 << (2 * p))) | ((str_t)member_of(code, r) << (2 * p));
    }
    return x;
}

// reverse complement of an l-base string (complement = 3 - base = ~base): bit reversal turns the 2-bit groups around and
// swaps the two bits inside each, which one mask-shift undoes
__device__ inline uint64_t fast_rc(uint64_t e, int l) {
    uint64_t r = __brevll(~e);
    r = ((r & 0x5555555555555555ull) << 1) | ((r >> 1) & 0x5555555555555555ull);
    return r >> (64 - 2 * l);
}
__device__ inline u128 fast_rc(u128 e, int l) {
    const u128 n = ~e;
    uint64_t hi = __brevll((uint64_t)n), lo = __brevll((uint64_t)(n >> 64));      // reversed halves change places
    hi = ((hi & 0x5555555555555555ull) << 1) | ((hi >> 1) & 0x5555555555555555ull);
    lo = ((lo & 0x5555555555555555ull) << 1) | ((lo >> 1) & 0x5555555555555555ull);
    return (((u128)hi << 64) | lo) >> (128 - 2 * l);
}
__device__ inline int gc_count(uint64_t e, uint64_t mask) { return __popcll((e ^ (e >> 1)) & 0x5555555555555555ull & mask); }      // C = 01, G = 10
__device__ inline int gc_count(u128 e, u128 mask) {
    const u128 x = (e ^ (e >> 1)) & mask;
    return __popcll((uint64_t)x & 0x5555555555555555ull) + __popcll((uint64_t)(x >> 64) & 0x5555555555555555ull);
}

struct PairEnds { int l_hi, l_lo; };
__device__ inline PairEnds end_lengths(int lx, int mode) {
    PairEnds e;
    if (mode == 0) { e.l_hi = lx < 18 ? lx : 18; e.l_lo = lx < 5 ? lx : 5; }      // FD:162-169
    else { e.l_hi = lx - 1; e.l_lo = 5; }                                         // MS:149-154
    return e;
}

// The ordered pair x -> y.  RC(end of length l) is the first l symbols of RC(x), so "some expansion of the end occurs reverse-
// complemented in some expansion of y at offset s" needs the base sets of RC(x)[0..l) and y[s..s+l) to intersect position by
// position: M(s) = length of that run, from four AND-ORs of bit planes and a count of trailing ones.  No end longer than
// max_s M(s) can match in any expansion, and an end can only sit at an offset with M(s) >= the shortest end length: the filter
// returns max M and the candidate offsets, 0 / nothing for most pairs.
template <bool LONG>
__device__ inline int pair_filter(const PrimRec<LONG> &X, const PrimRec<LONG> &Y, int l_lo, typename PrimTraits<LONG>::plane_t &cand) {
    typedef typename PrimTraits<LONG>::plane_t plane_t;
    int max_m = 0;
    plane_t cd = 0;
    const int ly = Y.len;
    for (int s = 0; s + l_lo <= ly; s++) {
        const plane_t m = (X.r[0] & (Y.y[0] >> s)) | (X.r[1] & (Y.y[1] >> s)) | (X.r[2] & (Y.y[2] >> s)) | (X.r[3] & (Y.y[3] >> s));
        const int run = m == ~(plane_t)0 ? PrimTraits<LONG>::kMax : low_bit((plane_t)~m);
        max_m = run > max_m ? run : max_m;
        cd |= (plane_t)(run >= l_lo) << s;
    }
    cand = cd;
    return max_m;
}

// The reference's search (longest end first, expansions of the end, expansions of y; first passing combination) over the end
// lengths the filter left, looking for RC(end) only at the candidate offsets.
template <bool LONG>
__device__ inline bool pair_search(const PrimRec<LONG> &X, const PrimRec<LONG> &Y, PairEnds E, int max_m, typename PrimTraits<LONG>::plane_t cand,
                                   const uint8_t *__restrict__ loss_hit, const double *__restrict__ dg, double dg_limit, int32_t (&rec)[4]) {
    typedef typename PrimTraits<LONG>::plane_t plane_t;
    typedef typename PrimTraits<LONG>::str_t str_t;
    const int lx = X.len, ly = Y.len;
    const str_t by = Y.first();
    const uint32_t dy = Y.deg;
    for (int l = E.l_hi < max_m ? E.l_hi : max_m; l >= E.l_lo; l--) {
        if (l > ly) continue;
        const str_t mask = l >= PrimTraits<LONG>::kMax ? ~(str_t)0 : (((str_t)1 << (2 * l)) - 1);
        const uint32_t de = fast_degeneracy<LONG>(X, lx - l, l);
        for (uint32_t ei = 0; ei < de; ei++) {
            const str_t e = fast_expand<LONG>(X, lx - l, l, ei);
            const str_t rc = fast_rc(e, l);
            for (uint32_t pi = 0; pi < dy; pi++) {
                const str_t p = dy == 1 ? by : fast_expand<LONG>(Y, 0, ly, pi);
                int idx = -1;
                for (plane_t cs = cand; cs; cs &= cs - 1) {                        // str.find: first occurrence
                    const int s = low_bit(cs);
                    if (s + l > ly) break;
                    if (((p >> (2 * s)) & mask) == rc) { idx = s; break; }
                }
                if (idx < 0) continue;
                const int gc = gc_count(e, mask);
                const int d2 = ly - l - idx;
                bool hit = loss_hit[((size_t)l * (MP_DIMER_MAX_LEN + 1) + gc) * 64 + d2] != 0;
                if (!hit && d2 == 0) hit = dm_delta_g(e, l, dg) < dg_limit;
                if (hit) { rec[0] = l; rec[1] = (int32_t)ei; rec[2] = (int32_t)pi; rec[3] = idx; return true; }
            }
        }
    }
    return false;
}

// All pairs of n primers: workgroup = one primer x (its record is wave-uniform), threads stride over the partners y the mode asks
// for.  Every partner goes through the bit-plane filter; the few that pass are queued in LDS and searched 256 at a time, so the
// search runs with full waves instead of one or two lanes of every wave.  Hits are collected in LDS and appended to the output
// with ONE device atomic per flush (~n atomics on the shared counter per launch instead of one per hit).
constexpr int kHitBuf = 1024, kQueue = 2 * kBlock;
template <bool LONG>
__global__ __launch_bounds__(kBlock) void dimer_rows_kernel(const DimerArgs A, const PrimRec<LONG> *__restrict__ prim) {
    typedef typename PrimTraits<LONG>::plane_t plane_t;
    __shared__ int32_t s_rec[kHitBuf][6];
    __shared__ int32_t s_qy[kQueue];
    __shared__ plane_t s_qc[kQueue];
    __shared__ uint8_t s_qm[kQueue];
    __shared__ int s_n, s_q;
    __shared__ unsigned long long s_base;
    const int x = blockIdx.x;
    const PrimRec<LONG> X = prim[x];
    const PairEnds E = end_lengths(X.len, A.mode);
    int y0, y1;                                                     // partners: FD:206-209 (y >= x), MS:199-204 (pairs touching a new primer)
    if (A.mode == 0) { y0 = x; y1 = A.n; }
    else { y0 = 0; y1 = x < A.n_new ? A.n : A.n_new; }
    if (threadIdx.x == 0) { s_n = 0; s_q = 0; }
    __syncthreads();
    if (E.l_hi < E.l_lo || E.l_hi < 1) return;                      // no end length to try (mode 1, primers of <= 5 bases)
    for (int yb = y0; yb < y1; yb += kBlock) {
        const int y = yb + threadIdx.x;
        const bool last = yb + kBlock >= y1;
        if (y < y1) {
            plane_t cand;
            const int max_m = pair_filter<LONG>(X, prim[y], E.l_lo, cand);
            if (max_m >= E.l_lo) {
                const int at = atomicAdd(&s_q, 1);
                s_qy[at] = y; s_qc[at] = cand; s_qm[at] = (uint8_t)max_m;
            }
        }
        __syncthreads();
        const int queued = s_q;
        if (queued >= kBlock || last) {                             // (queued < 2 * kBlock: at most kBlock - 1 were left, kBlock came)
            for (int i = threadIdx.x; i < queued; i += kBlock) {
                const int yq = s_qy[i];
                int32_t rec[4];
                if (pair_search<LONG>(X, prim[yq], E, s_qm[i], s_qc[i], A.loss_hit, A.dg, A.dg_limit, rec)) {
                    const int at = atomicAdd(&s_n, 1);
                    int32_t *r = s_rec[at];
                    r[0] = x; r[1] = yq; r[2] = rec[0]; r[3] = rec[1]; r[4] = rec[2]; r[5] = rec[3];
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) s_q = 0;
            const int held = s_n;
            if (held + kQueue > kHitBuf || last) {                  // the next drain might not fit, or this was the last one
                if (held) {
                    if (threadIdx.x == 0) s_base = atomicAdd(A.n_hits, (unsigned long long)held);
                    __syncthreads();
                    const unsigned long long base = s_base;
                    for (int i = threadIdx.x; i < held * 6; i += kBlock) {
                        const unsigned long long h = base + (unsigned long long)(i / 6);
                        if ((long long)h < A.cap) A.hits[6 * h + i % 6] = s_rec[i / 6][i % 6];
                    }
                    __syncthreads();
                    if (threadIdx.x == 0) s_n = 0;
                }
            }
            __syncthreads();
        }
    }
}

// explicit ordered pairs (get_multiPrime_V8.py:419-438): one thread per pair, any passing combination
template <bool LONG>
__global__ __launch_bounds__(kBlock) void dimer_pairs_kernel(const PrimRec<LONG> *__restrict__ prim, long long n_pairs,
                                                             const int32_t *__restrict__ pairs, const uint8_t *__restrict__ loss_hit,
                                                             const double *__restrict__ dg, double dg_limit, uint8_t *__restrict__ flags) {
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_pairs) return;
    const PrimRec<LONG> X = prim[pairs[2 * p]], Y = prim[pairs[2 * p + 1]];
    const PairEnds E = end_lengths(X.len, 0);
    typename PrimTraits<LONG>::plane_t cand;
    const int max_m = pair_filter<LONG>(X, Y, E.l_lo, cand);
    int32_t rec[4];
    flags[p] = max_m >= E.l_lo && pair_search<LONG>(X, Y, E, max_m, cand, loss_hit, dg, dg_limit, rec) ? 1 : 0;
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


// ----------------------------------------------------------------------------------------------
// (7) exact in-silico PCR (extract_PCR_product_V1.py:189-216)
// ----------------------------------------------------------------------------------------------
// thread = (primer pair, sequence).  The sequence is scanned with a rolling 2-bit window (newest base
// in the top position, so window position j sits at bits 2j like dm_expand's packing); any character
// other than upper-case A/C/G/T resets the window — the reference's regex search is case sensitive.
__device__ inline int pcr_base(uint8_t ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1; }

// first occurrence of the k-base pattern `pat` in s[from, to) (entirely inside), or -1
__device__ inline int pcr_find(const uint8_t *__restrict__ s, int from, int to, uint64_t pat, int k) {
    const uint64_t mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1ull);
    uint64_t win = 0;
    int valid = 0;
    for (int pos = from; pos < to; pos++) {
        const int b = pcr_base(s[pos]);
        if (b < 0) { valid = 0; win = 0; continue; }
        win = (win >> 2) | ((uint64_t)b << (2 * (k - 1)));
        if (++valid >= k && (win & mask) == pat) return pos - k + 1;
    }
    return -1;
}

// Primers of 33..MP_PATTERN_MAX_LEN bases (adaptor-tailed primers) do not fit the 64-bit rolling window: the per-thread search
// then compares characters — expansion `idx` of the codes (itertools.product order: last position fastest) written out as bases,
// or as the bases of its reverse complement, and matched position by position.  Only the fallback paths use it (pair tables
// beyond 4096 expansions, sequences with more occurrences than the block kernel's list holds).
__device__ inline void pcr_expansion(const uint8_t *__restrict__ codes, int L, unsigned long long idx, bool rc, uint8_t (&pat)[MP_PATTERN_MAX_LEN]) {
    for (int j = L - 1; j >= 0; j--) {
        const uint32_t m = codes[j] & 15u, sz = c_msize[m];
        const uint8_t b = c_member[m][idx % sz];
        idx /= sz;
        if (rc) pat[L - 1 - j] = (uint8_t)(3 - b);
        else pat[j] = b;
    }
}
__device__ inline unsigned long long pcr_degeneracy(const uint8_t *__restrict__ codes, int L) {
    unsigned long long d = 1;
    for (int j = 0; j < L; j++) d *= c_msize[codes[j] & 15u];          // check_primers bounds the product
    return d;
}
__device__ inline int pcr_find_chars(const uint8_t *__restrict__ s, int from, int to, const uint8_t (&pat)[MP_PATTERN_MAX_LEN], int L) {
    for (int pos = from; pos + L <= to; pos++) {
        int j = 0;
        while (j < L && pcr_base(s[pos + j]) == (int)pat[j]) j++;
        if (j == L) return pos;
    }
    return -1;
}
// the reference's order for one (pair, sequence) on characters: first forward expansion that occurs and whose Product holds a
// reverse expansion (extract_PCR_product_V1.py:189-216)
__device__ inline void pcr_resolve_chars(const uint8_t *__restrict__ s, int len, const uint8_t *__restrict__ cf, int lf,
                                         const uint8_t *__restrict__ cr, int lr, int32_t (&res)[4]) {
    const unsigned long long df = pcr_degeneracy(cf, lf), dr = pcr_degeneracy(cr, lr);
    uint8_t f[MP_PATTERN_MAX_LEN], r[MP_PATTERN_MAX_LEN];
    for (unsigned long long fi = 0; fi < df && res[0] < 0; fi++) {
        pcr_expansion(cf, lf, fi, false, f);
        const int p1 = pcr_find_chars(s, 0, len, f, lf);
        if (p1 < 0) continue;
        const int p2 = pcr_find_chars(s, p1 + lf, len, f, lf);
        const int end = p2 < 0 ? len : p2;
        for (unsigned long long ri = 0; ri < dr; ri++) {
            pcr_expansion(cr, lr, ri, true, r);
            const int q = pcr_find_chars(s, p1, end, r, lr);
            if (q >= 0) { res[0] = (int32_t)fi; res[1] = p1; res[2] = (int32_t)ri; res[3] = q; break; }
        }
    }
}

template <bool LONG>
__global__ __launch_bounds__(kBlock) void pcr_kernel(const uint8_t *__restrict__ bytes, const int64_t *__restrict__ row_off,
                                                     int n_rows, const uint8_t *__restrict__ codes,
                                                     const int32_t *__restrict__ off, int32_t *__restrict__ out) {
    const int row = blockIdx.x * kBlock + threadIdx.x, p = blockIdx.y;
    if (row >= n_rows) return;
    const uint8_t *s = bytes + row_off[row];
    const int len = (int)(row_off[row + 1] - row_off[row]);
    const int lf = off[2 * p + 1] - off[2 * p], lr = off[2 * p + 2] - off[2 * p + 1];
    if constexpr (LONG) {
        int32_t res[4] = {-1, -1, -1, -1};
        pcr_resolve_chars(s, len, codes + off[2 * p], lf, codes + off[2 * p + 1], lr, res);
        int32_t *o = out + ((size_t)p * n_rows + row) * 4;
        o[0] = res[0]; o[1] = res[1]; o[2] = res[2]; o[3] = res[3];
        return;
    }
    Nib cf, cr;
    cf.clear(); cr.clear();
    for (int j = 0; j < lf; j++) cf.set(j, codes[off[2 * p] + j]);
    for (int j = 0; j < lr; j++) cr.set(j, codes[off[2 * p + 1] + j]);
    const uint32_t df = dm_degeneracy(cf, 0, lf), dr = dm_degeneracy(cr, 0, lr);
    int32_t res[4] = {-1, -1, -1, -1};
    for (uint32_t fi = 0; fi < df && res[0] < 0; fi++) {                   // for sequence in Fseq
        const uint64_t f = dm_expand(cf, 0, lf, fi);
        const int p1 = pcr_find(s, 0, len, f, lf);                         // re.search(sequence, i)
        if (p1 < 0) continue;
        const int p2 = pcr_find(s, p1 + lf, len, f, lf);                   // next non-overlapping occurrence (str.split)
        const int end = p2 < 0 ? len : p2;                                 // Product = sequence + line[1]
        for (uint32_t ri = 0; ri < dr; ri++) {                             // for sequence2 in Rseq
            const uint64_t e = dm_expand(cr, 0, lr, ri);
            uint64_t rc = 0;
            for (int t = 0; t < lr; t++) rc |= (uint64_t)(3u - ((uint32_t)(e >> (2 * (lr - 1 - t))) & 3u)) << (2 * t);
            const int q = pcr_find(s, p1, end, rc, lr);                    // re.search(RC(sequence2), Product)
            if (q >= 0) { res[0] = (int32_t)fi; res[1] = p1; res[2] = (int32_t)ri; res[3] = q; break; }
        }
    }
    int32_t *o = out + ((size_t)p * n_rows + row) * 4;
    o[0] = res[0]; o[1] = res[1]; o[2] = res[2]; o[3] = res[3];
}


// ---- block-per-sequence form ---------------------------------------------------------------------------------------
// One workgroup owns one sequence.  Scan: every forward expansion and every reverse-complemented reverse expansion of
// every pair is a pattern {2-bit word, length mask}; the sequence is packed segment by segment into LDS (2 bits per base +
// a "matches nothing" bit for everything that is not an upper-case A/C/G/T: the reference's regex search is case
// sensitive) and every thread slides over its positions — per (position, pattern): XOR, fold, OR, AND, compare; the
// pattern table is wave-uniform and arrives through scalar loads.  Occurrences go into an LDS list.  Resolve: one thread
// per pair walks the reference's order (first forward expansion that occurs and whose Product holds a reverse expansion)
// on that short list.  A sequence with more occurrences than the list holds falls back to the rolling scan above.
// Traffic: the text once (1 byte per base); the work is integer VALU, ~6 wave-instructions per (64 positions, pattern).
constexpr int kPcrSeg = 4096;                        // positions packed per round
constexpr int kPcrHits = 3072;                       // occurrences kept per sequence

// NW = 64-bit words of a pattern: 1 up to 32 bases, 2 up to MP_PATTERN_MAX_LEN = 64
template <int NW>
struct PcrPat { unsigned long long word[NW], lenmask[NW]; int32_t len, pad; };
struct PcrPair { int32_t f0, nf, r0, nr; };         // pattern ranges of a pair: forward expansions, then RC(reverse expansions)

__device__ inline int pcr_first(const uint32_t *__restrict__ s_pos, const uint16_t *__restrict__ s_pat, int n, int pat, int from, int to_excl) {
    int best = 0x7fffffff;                           // smallest position >= from with position <= to_excl (caller subtracts the length)
    for (int i = 0; i < n; i++)
        if ((int)s_pat[i] == pat) { const int q = (int)s_pos[i]; if (q >= from && q <= to_excl && q < best) best = q; }
    return best == 0x7fffffff ? -1 : best;
}

// RES: the packed segment comes from the context's resident store (mp_seq_load, scan.hip) instead of the packing loop over characters;
// the characters are only touched by the overflow fall-back at the end
template <int NW, bool RES>
__global__ __launch_bounds__(kBlock) void pcr_block_kernel(const uint8_t *__restrict__ bytes, const int64_t *__restrict__ row_off, int n_rows,
                                                           const unsigned long long *__restrict__ st_code, const unsigned long long *__restrict__ st_flag,
                                                           const int64_t *__restrict__ st_woff,
                                                           const PcrPat<NW> *__restrict__ pats, int n_pats, const PcrPair *__restrict__ pairs,
                                                           int n_pairs, const uint8_t *__restrict__ codes, const int32_t *__restrict__ off,
                                                           int32_t *__restrict__ out, int prefilter) {
    constexpr int kPcrSegWords = kPcrSeg / 32 + 1 + NW;
    __shared__ unsigned long long s_b[kPcrSegWords], s_n[kPcrSegWords];
    __shared__ uint32_t s_pos[kPcrHits];
    __shared__ uint16_t s_pat[kPcrHits];
    __shared__ int s_nh;
    // [r6] prefilter (every pattern has >= 8 bases): one bit per 8-base prefix that SOME pattern starts with (65536 bits); a position
    // whose first 8 bases are all upper-case A/C/G/T and whose prefix bit is set is queued, and only the queued positions — a few per
    // cent — are compared with the pattern table, one position per lane.  Without it every position paid ~6 instructions for each of
    // the up to 4096 patterns: 3.3 ms for 20 727 sequences x 64 pairs, integer-VALU bound.
    __shared__ uint32_t s_bits[2048];
    __shared__ uint16_t s_q[kPcrSeg];
    __shared__ int s_nq;
    if (prefilter) {
        for (int i = threadIdx.x; i < 2048; i += kBlock) s_bits[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n_pats; i += kBlock) {
            const uint32_t pfx = (uint32_t)pats[i].word[0] & 0xFFFFu;
            atomicOr(&s_bits[pfx >> 5], 1u << (pfx & 31));
        }
    }
    const int row = blockIdx.x;
    const uint8_t *s = bytes + row_off[row];
    const int len = (int)(row_off[row + 1] - row_off[row]);
    if (threadIdx.x == 0) s_nh = 0;
    const unsigned long long kOdd = 0x5555555555555555ull;
    for (int base = 0; base < len; base += kPcrSeg) {
        __syncthreads();
        if (threadIdx.x == 0) s_nq = 0;
        for (int w = threadIdx.x; w < kPcrSegWords; w += kBlock) {
            unsigned long long b = 0, n = 0;
            const int p0 = base + w * 32;
            if (RES) {
                const long long gw = base / 32 + w, nwords = st_woff[row + 1] - st_woff[row];
                if (gw < nwords) {                   // the search is case sensitive: a lower-case base matches nothing either
                    const unsigned long long f = st_flag[st_woff[row] + gw];
                    b = st_code[st_woff[row] + gw]; n = (f | (f >> 1)) & kOdd;
                    b &= ~(n | (n << 1));            // (the packing loop leaves code 0 where nothing matches)
                } else n = kOdd;
                s_b[w] = b; s_n[w] = n;
                continue;
            }
            for (int j = 0; j < 32; j++) {
                const int p = p0 + j;
                const int c = p < len ? pcr_base(s[p]) : -1;
                b |= (unsigned long long)(c < 0 ? 0 : c) << (2 * j);
                n |= (unsigned long long)(c < 0) << (2 * j);
            }
            s_b[w] = b; s_n[w] = n;
        }
        __syncthreads();
        if (prefilter) {
            for (int q = threadIdx.x; q < kPcrSeg; q += kBlock) {
                if (base + q >= len) break;
                const int w = q >> 5, sh = (q & 31) * 2;
                unsigned long long w0 = s_b[w] >> sh, n0 = s_n[w] >> sh;
                if (sh) { w0 |= s_b[w + 1] << (64 - sh); n0 |= s_n[w + 1] << (64 - sh); }
                const uint32_t pfx = (uint32_t)w0 & 0xFFFFu;
                if (((uint32_t)n0 & 0xFFFFu) == 0 && ((s_bits[pfx >> 5] >> (pfx & 31)) & 1u)) s_q[atomicAdd(&s_nq, 1)] = (uint16_t)q;
            }
            __syncthreads();
        }
        const int n_todo = prefilter ? s_nq : kPcrSeg;
        for (int qi = threadIdx.x; qi < n_todo; qi += kBlock) {
            const int q = prefilter ? (int)s_q[qi] : qi;
            const int p = base + q;
            if (p >= len) break;
            const int w = q >> 5, sh = (q & 31) * 2;
            unsigned long long win[NW], nw[NW];
#pragma unroll
            for (int t = 0; t < NW; t++) {
                win[t] = s_b[w + t] >> sh; nw[t] = s_n[w + t] >> sh;
                if (sh) { win[t] |= s_b[w + t + 1] << (64 - sh); nw[t] |= s_n[w + t + 1] << (64 - sh); }
            }
            for (int i = 0; i < n_pats; i++) {
                const PcrPat<NW> P = pats[i];                              // uniform index: scalar loads
                unsigned long long diff = 0;
#pragma unroll
                for (int t = 0; t < NW; t++) {
                    const unsigned long long x = win[t] ^ P.word[t];
                    diff |= (((x | (x >> 1)) & kOdd) | nw[t]) & P.lenmask[t];
                }
                if (diff == 0 && p + P.len <= len) {
                    const int h = atomicAdd(&s_nh, 1);
                    if (h < kPcrHits) { s_pos[h] = (uint32_t)p; s_pat[h] = (uint16_t)i; }
                }
            }
        }
    }
    __syncthreads();
    const int nh = s_nh;
    for (int pr = threadIdx.x; pr < n_pairs; pr += kBlock) {
        int32_t res[4] = {-1, -1, -1, -1};
        if (nh <= kPcrHits) {
            const PcrPair Q = pairs[pr];
            for (int fi = 0; fi < Q.nf && res[0] < 0; fi++) {                                   // for sequence in Fseq
                const int lf = pats[Q.f0 + fi].len;
                const int p1 = pcr_first(s_pos, s_pat, nh, Q.f0 + fi, 0, len);                   // re.search(sequence, i)
                if (p1 < 0) continue;
                const int p2 = pcr_first(s_pos, s_pat, nh, Q.f0 + fi, p1 + lf, len);             // next non-overlapping occurrence (str.split)
                const int end = p2 < 0 ? len : p2;                                               // Product = sequence + line[1]
                for (int ri = 0; ri < Q.nr; ri++) {                                              // for sequence2 in Rseq
                    const int lr = pats[Q.r0 + ri].len;
                    const int q = pcr_first(s_pos, s_pat, nh, Q.r0 + ri, p1, end - lr);          // re.search(RC(sequence2), Product)
                    if (q >= 0) { res[0] = fi; res[1] = p1; res[2] = ri; res[3] = q; break; }
                }
            }
        } else {
            // too many occurrences for the list (low-complexity sequence): the rolling scan of pcr_kernel for this (pair, sequence)
            const int lf = off[2 * pr + 1] - off[2 * pr], lr = off[2 * pr + 2] - off[2 * pr + 1];
            if constexpr (NW > 1) {
                pcr_resolve_chars(s, len, codes + off[2 * pr], lf, codes + off[2 * pr + 1], lr, res);
                int32_t *o = out + ((size_t)pr * n_rows + row) * 4;
                o[0] = res[0]; o[1] = res[1]; o[2] = res[2]; o[3] = res[3];
                continue;
            }
            Nib cf, cr;
            cf.clear(); cr.clear();
            for (int j = 0; j < lf; j++) cf.set(j, codes[off[2 * pr] + j]);
            for (int j = 0; j < lr; j++) cr.set(j, codes[off[2 * pr + 1] + j]);
            const uint32_t df = dm_degeneracy(cf, 0, lf), dr = dm_degeneracy(cr, 0, lr);
            for (uint32_t fi = 0; fi < df && res[0] < 0; fi++) {
                const uint64_t f = dm_expand(cf, 0, lf, fi);
                const int p1 = pcr_find(s, 0, len, f, lf);
                if (p1 < 0) continue;
                const int p2 = pcr_find(s, p1 + lf, len, f, lf);
                const int end = p2 < 0 ? len : p2;
                for (uint32_t ri = 0; ri < dr; ri++) {
                    const uint64_t e = dm_expand(cr, 0, lr, ri);
                    uint64_t rc = 0;
                    for (int t = 0; t < lr; t++) rc |= (uint64_t)(3u - ((uint32_t)(e >> (2 * (lr - 1 - t))) & 3u)) << (2 * t);
                    const int q = pcr_find(s, p1, end, rc, lr);
                    if (q >= 0) { res[0] = (int32_t)fi; res[1] = p1; res[2] = (int32_t)ri; res[3] = q; break; }
                }
            }
        }
        int32_t *o = out + ((size_t)pr * n_rows + row) * 4;
        o[0] = res[0]; o[1] = res[1]; o[2] = res[2]; o[3] = res[3];
    }
}


}  // namespace

namespace mp {
int dimer_init() {
    // member order of every IUPAC symbol as the reference lists it (V20:105-107, FD:46-48), bases as A0 C1 G2 T3
    static const char *members[16] = {"", "A", "C", "AC", "G", "AG", "GC", "GAC", "T", "AT", "CT", "ATC", "GT", "GAT", "GTC", "ATGC"};
    uint8_t msize[16], member[16][4];
    for (int m = 0; m < 16; m++) {
        msize[m] = (uint8_t)strlen(members[m]);
        for (int t = 0; t < 4; t++) {
            char ch = t < msize[m] ? members[m][t] : 'A';
            member[m][t] = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
        }
    }
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_msize), msize, 16) != hipSuccess) return MP_ERR_DEVICE;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_member), member, 64) != hipSuccess) return MP_ERR_DEVICE;
    return MP_OK;
}
}  // namespace mp

namespace {
static int check_primers(mp_ctx *c, int32_t n, const uint8_t *codes, const int32_t *off, int max_len = MP_DIMER_MAX_LEN, int *longest = nullptr) {
    static const int msize[16] = {0, 1, 1, 2, 1, 2, 2, 3, 1, 2, 2, 3, 2, 3, 3, 4};
    if (longest) *longest = 0;
    for (int32_t i = 0; i < n; i++) {
        int len = off[i + 1] - off[i];
        if (len < 1 || len > max_len) return fail(c, MP_ERR_ARG, "primer %d has length %d (1..%d supported)", i, len, max_len);
        if (longest && len > *longest) *longest = len;
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
// The Loss decision table and the deltaG constants live in the context and are uploaded only when their contents change
// (get_Maxprimerset calls the scan once per candidate pair with the same tables).
static int ensure_dimer_tables(mp_ctx *c, const uint8_t *loss_hit, const double *dg) {
    const size_t tbl = (size_t)(MP_DIMER_MAX_LEN + 1) * kLossRow;
    int rc;
    if (!c->dm_loss) {
        if ((rc = dev_alloc(c, &c->dm_loss, tbl))) return rc;
        if ((rc = dev_alloc(c, &c->dm_dg, (size_t)kNdg))) return rc;
        c->dm_loss_host.clear();
    }
    if (c->dm_loss_host.size() != tbl || memcmp(c->dm_loss_host.data(), loss_hit, tbl) != 0) {
        HIPCK(c, hipMemcpyAsync(c->dm_loss, loss_hit, tbl, hipMemcpyHostToDevice, c->stream));
        c->dm_loss_host.assign(loss_hit, loss_hit + tbl);
    }
    if (c->dm_dg_host.size() != (size_t)kNdg || memcmp(c->dm_dg_host.data(), dg, sizeof(double) * kNdg) != 0) {
        HIPCK(c, hipMemcpyAsync(c->dm_dg, dg, sizeof(double) * kNdg, hipMemcpyHostToDevice, c->stream));
        c->dm_dg_host.assign(dg, dg + kNdg);
    }
    return MP_OK;
}

// device scratch of one call, released on every path
struct Scratch {
    mp_ctx *c;
    std::vector<std::pair<void **, size_t>> bufs;
    explicit Scratch(mp_ctx *c_) : c(c_) {}
    template <typename T>
    int alloc(T **p, size_t n) {
        int rc = dev_alloc(c, p, n);
        if (rc == MP_OK) bufs.push_back({(void **)p, (n ? n : 1) * sizeof(T)});
        return rc;
    }
    ~Scratch() {
        for (auto &b : bufs)
            if (*b.first) {                          // (dev_free without the type)
                void *q = *b.first;
                const size_t real = pool_forget(c, q, b.second);
                if (!real || !pool_give(c, q, real)) (void)hipFree(q);
                c->bytes -= (int64_t)(real ? real : b.second);
                *b.first = nullptr;
            }
    }
};

// lanes per pair: one thread per pair once there are enough pairs to fill the chip several times over, else a sub-wave
// dimer_group_kernel's workgroups stage 54 KB of tables into LDS before their first pair and then stride over the pairs: two fit a
// CU, so two per CU is the whole grid (a grid of one workgroup per four pairs staged the tables 1800 times for the core step's
// self-dimer test)
constexpr long long kGroupGrid = 2 * 256;

static int lanes_per_pair(long long n_pairs, int longest) {
    if (longest > 32) return 1;                         // primers of more than 32 bases: only the thread-per-pair kernels hold them
    if (const char *e = getenv("MP_DIMER_LANES")) { int g = atoi(e); if (g == 1 || g == 16 || g == 64) return g; }
    return n_pairs <= 32768 ? 64 : (n_pairs <= 262144 ? 16 : 1);
}

}  // namespace

extern "C" {

int mp_dimer_scan(mp_ctx *c, int32_t n, const uint8_t *codes, const int32_t *off, int32_t mode, int32_t n_new,
                  const uint8_t *loss_hit, const double *dg, double dg_limit, int64_t cap, int32_t *hits, int64_t *n_hits) {
    if (!c) return MP_ERR_ARG;
    if (n < 0 || !codes || !off || !loss_hit || !dg || !n_hits || cap < 0 || (cap && !hits) || (mode != 0 && mode != 1))
        return fail(c, MP_ERR_ARG, "mp_dimer_scan: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    *n_hits = 0;
    if (n == 0) return MP_OK;
    int rc, longest = 0;
    if ((rc = check_primers(c, n, codes, off, MP_DIMER_MAX_LEN, &longest))) return rc;
    if ((rc = ensure_dimer_tables(c, loss_hit, dg))) return rc;
    const size_t total = (size_t)off[n];
    Scratch sc(c);
    uint8_t *d_codes = nullptr;
    int32_t *d_off = nullptr, *d_hits = nullptr;
    unsigned long long *d_n = nullptr;
    if ((rc = sc.alloc(&d_codes, total)) || (rc = sc.alloc(&d_off, (size_t)n + 1)) || (rc = sc.alloc(&d_hits, (size_t)cap * 6)) ||
        (rc = sc.alloc(&d_n, 1))) return rc;
    HIPCK(c, hipMemcpyAsync(d_codes, codes, total, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_off, off, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemsetAsync(d_n, 0, sizeof(unsigned long long), c->stream));
    DimerArgs da{d_codes, d_off, n, mode, n_new, c->dm_loss, c->dm_dg, dg_limit, (long long)cap, d_hits, d_n};
    // pairs the scan really visits: mode 1 only those touching one of the n_new new primers
    const long long all = (long long)n * n;
    const long long visited = mode == 0 ? all / 2 + n : (long long)n_new * (2LL * n - n_new);
    const int G = lanes_per_pair(visited, longest);
    PrimRec<false> *d_prim = nullptr;                  // (function scope: Scratch frees through the variables' addresses)
    PrimRec<true> *d_priml = nullptr;
    if (G == 1 && longest > 32) {
        if ((rc = sc.alloc(&d_priml, (size_t)n))) return rc;
        hipLaunchKernelGGL(prim_kernel<true>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, d_codes, d_off, (int)n, d_priml);
        hipLaunchKernelGGL(dimer_rows_kernel<true>, dim3((unsigned)n), dim3(kBlock), 0, c->stream, da, (const PrimRec<true> *)d_priml);
    } else if (G == 1) {
        if ((rc = sc.alloc(&d_prim, (size_t)n))) return rc;
        hipLaunchKernelGGL(prim_kernel<false>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, d_codes, d_off, (int)n, d_prim);
        hipLaunchKernelGGL(dimer_rows_kernel<false>, dim3((unsigned)n), dim3(kBlock), 0, c->stream, da, (const PrimRec<false> *)d_prim);
    } else {
        const unsigned blocks = (unsigned)std::min<long long>((all + kBlock / G - 1) / (kBlock / G), kGroupGrid);
        if (G == 64) hipLaunchKernelGGL(dimer_group_kernel<64>, dim3(blocks), dim3(kBlock), 0, c->stream, da, all, (const int32_t *)nullptr, (uint8_t *)nullptr, 1);
        else hipLaunchKernelGGL(dimer_group_kernel<16>, dim3(blocks), dim3(kBlock), 0, c->stream, da, all, (const int32_t *)nullptr, (uint8_t *)nullptr, 1);
    }
    HIPCK(c, hipGetLastError());
    unsigned long long nh = 0;
    HIPCK(c, hipMemcpyAsync(&nh, d_n, sizeof(nh), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    size_t ncopy = (size_t)std::min<unsigned long long>(nh, (unsigned long long)cap);
    if (ncopy) HIPCK(c, hipMemcpy(hits, d_hits, sizeof(int32_t) * 6 * ncopy, hipMemcpyDeviceToHost));
    *n_hits = (int64_t)nh;
    return MP_OK;
}


int mp_dimer_pairs(mp_ctx *c, int32_t n, const uint8_t *codes, const int32_t *off, int64_t n_pairs, const int32_t *pairs,
                   const uint8_t *loss_hit, const double *dg, double dg_limit, uint8_t *flags) {
    if (!c) return MP_ERR_ARG;
    if (n < 0 || !codes || !off || !loss_hit || !dg || n_pairs < 0 || (n_pairs && (!pairs || !flags)))
        return fail(c, MP_ERR_ARG, "mp_dimer_pairs: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    if (n_pairs == 0) return MP_OK;
    int rc, longest = 0;
    if ((rc = check_primers(c, n, codes, off, MP_DIMER_MAX_LEN, &longest))) return rc;
    for (int64_t p = 0; p < 2 * n_pairs; p++)
        if (pairs[p] < 0 || pairs[p] >= n) return fail(c, MP_ERR_ARG, "pair %lld out of range", (long long)(p / 2));
    if ((rc = ensure_dimer_tables(c, loss_hit, dg))) return rc;
    const size_t total = (size_t)off[n];
    Scratch sc(c);
    uint8_t *d_codes = nullptr, *d_flags = nullptr;
    int32_t *d_off = nullptr, *d_pairs = nullptr;
    if ((rc = sc.alloc(&d_codes, total)) || (rc = sc.alloc(&d_off, (size_t)n + 1)) || (rc = sc.alloc(&d_pairs, (size_t)2 * n_pairs)) ||
        (rc = sc.alloc(&d_flags, (size_t)n_pairs))) return rc;
    HIPCK(c, hipMemcpyAsync(d_codes, codes, total, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_off, off, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_pairs, pairs, sizeof(int32_t) * 2 * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    const int G = lanes_per_pair((long long)n_pairs, longest);
    PrimRec<false> *d_prim = nullptr;                  // (function scope: Scratch frees through the variables' addresses)
    PrimRec<true> *d_priml = nullptr;
    if (G == 1 && longest > 32) {
        if ((rc = sc.alloc(&d_priml, (size_t)n))) return rc;
        hipLaunchKernelGGL(prim_kernel<true>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, d_codes, d_off, (int)n, d_priml);
        hipLaunchKernelGGL(dimer_pairs_kernel<true>, dim3((unsigned)((n_pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
                           (const PrimRec<true> *)d_priml, (long long)n_pairs, d_pairs, c->dm_loss, c->dm_dg, dg_limit, d_flags);
    } else if (G == 1) {
        if ((rc = sc.alloc(&d_prim, (size_t)n))) return rc;
        hipLaunchKernelGGL(prim_kernel<false>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, d_codes, d_off, (int)n, d_prim);
        hipLaunchKernelGGL(dimer_pairs_kernel<false>, dim3((unsigned)((n_pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
                           (const PrimRec<false> *)d_prim, (long long)n_pairs, d_pairs, c->dm_loss, c->dm_dg, dg_limit, d_flags);
    } else {
        DimerArgs da{d_codes, d_off, n, 0, 0, c->dm_loss, c->dm_dg, dg_limit, 0, nullptr, nullptr};
        // few pairs: several groups per pair, until the chip's wave slots are covered about twice
        int split = 1;
        if (const char *e = getenv("MP_DIMER_SPLIT")) split = std::max(1, std::min(64, atoi(e)));
        else while (split < 16 && (long long)n_pairs * split * G < 4LL * kGroupGrid * kBlock) split *= 2;
        if (split > 1) HIPCK(c, hipMemsetAsync(d_flags, 0, (size_t)n_pairs, c->stream));
        const long long groups = (long long)n_pairs * split;
        const unsigned blocks = (unsigned)std::min<long long>((groups + kBlock / G - 1) / (kBlock / G), kGroupGrid);
        if (G == 64) hipLaunchKernelGGL(dimer_group_kernel<64>, dim3(blocks), dim3(kBlock), 0, c->stream, da, (long long)n_pairs, (const int32_t *)d_pairs, d_flags, split);
        else hipLaunchKernelGGL(dimer_group_kernel<16>, dim3(blocks), dim3(kBlock), 0, c->stream, da, (long long)n_pairs, (const int32_t *)d_pairs, d_flags, split);
    }
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(flags, d_flags, (size_t)n_pairs, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
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


// the search on device text: the bytes of this call, or the context's resident store (st_code set)
static int pcr_scan_device(mp_ctx *c, const uint8_t *d_bytes, const int64_t *d_roff, const unsigned long long *st_code, const unsigned long long *st_flag,
                           const int64_t *st_woff, int32_t n_rows, int32_t n_pairs, const uint8_t *codes, const int32_t *off, int32_t *out) {
    int rc;
    if ((rc = check_primers(c, 2 * n_pairs, codes, off, MP_PATTERN_MAX_LEN))) return rc;
    const size_t ncodes = (size_t)off[2 * n_pairs];
    const size_t nout = (size_t)n_pairs * (size_t)n_rows * 4;
    Scratch sc(c);
    uint8_t *d_codes = nullptr;
    int32_t *d_off = nullptr, *d_out = nullptr;
    if ((rc = sc.alloc(&d_codes, ncodes)) || (rc = sc.alloc(&d_off, (size_t)2 * n_pairs + 1)) || (rc = sc.alloc(&d_out, nout))) return rc;
    HIPCK(c, hipMemcpyAsync(d_codes, codes, ncodes, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_off, off, sizeof(int32_t) * ((size_t)2 * n_pairs + 1), hipMemcpyHostToDevice, c->stream));
    // pattern table of the block-per-sequence kernel: every forward expansion and RC(reverse expansion), in expansion order
    static const char *members[16] = {"", "A", "C", "AC", "G", "AG", "GC", "GAC", "T", "AT", "CT", "ATC", "GT", "GAT", "GTC", "ATGC"};
    int longest = 0;
    for (int32_t q = 0; q < 2 * n_pairs; q++) longest = std::max(longest, off[q + 1] - off[q]);
    const bool two_words = longest > 32;            // one primer longer than 32 bases: two-word patterns / character search for the call
    std::vector<PcrPat<2>> pats;                    // built two words wide, narrowed below when one suffices
    std::vector<PcrPair> prs((size_t)n_pairs);
    bool fits = !getenv("MP_PCR_ROLLING");
    for (int32_t p = 0; p < n_pairs && fits; p++) {
        for (int side = 0; side < 2; side++) {
            const int a0 = off[2 * p + side], L = off[2 * p + side + 1] - a0;
            long long d = 1;
            for (int j = 0; j < L; j++) d *= (long long)strlen(members[codes[a0 + j]]);
            if ((long long)pats.size() + d > 4096) { fits = false; break; }
            if (side == 0) { prs[(size_t)p].f0 = (int32_t)pats.size(); prs[(size_t)p].nf = (int32_t)d; }
            else { prs[(size_t)p].r0 = (int32_t)pats.size(); prs[(size_t)p].nr = (int32_t)d; }
            for (long long e = 0; e < d; e++) {                      // expansion e: last position fastest (itertools.product)
                int base[MP_PATTERN_MAX_LEN];
                long long idx = e;
                for (int j = L - 1; j >= 0; j--) {
                    const char *m = members[codes[a0 + j]];
                    const int sz = (int)strlen(m);
                    const char ch = m[idx % sz];
                    idx /= sz;
                    base[j] = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3;
                }
                PcrPat<2> P{};
                for (int j = 0; j < L; j++) {
                    const int code = side == 0 ? base[j] : 3 - base[L - 1 - j];     // the text reads RC(reverse expansion)
                    P.word[j >> 5] |= (unsigned long long)code << (2 * (j & 31));
                    P.lenmask[j >> 5] |= 1ull << (2 * (j & 31));
                }
                P.len = L;
                pats.push_back(P);
            }
        }
    }
    std::vector<PcrPat<1>> narrow;
    if (fits && !two_words)
        for (const PcrPat<2> &P : pats) narrow.push_back(PcrPat<1>{{P.word[0]}, {P.lenmask[0]}, P.len, 0});
    const size_t pat_bytes = two_words ? sizeof(PcrPat<2>) * pats.size() : sizeof(PcrPat<1>) * narrow.size();
    uint8_t *d_pats = nullptr;
    PcrPair *d_prs = nullptr;
    if (fits && ((rc = sc.alloc(&d_pats, pat_bytes)) || (rc = sc.alloc(&d_prs, prs.size())))) fits = false;
    if (fits) {
        HIPCK(c, hipMemcpyAsync(d_pats, two_words ? (const void *)pats.data() : (const void *)narrow.data(), pat_bytes, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(d_prs, prs.data(), sizeof(PcrPair) * prs.size(), hipMemcpyHostToDevice, c->stream));
        int shortest = MP_PATTERN_MAX_LEN;
        for (int32_t q = 0; q < 2 * n_pairs; q++) shortest = std::min(shortest, off[q + 1] - off[q]);
        const int prefilter = shortest >= 8 && !getenv("MP_PCR_NO_PREFILTER") ? 1 : 0;
#define MP_PCR_LAUNCH(NW, RES, NP)                                                                                                             \
    hipLaunchKernelGGL((pcr_block_kernel<NW, RES>), dim3((unsigned)n_rows), dim3(kBlock), 0, c->stream, d_bytes, d_roff, n_rows, st_code, st_flag, \
                       st_woff, reinterpret_cast<const PcrPat<NW> *>(d_pats), (int)(NP), (const PcrPair *)d_prs, n_pairs, d_codes, d_off, d_out, prefilter)
        if (st_code) { if (two_words) MP_PCR_LAUNCH(2, true, pats.size()); else MP_PCR_LAUNCH(1, true, narrow.size()); }
        else { if (two_words) MP_PCR_LAUNCH(2, false, pats.size()); else MP_PCR_LAUNCH(1, false, narrow.size()); }
#undef MP_PCR_LAUNCH
    } else if (two_words) {
        hipLaunchKernelGGL(pcr_kernel<true>, dim3((unsigned)((n_rows + kBlock - 1) / kBlock), (unsigned)n_pairs), dim3(kBlock), 0, c->stream,
                           d_bytes, d_roff, n_rows, d_codes, d_off, d_out);
    } else {
        hipLaunchKernelGGL(pcr_kernel<false>, dim3((unsigned)((n_rows + kBlock - 1) / kBlock), (unsigned)n_pairs), dim3(kBlock), 0, c->stream,
                           d_bytes, d_roff, n_rows, d_codes, d_off, d_out);
    }
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(out, d_out, sizeof(int32_t) * nout, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

int mp_pcr_scan(mp_ctx *c, const uint8_t *bytes, const int64_t *row_off, int32_t n_rows, int32_t n_pairs,
                const uint8_t *codes, const int32_t *off, int32_t *out) {
    if (!c) return MP_ERR_ARG;
    if (n_rows < 0 || n_pairs < 0 || (n_rows && (!bytes || !row_off)) || (n_pairs && (!codes || !off)) || (n_rows && n_pairs && !out))
        return fail(c, MP_ERR_ARG, "mp_pcr_scan: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    if (n_rows == 0 || n_pairs == 0) return MP_OK;
    const size_t total = (size_t)(row_off[n_rows] - row_off[0]);
    Scratch sc(c);
    uint8_t *d_bytes = nullptr;
    int64_t *d_roff = nullptr;
    int rc;
    if ((rc = sc.alloc(&d_bytes, total + 16)) || (rc = sc.alloc(&d_roff, (size_t)n_rows + 1))) return rc;
    std::vector<int64_t> roff((size_t)n_rows + 1);
    for (int32_t r = 0; r <= n_rows; r++) roff[(size_t)r] = row_off[r] - row_off[0];
    HIPCK(c, hipMemcpyAsync(d_bytes, bytes + row_off[0], total, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(d_roff, roff.data(), sizeof(int64_t) * roff.size(), hipMemcpyHostToDevice, c->stream));
    rc = pcr_scan_device(c, d_bytes, d_roff, nullptr, nullptr, nullptr, n_rows, n_pairs, codes, off, out);
    (void)hipStreamSynchronize(c->stream);               // (roff leaves scope)
    return rc;
}

int mp_pcr_scan_resident(mp_ctx *c, int32_t n_pairs, const uint8_t *codes, const int32_t *off, int32_t *out) {
    if (!c) return MP_ERR_ARG;
    if (n_pairs < 0 || (n_pairs && (!codes || !off)) || (c->sq_n && n_pairs && !out)) return fail(c, MP_ERR_ARG, "mp_pcr_scan_resident: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    if (c->sq_n == 0 || n_pairs == 0) return MP_OK;
    return pcr_scan_device(c, c->sq_bytes, c->sq_roff, c->sq_code, c->sq_flag, c->sq_woff, c->sq_n, n_pairs, codes, off, out);
}

}  // extern "C"
