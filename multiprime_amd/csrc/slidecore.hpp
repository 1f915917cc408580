// slidecore.hpp — the band routine of the sliding evaluation (see slideplan.hpp for the idea), written once for two environments:
//   * evalslide.hip: one wave of the GPU kernel (Env = buffer loads, an LDS ring, DPP wave sums);
//   * tools/slide_emul.cpp: one "lane" at a time on the CPU (Env = plain arrays) — the same arithmetic, the same host-written plan,
//     checked against brute force without a GPU.
// Everything here is per-lane arithmetic on GW 32-bit words (32 sequences each); what is wave-uniform comes through env.u*().
#pragma once

#include <cstdint>

#include "slideplan.hpp"

#if defined(__HIPCC__)
#define SLIDE_HD __host__ __device__ __forceinline__
#else
#define SLIDE_HD inline
#endif

namespace mp {

// D = f(a, b, c) bit by bit, f given by its truth table (bit (a << 2 | b << 1 | c) of LUT): one v_bitop3_b32 on gfx950
template <int LUT>
SLIDE_HD uint32_t bop(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, LUT);
#else
    uint32_t r = 0;
    if (LUT & 0x01) r |= ~a & ~b & ~c;
    if (LUT & 0x02) r |= ~a & ~b & c;
    if (LUT & 0x04) r |= ~a & b & ~c;
    if (LUT & 0x08) r |= ~a & b & c;
    if (LUT & 0x10) r |= a & ~b & ~c;
    if (LUT & 0x20) r |= a & ~b & c;
    if (LUT & 0x40) r |= a & b & ~c;
    if (LUT & 0x80) r |= a & b & c;
    return r;
#endif
}
constexpr int slide_lut(bool (*f)(bool, bool, bool)) {
    int t = 0;
    for (int i = 0; i < 8; i++)
        if (f((i >> 2) & 1, (i >> 1) & 1, i & 1)) t |= 1 << i;
    return t;
}
constexpr int kSlXor3 = slide_lut([](bool a, bool b, bool c) { return (a != b) != c; });
constexpr int kSlMaj = slide_lut([](bool a, bool b, bool c) { return (a && b) || (a && c) || (b && c); });
constexpr int kSlProp = slide_lut([](bool p, bool ci, bool dir) { return p && (ci == dir); });          // p & ~(ci ^ dir)
constexpr int kSlBorrow = slide_lut([](bool ci, bool s, bool br) { return (!ci && s) || (!ci && br) || (s && br); });   // maj(~ci, s, br)
constexpr int kSlAndNot = slide_lut([](bool a, bool b, bool) { return a && !b; });
constexpr int kSlOr3 = slide_lut([](bool a, bool b, bool c) { return a || b || c; });
constexpr int kSlOrAnd = slide_lut([](bool a, bool b, bool c) { return a || (b && c); });                // a | (b & c)
constexpr int kSlAndNotNot = slide_lut([](bool a, bool b, bool c) { return a && !b && !c; });

SLIDE_HD int slide_popc(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}

// 5-bit bit-sliced counters (k <= 31 columns): c += up - down for two disjoint one-bit planes given as (changed, direction):
// x = rows that change, dir = rows that go UP among them (rows of x outside dir go down).  10 instructions per word.
struct SlideCount { uint32_t b0, b1, b2, b3, b4; };
SLIDE_HD void slide_updown(SlideCount &c, uint32_t x, uint32_t dir) {
    uint32_t p = x;                                    // carry (dir) / borrow (~dir) into the next bit
    uint32_t n = c.b0 ^ p; p = bop<kSlProp>(p, c.b0, dir); c.b0 = n;
    n = c.b1 ^ p; p = bop<kSlProp>(p, c.b1, dir); c.b1 = n;
    n = c.b2 ^ p; p = bop<kSlProp>(p, c.b2, dir); c.b2 = n;
    n = c.b3 ^ p; p = bop<kSlProp>(p, c.b3, dir); c.b3 = n;
    c.b4 ^= p;
}

struct SlideArgs {
    const SlideBand *bands;
    const uint32_t *iters;
    const uint32_t *recs;
    int k, p0, ns;
    uint32_t spos, fmask, rmask;
};

// One band.  Env provides (all arrays are [GW] words of this lane):
//   uband(b) -> SlideBand, uiter(idx) -> uint32_t, urec(item, dword) -> uint32_t      wave-uniform values
//   fetch(plane_row, d)                the lane's words of column plane row `plane_row` (= column * 4 + base)
//   valid_of(window, v)                rows the column-plane pass may count for this window
//   ring_swap(slot, in, out, have_old) out = ring[slot] (0 when !have_old); ring[slot] = in
//   ring_read(slot, out)
//   commit(item_in_band, item, acc)    the lane's packed counts of the item's 8 member slots (perfect | forward << 10 | reverse << 20)
template <int LV, int GW, class Env>
SLIDE_HD void slide_band(Env &env, const SlideArgs &A, int band_index) {
    static_assert(LV >= 1 && LV <= 4 && 32 * GW < 1024, "three 10-bit counts per register");
    const SlideBand bd = env.uband(band_index);
    const int k = A.k;
    SlideCount cnt[GW];
#pragma unroll
    for (int i = 0; i < GW; i++) cnt[i] = SlideCount{0u, 0u, 0u, 0u, 0u};
    int slot = 0;
#pragma unroll 1
    for (int t = -(k - 1); t < bd.n_win; t++) {
        const uint32_t it0 = env.uiter(bd.iter0 + 2 * (t + k - 1)), it1 = env.uiter(bd.iter0 + 2 * (t + k - 1) + 1);
        uint32_t bn[GW], bo[GW];
        env.fetch(it0, bn);
#pragma unroll
        for (int i = 0; i < GW; i++) bn[i] = ~bn[i];                   // rows that do not carry the reference base here
        env.ring_swap(slot, bn, bo, t >= 1);                           // the column sliding out shares the slot (k columns apart)
#pragma unroll
        for (int i = 0; i < GW; i++) slide_updown(cnt[i], bn[i] ^ bo[i], bn[i]);
        const int slot_now = slot;
        slot = slot + 1 == k ? 0 : slot + 1;
        const int n_items = (int)(it1 >> 24);
        if (t < 0 || n_items == 0) continue;
        const int win = bd.w0 + t;
        // mismatch words of the reference at the strict positions of this window, out of the ring
        uint32_t sv[kSlideStrict][GW];
#pragma unroll
        for (int q = 0; q < kSlideStrict; q++) {
            if (q < A.ns) {
                int s = slot_now + 1 + (int)((A.spos >> (5 * q)) & 31u);
                if (s >= k) s -= k;
                env.ring_read(s, sv[q]);
            } else {
#pragma unroll
                for (int i = 0; i < GW; i++) sv[q][i] = 0u;
            }
        }
        uint32_t valid[GW];
        env.valid_of(win, valid);
        const int first = (int)(it1 & 0xFFFFFFu);
#pragma unroll 1
        for (int ii = 0; ii < n_items; ii++) {
            const int item = bd.item0 + first + ii;
            const uint32_t hdr = env.urec(item, 0);
            const int n_slots = (int)(hdr & 15u), n_extra = (int)((hdr >> 8) & 15u);
            const uint32_t row0 = (uint32_t)(A.p0 + win) * 4u;
            // (a) the event planes, once, into registers
            uint32_t en[kSlideKept + 1], d[kSlideKept + 1][GW];
#pragma unroll
            for (int s = 1; s <= kSlideKept; s++) {
                en[s] = s < n_slots ? env.urec(item, s) : 0u;
                if (en[s] & kSlPresent) env.fetch(row0 + (en[s] & 127u), d[s]);
                else {
#pragma unroll
                    for (int i = 0; i < GW; i++) d[s][i] = 0u;
                }
            }
            // (b) mismatch count of the most degenerate member: the reference's count minus the planes of the bases it accepts beyond
            // the reference — a 7-input carry-save sum, then a 5-bit minus 3-bit subtraction
            uint32_t t1[GW], t2[GW], t3[GW], t4[GW], sf[GW], sr[GW];
            SlideCount c0[GW];
#pragma unroll
            for (int i = 0; i < GW; i++) {
                uint32_t e[kSlideKept + 1];
#pragma unroll
                for (int s = 1; s <= kSlideKept; s++) e[s] = (en[s] & kSlSub) ? d[s][i] : 0u;
                const uint32_t sa = bop<kSlXor3>(e[1], e[2], e[3]), ca = bop<kSlMaj>(e[1], e[2], e[3]);
                const uint32_t sb = bop<kSlXor3>(e[4], e[5], e[6]), cb = bop<kSlMaj>(e[4], e[5], e[6]);
                const uint32_t s0 = bop<kSlXor3>(sa, sb, e[7]), cc = bop<kSlMaj>(sa, sb, e[7]);
                const uint32_t s1 = bop<kSlXor3>(ca, cb, cc), s2 = bop<kSlMaj>(ca, cb, cc);
                const SlideCount c = cnt[i];
                SlideCount r;
                r.b0 = c.b0 ^ s0;
                uint32_t br = bop<kSlAndNot>(s0, c.b0, 0u);
                r.b1 = bop<kSlXor3>(c.b1, s1, br); br = bop<kSlBorrow>(c.b1, s1, br);
                r.b2 = bop<kSlXor3>(c.b2, s2, br); br = bop<kSlBorrow>(c.b2, s2, br);
                r.b3 = c.b3 ^ br; br = bop<kSlAndNot>(br, c.b3, 0u);
                r.b4 = c.b4 ^ br;
                c0[i] = r;
            }
            // corrections that are not events (rare): one plane each, straight onto the count
#pragma unroll 1
            for (int x = 0; x < n_extra; x++) {
                const uint32_t ex = env.urec(item, 16 + x);
                uint32_t pl[GW];
                env.fetch(row0 + (ex & 127u), pl);
                const uint32_t up = (ex & kSlSub) ? 0u : 0xFFFFFFFFu;
#pragma unroll
                for (int i = 0; i < GW; i++) slide_updown(c0[i], pl[i], up);
            }
            // saturating thermometer counters of the walk: t_j = "at least j mismatches"
#pragma unroll
            for (int i = 0; i < GW; i++) {
                const SlideCount c = c0[i];
                const uint32_t hi = bop<kSlOr3>(c.b2, c.b3, c.b4);
                t1[i] = bop<kSlOr3>(c.b0, c.b1, hi);
                t2[i] = c.b1 | hi;
                t3[i] = bop<kSlOrAnd>(hi, c.b1, c.b0);
                t4[i] = hi;
            }
            // strict positions of the most degenerate member: the reference's mismatch word there, minus the rows a SUB plane of that
            // position takes back (they carry a base the member accepts)
            const uint32_t sm_lo = env.urec(item, 24), sm_hi = env.urec(item, 25);
#pragma unroll
            for (int i = 0; i < GW; i++) sf[i] = sr[i] = 0u;
#pragma unroll
            for (int q = 0; q < kSlideStrict; q++) {
                if (q >= A.ns) break;
                const uint32_t sm = ((q < 4 ? sm_lo : sm_hi) >> (8 * (q & 3))) & 255u;
                uint32_t v[GW];
#pragma unroll
                for (int i = 0; i < GW; i++) v[i] = sv[q][i];
                if (sm) {
#pragma unroll
                    for (int s = 1; s <= kSlideKept; s++)
                        if ((sm >> s) & 1u) {
#pragma unroll
                            for (int i = 0; i < GW; i++) v[i] ^= d[s][i];
                        }
                }
                const uint32_t fF = ((A.fmask >> q) & 1u) ? 0xFFFFFFFFu : 0u, fR = ((A.rmask >> q) & 1u) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    sf[i] = bop<kSlOrAnd>(sf[i], v[i], fF);
                    sr[i] = bop<kSlOrAnd>(sr[i], v[i], fR);
                }
            }
            // (c) walk down the chain: event plane s, then member slot s is counted
            uint32_t acc[8];
#pragma unroll
            for (int s = 0; s < 8; s++) {
                acc[s] = 0u;
                if (s >= n_slots) continue;
                if (s > 0 && (en[s] & kSlPresent)) {
#pragma unroll
                    for (int i = 0; i < GW; i++) {
                        if (LV >= 4) t4[i] = bop<kSlOrAnd>(t4[i], t3[i], d[s][i]);
                        if (LV >= 3) t3[i] = bop<kSlOrAnd>(t3[i], t2[i], d[s][i]);
                        if (LV >= 2) t2[i] = bop<kSlOrAnd>(t2[i], t1[i], d[s][i]);
                        t1[i] |= d[s][i];
                    }
                    if (en[s] & (kSlStrictF | kSlStrictR)) {
                        const uint32_t fF = (en[s] & kSlStrictF) ? 0xFFFFFFFFu : 0u, fR = (en[s] & kSlStrictR) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                        for (int i = 0; i < GW; i++) {
                            sf[i] = bop<kSlOrAnd>(sf[i], d[s][i], fF);
                            sr[i] = bop<kSlOrAnd>(sr[i], d[s][i], fR);
                        }
                    }
                }
                if ((int32_t)env.urec(item, 8 + s) < 0) continue;      // a slot in the middle of a step: no member to count
                uint32_t nP = 0, nF = 0, nR = 0;
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    const uint32_t far = LV == 1 ? t1[i] : (LV == 2 ? t2[i] : (LV == 3 ? t3[i] : t4[i]));
                    nP += (uint32_t)slide_popc(bop<kSlAndNot>(valid[i], t1[i], 0u));
                    nF += (uint32_t)slide_popc(bop<kSlAndNotNot>(valid[i], far, sf[i]));
                    nR += (uint32_t)slide_popc(bop<kSlAndNotNot>(valid[i], far, sr[i]));
                }
                acc[s] = nP | (nF << 10) | (nR << 20);
            }
            env.commit(first + ii, item, acc);
        }
    }
}

}  // namespace mp
