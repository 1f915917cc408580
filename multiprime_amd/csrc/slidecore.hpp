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

// the side of a wave-uniform branch that is laid out out of line: a taken branch costs a wave a refill of its instruction buffer, the
// fall-through nothing — the usual case goes straight on
#ifndef SLIDE_UNLIKELY
#define SLIDE_UNLIKELY(x) __builtin_expect(!!(x), 0)
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
constexpr int kSlOrAndNot = slide_lut([](bool a, bool b, bool c) { return a || (b && !c); });            // a | (b & ~c)
constexpr int kSlAndNotNot = slide_lut([](bool a, bool b, bool c) { return a && !b && !c; });
constexpr int kSlOrNot = slide_lut([](bool a, bool b, bool) { return a || !b; });                         // a | ~b
constexpr int kSlOrOrNot = slide_lut([](bool a, bool b, bool c) { return a || b || !c; });               // a | b | ~c
constexpr int kSlXorAndNot = slide_lut([](bool a, bool b, bool c) { return a != (b && !c); });         // a ^ (b & ~c)

SLIDE_HD int slide_popc(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}

// keeps a value's computation where it is written: the compiler otherwise sinks the popcounts of a member slot behind the branches of the
// later events and carries the slot's three sets along until then (48 more registers at two words per lane)
SLIDE_HD void slide_pin(uint32_t &x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#else
    (void)x;
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
    uint32_t row_scale;                // plane row -> what Env::fetch takes (bytes of a plane row on the GPU, 1 in the emulation)
    uint32_t fpos, rpos;               // FAST strict form (slide_strict_lists): up to three forward / reverse strict positions, 5 bits each, their number << 15
};

// The strict positions of a launch as two lists (forward, reverse) of at most three positions each — what `-c` gives by default (V20:85:
// "1,2,-1").  False when a side has more: the launch keeps the per-position form.
inline bool slide_strict_lists(int k, uint32_t sF, uint32_t sR, uint32_t &fpos, uint32_t &rpos) {
    fpos = rpos = 0u;
    uint32_t nf = 0, nr = 0;
    for (int j = 0; j < k && j < 32; j++) {
        if ((sF >> j) & 1u) { if (nf == 3) return false; fpos |= (uint32_t)j << (5 * nf++); }
        if ((sR >> j) & 1u) { if (nr == 3) return false; rpos |= (uint32_t)j << (5 * nr++); }
    }
    fpos |= nf << 15; rpos |= nr << 15;
    return true;
}

// The planes an item needs beyond the sliding count: its event planes (entries 1 .. n_slots - 1) and the rows its window lets the
// column-plane pass count.  Requested one item AHEAD of their use (slide_band), so that an item's memory latency hides behind the
// arithmetic of the item before it.
template <int GW>
struct SlideFetch {
    uint32_t d[kSlideKept + 1][GW];     // d[0] is unused
    uint32_t valid[GW];
};

// Always the same requests (with USE_VALID the rows the window counts, then seven planes; a slot the item does not have reads the all-zero row
// behind the last column): no branch, so the loads stay in flight behind the item that is being computed.  The record holds what a
// fetch takes as it stands.
template <int GW, bool USE_VALID, class Env>
SLIDE_HD void slide_request(Env &env, const typename Env::Rec &rec, SlideFetch<GW> &F) {
    if (USE_VALID) env.valid_of(env.rec_word(rec, 29), F.valid);
#pragma unroll
    for (int s = 1; s <= kSlideKept; s++) env.fetch_event(env.rec_word(rec, s), F.d[s]);
}

// One item: all eight member slots, straight-line (a slot the item does not have repeats the counts of the one before it — its plane
// is the all-zero row — and reports to nobody).  SIMPLE (the host's flag; every chain of a refinement run has it): every event plane is
// the plane of a base beyond the reference — no per-plane masks in the carry-save sum.
// FAST (simple items of a launch with at most three strict positions per side): `sv` does not hold the reference's mismatch words position by
// position but, per side, their SUM over the side's strict positions as a two-bit bit-sliced count (sv[0], sv[1]: forward; sv[2], sv[3]:
// reverse; slide_band computes them once per window).  An event at a strict position is the plane of a base the most degenerate member
// accepts there: its rows mismatch the reference at that position (they are in the count) and must not — one decrement of the count, two
// instructions per word and side, for the flagged events only.  A row is out by its strict positions when the count is not zero.  (The
// per-position form builds, per strict position with an event, the union of its event planes by seven masked ORs: with seven events per
// item and six strict positions of eighteen, two positions of six take that path — ~52 vector instructions per item against ~17 here.)
template <int LV, int GW, bool SIMPLE, bool USE_VALID, bool NO_EXTRA, bool FAST, class Env>
SLIDE_HD void slide_item(Env &env, const SlideArgs &A, const typename Env::Rec &rec, uint32_t hdr, const SlideCount (&cnt)[GW],
                         const uint32_t (&sv)[kSlideStrict][GW], const SlideFetch<GW> &F, uint32_t (&accPF)[8], uint32_t (&accR)[4]) {
    const int n_extra = (int)((hdr >> 8) & 15u);
    const uint32_t flags = env.rec_word(rec, 28);
    // (b) mismatch count of the most degenerate member: the reference's count minus the planes of the bases it accepts beyond
    // the reference — a 7-input carry-save sum, then a 5-bit minus 3-bit subtraction
    SlideCount c0[GW];
    uint32_t sub[kSlideKept + 1];
#pragma unroll
    for (int s = 1; s <= kSlideKept; s++) sub[s] = SIMPLE ? 0xFFFFFFFFu : (uint32_t)((int32_t)(flags << (15 - s)) >> 31);      // bit 16 + s
#pragma unroll
    for (int i = 0; i < GW; i++) {
        uint32_t e[kSlideKept + 1];
#pragma unroll
        for (int s = 1; s <= kSlideKept; s++) e[s] = SIMPLE ? F.d[s][i] : (F.d[s][i] & sub[s]);
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
    // corrections that are not events (rare; a plan of simple items only has none — NO_EXTRA: the loop and the copies of the count
    // around it are not even compiled): one plane each, straight onto the count
    if (!NO_EXTRA && n_extra) {
        const uint32_t row0 = env.rec_word(rec, 30);
#pragma unroll 1
        for (int x = 0; x < n_extra; x++) {
            const uint32_t ex = env.rec_word_dyn(rec, 16 + x);
            uint32_t pl[GW];
            env.fetch(row0 + (ex & 127u) * A.row_scale, pl);
            const uint32_t up = (ex & kSlSub) ? 0u : 0xFFFFFFFFu;
#pragma unroll
            for (int i = 0; i < GW; i++) slide_updown(c0[i], pl[i], up);
        }
    }
    // Saturating thermometer counters of the walk, kept as "rows that are OUT": T[j] = "at least j mismatches" (T[1] also holds the
    // rows the window does not count at all), DF / DR = "not a forward / reverse hit": more than v mismatches, a mismatch at a strict
    // position, or not counted.  All of them only grow along the chain, so an event plane is one instruction per word and set, and what
    // is counted per member slot is the OUT rows — the flush turns them round (perfect = rows - out1, forward = out1 - outF).
    uint32_t T[4][GW], DF[GW], DR[GW];
    const uint32_t sm_lo = env.rec_word(rec, 24), sm_hi = env.rec_word(rec, 25);
#pragma unroll
    for (int i = 0; i < GW; i++) DF[i] = DR[i] = 0u;
    // strict positions of the most degenerate member: the reference's mismatch word there, minus the rows a SUB plane of that
    // position takes back (they carry a base the member accepts)
    // (DF |= word & ~(x | ~fF): with no event plane at the position — the usual case — x is nothing and ~fF a launch constant in a
    // scalar register, so a position costs one instruction per word and set, and the ring's words are never copied)
    uint32_t cF0[GW], cF1[GW], cR0[GW], cR1[GW];
    if (FAST) {
        static_assert(!FAST || SIMPLE, "the two-bit strict counts take back SUB planes only");
#pragma unroll
        for (int i = 0; i < GW; i++) { cF0[i] = sv[0][i]; cF1[i] = sv[1][i]; cR0[i] = sv[2][i]; cR1[i] = sv[3][i]; }
#pragma unroll
        for (int s = 1; s <= kSlideKept; s++) {
            if (SLIDE_UNLIKELY(((flags >> s) & 0x101u) != 0u)) {
                if ((flags >> s) & 1u) {
#pragma unroll
                    for (int i = 0; i < GW; i++) { cF1[i] = bop<kSlXorAndNot>(cF1[i], F.d[s][i], cF0[i]); cF0[i] ^= F.d[s][i]; }
                }
                if ((flags >> (8 + s)) & 1u) {
#pragma unroll
                    for (int i = 0; i < GW; i++) { cR1[i] = bop<kSlXorAndNot>(cR1[i], F.d[s][i], cR0[i]); cR0[i] ^= F.d[s][i]; }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < kSlideStrict; q++) {
        if (FAST) break;
        if (q >= 4 && A.ns <= 4) break;
        const uint32_t sm = ((q < 4 ? sm_lo : sm_hi) >> (8 * (q & 3))) & 255u;
        const uint32_t nfF = ~(uint32_t)((int32_t)(A.fmask << (31 - q)) >> 31), nfR = ~(uint32_t)((int32_t)(A.rmask << (31 - q)) >> 31);
        if (SLIDE_UNLIKELY(sm != 0u)) {
            uint32_t x[GW];
#pragma unroll
            for (int i = 0; i < GW; i++) x[i] = 0u;
#pragma unroll
            for (int s = 1; s <= kSlideKept; s++) {
                const uint32_t m = (uint32_t)((int32_t)(sm << (31 - s)) >> 31);
#pragma unroll
                for (int i = 0; i < GW; i++) x[i] = bop<kSlOrAnd>(x[i], F.d[s][i], m);
            }
#pragma unroll
            for (int i = 0; i < GW; i++) {
                DF[i] = bop<kSlOrAndNot>(DF[i], sv[q][i], x[i] | nfF);
                DR[i] = bop<kSlOrAndNot>(DR[i], sv[q][i], x[i] | nfR);
            }
        } else {
#pragma unroll
            for (int i = 0; i < GW; i++) {
                DF[i] = bop<kSlOrAndNot>(DF[i], sv[q][i], nfF);
                DR[i] = bop<kSlOrAndNot>(DR[i], sv[q][i], nfR);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < GW; i++) {
        const SlideCount c = c0[i];
        const uint32_t hi = bop<kSlOr3>(c.b2, c.b3, c.b4);
        const uint32_t t1 = bop<kSlOr3>(c.b0, c.b1, hi), t2 = c.b1 | hi, t3 = bop<kSlOrAnd>(hi, c.b1, c.b0);
        const uint32_t far = LV == 1 ? t1 : (LV == 2 ? t2 : (LV == 3 ? t3 : hi));
        T[0][i] = USE_VALID ? bop<kSlOrNot>(t1, F.valid[i], 0u) : t1;   // t1 | ~valid
        T[1][i] = t2; T[2][i] = t3; T[3][i] = hi;
        if (FAST && !USE_VALID) {
            DF[i] = bop<kSlOr3>(cF0[i], cF1[i], far);
            DR[i] = bop<kSlOr3>(cR0[i], cR1[i], far);
        } else {
            if (FAST) { DF[i] = cF0[i] | cF1[i]; DR[i] = cR0[i] | cR1[i]; }
            DF[i] = USE_VALID ? bop<kSlOrOrNot>(DF[i], far, F.valid[i]) : (DF[i] | far);
            DR[i] = USE_VALID ? bop<kSlOrOrNot>(DR[i], far, F.valid[i]) : (DR[i] | far);
        }
    }
    // (c) walk down the chain: event plane s, then member slot s is counted.  Counts leave in the layout the wave sums want:
    // accPF[s] = out1 | outF << 16, accR[s / 2] = outR of an even slot | outR of the odd one << 16
#pragma unroll
    for (int s = 0; s < 8; s++) {
        if (s > 0) {
            // one more mismatch puts a row OUT when it already has v of them (T[LV - 2]; any row when v = 0) or the position is strict:
            // DF |= d & (T | strict), with the strict flag of the event as an all-ones / all-zeros scalar — no select, no branch
            // (most events are at no strict position: a uniform branch keeps their two masks and two ORs per word out of the way)
#ifndef SLIDE_WALK_BRANCH
#define SLIDE_WALK_BRANCH 2
#endif
            if (SLIDE_WALK_BRANCH == 2) {
                // the count-based update for every event; an event at a strict position (the minority: a uniform branch) puts its rows out
                // whatever their count — two more instructions per word for those events only, no masks for the others
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    const uint32_t d = F.d[s][i];
                    if (LV == 1) { DF[i] |= d; DR[i] |= d; }
                    else {
                        DF[i] = bop<kSlOrAnd>(DF[i], d, T[LV >= 2 ? LV - 2 : 0][i]);
                        DR[i] = bop<kSlOrAnd>(DR[i], d, T[LV >= 2 ? LV - 2 : 0][i]);
                    }
                }
                if (LV >= 2 && SLIDE_UNLIKELY(((flags >> s) & 0x101u) != 0u)) {
                    const uint32_t mF = (uint32_t)((int32_t)(flags << (31 - s)) >> 31), mR = (uint32_t)((int32_t)(flags << (23 - s)) >> 31);
#pragma unroll
                    for (int i = 0; i < GW; i++) {
                        DF[i] = bop<kSlOrAnd>(DF[i], F.d[s][i], mF);
                        DR[i] = bop<kSlOrAnd>(DR[i], F.d[s][i], mR);
                    }
                }
            } else if (LV >= 2 && (!SLIDE_WALK_BRANCH || ((flags >> s) & 0x101u))) {
                const uint32_t mF = (uint32_t)((int32_t)(flags << (31 - s)) >> 31), mR = (uint32_t)((int32_t)(flags << (23 - s)) >> 31);
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    const uint32_t d = F.d[s][i], t = T[LV >= 2 ? LV - 2 : 0][i];
                    DF[i] = bop<kSlOrAnd>(DF[i], d, t | mF);
                    DR[i] = bop<kSlOrAnd>(DR[i], d, t | mR);
                }
            } else {
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    const uint32_t d = F.d[s][i];
                    if (LV == 1) { DF[i] |= d; DR[i] |= d; }
                    else {
                        DF[i] = bop<kSlOrAnd>(DF[i], d, T[LV >= 2 ? LV - 2 : 0][i]);
                        DR[i] = bop<kSlOrAnd>(DR[i], d, T[LV >= 2 ? LV - 2 : 0][i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < GW; i++) {
                const uint32_t d = F.d[s][i];
                if (LV >= 4) T[2][i] = bop<kSlOrAnd>(T[2][i], T[1][i], d);
                if (LV >= 3) T[1][i] = bop<kSlOrAnd>(T[1][i], T[0][i], d);
                T[0][i] |= d;
            }
        }
        uint32_t nP = 0, nF = 0, nR = 0;
#pragma unroll
        for (int i = 0; i < GW; i++) {
            nP += (uint32_t)slide_popc(T[0][i]);
            nF += (uint32_t)slide_popc(DF[i]);
            nR += (uint32_t)slide_popc(DR[i]);
        }
        accPF[s] = nP | (nF << 16);
        if (s & 1) accR[s >> 1] |= nR << 16;
        else accR[s >> 1] = nR;
        if (SLIDE_WALK_BRANCH) { slide_pin(accPF[s]); slide_pin(accR[s >> 1]); }
    }
}

// Warm-up of a band: N columns slide in at once — their planes are requested TOGETHER, then counted one after the other (nothing slides
// out yet, no window is complete: no items).  The two-iterations-ahead pipeline of the main loop hides a column's latency behind the
// ITEMS of two windows; a warm-up iteration has none, so k - 1 of them one behind the other cost k - 1 half memory round trips — a
// third of the kernel's time at the 131072-row shard, where every wave of the chip warms up at the same moment
// (tools/r05_exp2.sh: 17 columns, ~10 us of 29).
template <int N, int GW, class Env>
SLIDE_HD void slide_warm_chunk(Env &env, SlideCount (&cnt)[GW], int &slot, int j) {
    uint32_t w[N][GW];
#pragma unroll
    for (int u = 0; u < N; u++) env.fetch(env.iter_word(2 * (j + u)), w[u]);
#pragma unroll
    for (int u = 0; u < N; u++) {
        uint32_t bn[GW];
#pragma unroll
        for (int i = 0; i < GW; i++) {
            bn[i] = ~w[u][i];                                           // rows that do not carry the reference base here
            slide_updown(cnt[i], bn[i], bn[i]);                         // all of them go up
        }
        env.ring_write(slot, bn);
        slot++;                                                         // (warm-up columns never wrap: fewer than k of them)
    }
}

// One band.  Env provides (all arrays are [GW] words of this lane):
//   uband(b) -> SlideBand; load_iters(idx) then iter_word(j) = iteration word idx + j, j < 64             wave-uniform values
//   Rec, load_rec(item) -> Rec, rec_word(rec, q) (q a constant), rec_word_dyn(rec, q)                      an item's record
//   fetch(plane_row x row_scale, d)    the lane's words of column plane row `plane_row` (= column * 4 + base)
//   fetch_event(...)                   the same for an item's event planes (one function in every product build)
//   valid_of(window x row_scale, v)    rows the column-plane pass may count for this window
//   ring_zero(k); ring_write(slot, in); ring_read(slot, out)
//   progress(quarter)                  quarters of the band behind the wave (a hint for the issue priority)
//   commit(item_in_band, accPF, accR)  the lane's OUT counts of the item's 8 member slots (layout: slide_item)
// Software pipeline: the column sliding in is requested two iterations ahead, an item's record two items ahead, its planes one item
// ahead — a wave has few neighbours on its SIMD (the ring takes LDS), so it hides its own latencies.  The two register sets of each
// pipeline swap ROLES (two copies of the iteration, two of the item, by parity), never contents: a register copy would wait for the
// youngest load, and so would a branch that picks the set.
// ONLY_SIMPLE: the plan holds simple items without extra corrections only (build_slide_plan's simple_only; the GPU kernel).
// USE_VALID false (the GPU kernel): every row of a window is counted as its plain column slice — rows with more than v gaps or past the
// alignment's end never reach a count anyway (a gap mismatches everything), and what the caller must NOT count as a plain slice (edge-
// gap repaired rows, IUPAC rows) it takes back itself (eval.hip: the subtracting run on the plain-slice planes of the patch list).
// FAST: the strict positions as two-bit counts per side (slide_item); needs ONLY_SIMPLE and A.fpos / A.rpos (slide_strict_lists).
template <int LV, int GW, bool ONLY_SIMPLE, bool USE_VALID, bool FAST = false, class Env>
SLIDE_HD void slide_band(Env &env, const SlideArgs &A, int band_index) {
    static_assert(LV >= 1 && LV <= 4 && GW >= 1 && GW <= 4, "counter levels / words per lane");
    static_assert(!FAST || ONLY_SIMPLE, "FAST strict counts: simple items only");
    const SlideBand bd = env.uband(band_index);
    const int k = A.k;
    SlideCount cnt[GW];
#pragma unroll
    for (int i = 0; i < GW; i++) cnt[i] = SlideCount{0u, 0u, 0u, 0u, 0u};
    env.ring_zero(k);                                                   // a slot's first visitor slides nothing out
    int slot = 0;
    const int n_iter = bd.n_win + k - 1;                                // k - 1 warm-up columns, then one column per window
    const int last_item = bd.item0 + bd.n_items - 1;
    // item pipeline: records of the next two items, planes of the next one (set 0 first)
    typename Env::Rec rec0 = env.load_rec(bd.item0), rec1 = env.load_rec(bd.item0 + 1 <= last_item ? bd.item0 + 1 : last_item);
    SlideFetch<GW> F0, F1;
    slide_request<GW, USE_VALID>(env, rec0, F0);
    int done = 0;                                                       // items of the band behind us
    env.load_iters(bd.iter0);
    env.stamp(2);
    // warm-up: an EVEN number j0 of the band's first k - 1 iterations (they have no items and push nothing out) in chunks of 16, 8,
    // 4, 2 columns in flight; what is left of them (at most one) and everything else runs in the main loop below, from iteration j0.
    // (k - 1 <= 30 iterations: inside the first 64 iteration words.)
    int j0 = 0;
    {
        const int n_warm = k - 1 < n_iter ? k - 1 : n_iter;
        while (n_warm - j0 >= 16) { slide_warm_chunk<16, GW>(env, cnt, slot, j0); j0 += 16; }
        if (n_warm - j0 >= 8) { slide_warm_chunk<8, GW>(env, cnt, slot, j0); j0 += 8; }
        if (n_warm - j0 >= 4) { slide_warm_chunk<4, GW>(env, cnt, slot, j0); j0 += 4; }
        if (n_warm - j0 >= 2) { slide_warm_chunk<2, GW>(env, cnt, slot, j0); j0 += 2; }
    }
    env.stamp(3);
    // column pipeline: iteration j's column waits in set j & 1
    uint32_t bA[GW], bB[GW];
    env.fetch(env.iter_word(2 * j0), bA);
    env.fetch(env.iter_word(n_iter > j0 + 1 ? 2 * j0 + 2 : 2 * j0), bB);

    auto item = [&](typename Env::Rec &rec_cur, typename Env::Rec &rec_other, const SlideFetch<GW> &F_cur, SlideFetch<GW> &F_other,
                    const uint32_t (&sv)[kSlideStrict][GW]) __attribute__((always_inline)) {
        const int after = bd.item0 + done + 2;
        const typename Env::Rec rec = rec_cur;
        rec_cur = env.load_rec(after <= last_item ? after : last_item);
        slide_request<GW, USE_VALID>(env, rec_other, F_other);                    // (behind the band's last item: that item's again, unused)
        const uint32_t hdr = env.rec_word(rec, 0);
        uint32_t accPF[8], accR[4];
        if (ONLY_SIMPLE || (hdr & kSlSimple)) slide_item<LV, GW, true, USE_VALID, ONLY_SIMPLE, FAST>(env, A, rec, hdr, cnt, sv, F_cur, accPF, accR);
        else slide_item<LV, GW, false, USE_VALID, false, false>(env, A, rec, hdr, cnt, sv, F_cur, accPF, accR);
        env.commit(done, accPF, accR);
        done++;
    };
    // one iteration: the column waiting in `b` slides in (the column it pushes out of the ring was read an iteration ago into `bo`),
    // `b` is refilled with the column two iterations on, `bo` with the next iteration's outgoing column, the window's items run
    auto iteration = [&](uint32_t (&b)[GW], uint32_t (&bo)[GW], uint32_t (&bo_next)[GW], int j) __attribute__((always_inline)) {
        const uint32_t it1 = env.iter_word(2 * j + 1);
        uint32_t bn[GW];
#pragma unroll
        for (int i = 0; i < GW; i++) bn[i] = ~b[i];                    // rows that do not carry the reference base here
        env.fetch(env.iter_word(2 * j + 4), b);                        // (past the band's end: a row of the next band, unused)
#pragma unroll
        for (int i = 0; i < GW; i++) slide_updown(cnt[i], bn[i] ^ bo[i], bn[i]);
        env.ring_write(slot, bn);                                      // the column sliding out shared the slot (k columns apart)
        const int slot_now = slot;
        slot = slot + 1 == k ? 0 : slot + 1;
        env.ring_read(slot, bo_next);
        const int n_items = (int)(it1 >> 24);
        if (n_items == 0) return;                                      // warming up, or a window without chains
        // mismatch words of the reference at the strict positions of this window, out of the ring (slots beyond the launch's strict
        // positions read position 0 and meet empty masks)
        uint32_t sv[kSlideStrict][GW];
        if (FAST) {
            // the reference's mismatch words at the side's (up to three) strict positions, summed: bit 0 = xor3, bit 1 = majority
            uint32_t p[2][3][GW];
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const uint32_t pos = side ? A.rpos : A.fpos;
                const int n = (int)(pos >> 15);
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    if (SLIDE_UNLIKELY(q >= n)) {
#pragma unroll
                        for (int i = 0; i < GW; i++) p[side][q][i] = 0u;
                    } else {
                        int s = slot_now + 1 + (int)((pos >> (5 * q)) & 31u);
                        if (s >= k) s -= k;
                        env.ring_read(s, p[side][q]);
                    }
                }
            }
#pragma unroll
            for (int side = 0; side < 2; side++)
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    sv[2 * side][i] = bop<kSlXor3>(p[side][0][i], p[side][1][i], p[side][2][i]);
                    sv[2 * side + 1][i] = bop<kSlMaj>(p[side][0][i], p[side][1][i], p[side][2][i]);
                }
        }
#pragma unroll
        for (int q = 0; q < kSlideStrict; q++) {
            if (FAST) break;
            if (q < 4 || A.ns > 4) {
                int s = slot_now + 1 + (int)((A.spos >> (5 * q)) & 31u);
                if (s >= k) s -= k;
                env.ring_read(s, sv[q]);
            } else {
#pragma unroll
                for (int i = 0; i < GW; i++) sv[q][i] = 0u;
            }
        }
        // items in pairs: inside the loop the two register sets have fixed roles (a loop over single items that picks the set by parity
        // ends in one copy of the code that moves the freshly loaded set into place: 14 moves and a wait for the youngest load per item)
        int ii = 0;
        if (done & 1) { item(rec1, rec0, F1, F0, sv); ii = 1; }
#pragma unroll 1
        for (; ii + 1 < n_items; ii += 2) {
            item(rec0, rec1, F0, F1, sv);
            item(rec1, rec0, F1, F0, sv);
        }
        if (ii < n_items) item(rec0, rec1, F0, F1, sv);
    };
    // quarters of the band's windows behind the wave (the warm-up columns are nobody's progress: every wave has them).  Long bands only:
    // with the 8-window bands of a 131072-row shard the workgroups of a CU ending together only delays the patch units that wait for
    // their slots (tools/r05_prio.sh, profiles/r05_exp_prio.txt: 0.0334 against 0.0312 ms)
#ifndef SLIDE_PRIO_MIN_WIN
#define SLIDE_PRIO_MIN_WIN 32
#endif
    const int quarter_len = bd.n_win < SLIDE_PRIO_MIN_WIN ? 1 << 20 : ((bd.n_win + 3) / 4 > 2 ? (bd.n_win + 3) / 4 : 2);
    int quarter = 0, next_quarter = k - 1 + quarter_len;
    env.progress(0);
    uint32_t boA[GW], boB[GW];
    env.ring_read(slot, boA);                                           // zeros: the first k columns push nothing out
#pragma unroll 1
    for (int base = 0; base < n_iter; base += 30) {                     // 64 iteration words = 32 iterations, two of them look-ahead
        if (base) env.load_iters(bd.iter0 + 2 * base);
        const int n_here = n_iter - base < 30 ? n_iter - base : 30;     // 30 is even: an iteration's parity is that of j
        int j = base ? 0 : j0;                                          // (j0 is even and at most 30)
#pragma unroll 1
        for (; j + 1 < n_here; j += 2) {
            if (base + j >= next_quarter) { env.progress(++quarter); next_quarter += quarter_len; }
            iteration(bA, boA, boB, j);
            iteration(bB, boB, boA, j + 1);
        }
        if (j < n_here) iteration(bA, boA, boB, j);                     // (an odd tail ends the band)
    }
    env.progress(3);
}

}  // namespace mp
