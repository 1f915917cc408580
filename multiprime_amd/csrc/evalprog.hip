// evalprog.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of include/mprime.h.
// Candidate x sequence coverage evaluation of nested refinement chains (mis_primer_check + Y_distance, V20:1103-1130, 229-233):
// eval_prog_kernel — eval_chain_kernel's arithmetic (eval.hip: saturating bit-sliced mismatch counters over the one-hot column
// planes, one pass for the most degenerate member, one plane per refinement step) written for the ceiling that actually binds it.
//
// What binds: instruction ISSUE.  A SIMD of gfx950 issues about one instruction per 2.5 cycles whatever its kind — vector and
// scalar instructions share the slots (tools/ubench_salu.hip, profiles/r03_ubench_salu.json: 8 s_add + 8 v_add per iteration run
// at half the v_add rate of a pure VALU loop) — so the floor of a kernel is (VALU + SALU + memory instructions) / issue rate, not
// VALU alone.  eval_chain_kernel spends 765 scalar and 1198 vector instructions per wave of 8 x 64 row words, of which 872 are the
// counter arithmetic: 19 us of issue time at 131072 x 1000, measured 29.7.  Neither fewer bytes (evaltile.hip) nor more loads in
// flight change an instruction count; this kernel does:
//   * the host writes a per-item FETCH PROGRAM at upload time: one 32-bit entry per plane fetch (plane row = window position * 4
//     + base, the chain step of an event), one entry per lane of a register, broadcast with ONE v_readlane when its turn comes;
//     entries are grouped by what their consumption needs — single-base positions, two-base positions, each without / with a
//     strict position — so the loops carry no per-entry tests;
//   * a fetch is a buffer load: resource = the item's first plane row, scalar offset = entry row * row bytes, vector offset = the
//     lane's word offset, a constant of the wave: no vector address arithmetic, two scalar instructions per fetch;
//   * no software pipelining: with the issue slots full of other waves' work a fetch group's latency is hidden by them, and the
//     register copies and clamped re-fetches a rolling prefetch needs are instructions too;
//   * the 24 output slots of an item sit in the program, in the lanes that commit them (no late load).
#include "common.hpp"
#include "bitslice.hpp"
#include "evalprog.hpp"

using namespace mp;

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct EvalProgArgs {
    const uint32_t *cols32;            // [n_cols][4][nw32] one-hot column planes
    const uint32_t *excl32;            // [W][nw32]
    int nw32, p0;
    const uint32_t *prog;              // [chain item][kProgRegs][64]
    unsigned long long *out;
    BlockMap map;
    PatchArgs patch;                   // patch / IUPAC rows: the first patch.n_blocks workgroups run on their planes
};

struct Prog { uint32_t r[kProgRegs]; };

__device__ __forceinline__ uint32_t lane_of(uint32_t reg, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)reg, lane); }

// The lane's GW words of plane row `entry & kRowMask`.  GW = 16: two groups of 8 words with vector offsets of their own (plane
// rows are padded to multiples of 8 words, so either group is inside the row or entirely past it).
template <int GW>
struct Lane {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff[(GW + 7) / 8];
    uint32_t row_bytes;
    __device__ __forceinline__ void fetch(uint32_t entry, uint32_t (&d)[GW]) const {
        const int soff = (int)((entry & kRowMask) * row_bytes);
        if constexpr (GW >= 4) {
#pragma unroll
            for (int q = 0; q < GW / 4; q++) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[q / 2] + 16 * (q & 1), soff, 0);
                d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
            }
        } else if constexpr (GW == 2) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff[0], soff, 0);
            d[0] = v.x; d[1] = v.y;
        } else {
            d[0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff[0], soff, 0);
        }
    }
};

// WIDE: an item with more than 64 first-pass entries or events (k > 16 with many degenerate positions): entries come out of two
// registers.  Everything else is one v_readlane per entry.
// KS > 0: the planes a later event needs again are parked in LDS when the first pass fetches them (`keep`: the wave's own
// KS x 64 x GW words) — every event plane IS one of the first pass's planes, because the most degenerate member holds every base a
// later member loses — so the walk down the chain reads LDS instead of waiting for memory once per event.
template <int LV, int GW, int D, bool WIDE, int KS>
__device__ __forceinline__ void run_item(const Prog &P, const Lane<GW> &L, const uint32_t (&valid)[GW], uint32_t (&acc)[16], uint32_t *keep) {
    constexpr int CC = 8;
    const int lane = (int)(threadIdx.x & 63);
    auto park = [&](uint32_t en, const uint32_t (&d)[GW]) {              // slot (en >> 8) & 15 of the wave's scratch
        uint32_t *dst = keep + (((en >> 8) & 15u) * 64 + (uint32_t)lane) * GW;
        if constexpr (GW == 2) *reinterpret_cast<uint2 *>(dst) = uint2{d[0], d[1]};
#pragma unroll
        for (int q = 0; q < GW / 4; q++) *reinterpret_cast<u32x4 *>(dst + 4 * q) = u32x4{d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
    };
    auto unpark = [&](uint32_t en, uint32_t (&d)[GW]) {
        const uint32_t *src = keep + (((en >> 8) & 15u) * 64 + (uint32_t)lane) * GW;
        if constexpr (GW == 2) { const uint2 v = *reinterpret_cast<const uint2 *>(src); d[0] = v.x; d[1] = v.y; }
#pragma unroll
        for (int q = 0; q < GW / 4; q++) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(src + 4 * q);
            d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
        }
    };
    const int n_steps = (int)lane_of(P.r[0], 33), n_ev = (int)lane_of(P.r[0], 34);
    const int nA = (int)lane_of(P.r[0], 35), nB = (int)lane_of(P.r[0], 36), nC = (int)lane_of(P.r[0], 37), nD = (int)lane_of(P.r[0], 38);
    const int nE = (int)lane_of(P.r[0], 39);
    auto pass_entry = [&](int q) -> uint32_t {
        if constexpr (WIDE) { const uint32_t a = lane_of(P.r[1], q & 63), b = lane_of(P.r[2], q & 63); return q < 64 ? a : b; }
        else return lane_of(P.r[1], q);
    };
    auto event_entry = [&](int q) -> uint32_t {
        if constexpr (WIDE) { const uint32_t a = lane_of(P.r[3], q & 63), b = lane_of(P.r[4], q & 63); return q < 64 ? a : b; }
        else return lane_of(P.r[3], q);
    };
    uint32_t t1[GW], t2[GW], t3[GW], t4[GW], sf[GW], sr[GW];
#pragma unroll
    for (int i = 0; i < GW; i++) t1[i] = t2[i] = t3[i] = t4[i] = sf[i] = sr[i] = 0;
    auto count = [&](const uint32_t (&m)[GW]) {
#pragma unroll
        for (int i = 0; i < GW; i++) count_unmatched<LV>(t1[i], t2[i], t3[i], t4[i], m[i]);
    };
    auto strict = [&](uint32_t e, const uint32_t (&m)[GW]) {           // e carries at least one of the two flags
        const uint32_t fF = (e & kStrictF) ? 0xFFFFFFFFu : 0u, fR = (e & kStrictR) ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int i = 0; i < GW; i++) {
            sf[i] = __builtin_amdgcn_bitop3_b32(sf[i], m[i], fF, kLutOrNotAnd);
            sr[i] = __builtin_amdgcn_bitop3_b32(sr[i], m[i], fR, kLutOrNotAnd);
        }
    };
    // (1) the first (most degenerate) member over all k positions.  Lists A / B: positions whose symbol is ONE base, without /
    // with a strict position; C / D: TWO bases (two entries each); E: the rest, one entry per base, kMore = more bases follow.
    int q = 0;
    auto singles = [&](int n, bool with_strict) {
        const int end = q + n;
#pragma unroll 1
        for (; q + D <= end; q += D) {
            uint32_t en[D], ld[D][GW];
#pragma unroll
            for (int u = 0; u < D; u++) { en[u] = pass_entry(q + u); L.fetch(en[u], ld[u]); }
#pragma unroll
            for (int u = 0; u < D; u++) { count(ld[u]); if (with_strict) strict(en[u], ld[u]); }
        }
#pragma unroll 1
        for (; q < end; q++) {
            uint32_t ld[GW];
            const uint32_t en = pass_entry(q);
            L.fetch(en, ld);
            count(ld);
            if (with_strict) strict(en, ld);
        }
    };
    auto pairs = [&](int n, bool with_strict) {
        constexpr int D2 = D > 1 ? D / 2 : 1;
        const int end = q + 2 * n;
#pragma unroll 1
        for (; q + 2 * D2 <= end; q += 2 * D2) {
            uint32_t en[D2], la[D2][GW], lb[D2][GW];
#pragma unroll
            for (int u = 0; u < D2; u++) { en[u] = pass_entry(q + 2 * u); L.fetch(en[u], la[u]); L.fetch(pass_entry(q + 2 * u + 1), lb[u]); }
#pragma unroll
            for (int u = 0; u < D2; u++) {
                if (KS > 0 && (en[u] & kKeepNext)) park(en[u], lb[u]);     // the pair's second base is the one a later member loses
#pragma unroll
                for (int i = 0; i < GW; i++) la[u][i] |= lb[u][i];
                count(la[u]);
                if (with_strict) strict(en[u], la[u]);
            }
        }
#pragma unroll 1
        for (; q < end; q += 2) {
            uint32_t la[GW], lb[GW];
            const uint32_t en = pass_entry(q);
            L.fetch(en, la);
            L.fetch(pass_entry(q + 1), lb);
            if (KS > 0 && (en & kKeepNext)) park(en, lb);
#pragma unroll
            for (int i = 0; i < GW; i++) la[i] |= lb[i];
            count(la);
            if (with_strict) strict(en, la);
        }
    };
    singles(nA, false);
    singles(nB, true);
    pairs(nC, false);
    pairs(nD, true);
    {
        const int end = q + nE;
#pragma unroll 1
        while (q < end) {
            uint32_t m[GW], en;
#pragma unroll
            for (int i = 0; i < GW; i++) m[i] = 0u;
#pragma unroll 1
            do {
                en = pass_entry(q);
                q++;
                uint32_t pl[GW];
                L.fetch(en, pl);
                if (KS > 0 && (en & kKeepThis)) park(en, pl);
#pragma unroll
                for (int i = 0; i < GW; i++) m[i] |= pl[i];
            } while ((en & kMore) && q < end);
            count(m);
            if (en & (kStrictF | kStrictR)) strict(en, m);
        }
    }
    // (2) walk down the chain: the events of step s (one lost base each: its plane IS the increment), then member s is counted.
    // The three counts of a member are at most 32 * GW per thread: one register per member while they fit 10 bits each.
    int e = 0;
    uint32_t evw = n_ev ? event_entry(0) : 0u;
#pragma unroll
    for (int s = 0; s < CC; s++) {
        if (s >= n_steps) break;
        if (s > 0) {
#pragma unroll 1
            while (e < n_ev && (int)((evw >> 24) & 15u) == s) {
                uint32_t d[GW];
                if (KS > 0 && (evw & kKeepThis)) unpark(evw, d);
                else L.fetch(evw, d);
#pragma unroll
                for (int i = 0; i < GW; i++) count_plane<LV>(t1[i], t2[i], t3[i], t4[i], d[i]);
                if (evw & (kStrictF | kStrictR)) {
                    const uint32_t fF = (evw & kStrictF) ? 0xFFFFFFFFu : 0u, fR = (evw & kStrictR) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                    for (int i = 0; i < GW; i++) {
                        sf[i] = __builtin_amdgcn_bitop3_b32(sf[i], d[i], fF, kLutOrAnd);
                        sr[i] = __builtin_amdgcn_bitop3_b32(sr[i], d[i], fR, kLutOrAnd);
                    }
                }
                e++;
                evw = e < n_ev ? event_entry(e) : 0u;
            }
        }
        uint32_t nP = 0, nF = 0, nR = 0;
#pragma unroll
        for (int i = 0; i < GW; i++) {
            const uint32_t far = LV == 1 ? t1[i] : (LV == 2 ? t2[i] : (LV == 3 ? t3[i] : t4[i]));
            nP += __popc(valid[i] & ~t1[i]);
            nF += __popc(__builtin_amdgcn_bitop3_b32(valid[i], far, sf[i], kLutAndNotNot));
            nR += __popc(__builtin_amdgcn_bitop3_b32(valid[i], far, sr[i], kLutAndNotNot));
        }
        if constexpr (32 * GW < 1024) acc[s] = nP | (nF << 10) | (nR << 20);
        else { acc[s] = nP | (nF << 16); acc[CC + s] = nR; }
    }
}

template <int LV, int GW, int D, int KS>
__global__ __launch_bounds__(kBlock) void eval_prog_kernel(const EvalProgArgs A) {
    static_assert(GW <= 8 || GW == 16, "plane rows are padded to multiples of 8 words");
    static_assert(KS == 0 || GW >= 2, "parked planes move as 8- or 16-byte vectors");
    constexpr int CC = 8, NG = (GW + 7) / 8;
    constexpr bool kPacked = 32 * GW < 1024;               // three 10-bit counts per register
    __shared__ uint32_t s_part[kBlock / 64][3 * CC / 2];
    __shared__ __align__(16) uint32_t s_keep[KS > 0 ? (kBlock / 64) * KS * 64 * GW : 4];
    const bool on_patch = (int)blockIdx.x < A.patch.n_blocks;
    const int lane = (int)(threadIdx.x & 63);
    int slice, item, word0;
    if (on_patch) {                                    // a wave per patch unit: everything below is wave-uniform
        const int unit = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)));
        item = unit / A.patch.per_item;
        slice = unit % A.patch.per_item;
        if (item >= A.map.n_items) return;
        word0 = (slice * 64 + lane) * GW;
    } else {
        if (!map_block(A.map, blockIdx.x - A.patch.n_blocks, slice, item)) return;
        word0 = (slice * kBlock + (int)threadIdx.x) * GW;
    }
    const bool per_wave = on_patch;                            // a patch unit's wave commits its own totals
    Prog P;
    {
        const uint32_t *src = A.prog + (size_t)item * (kProgRegs * 64) + lane;
#pragma unroll
        for (int q = 0; q < kProgRegs; q++) P.r[q] = src[q * 64];
    }
    const int win = (int)lane_of(P.r[0], 32);
    const uint32_t *base, *mask;
    uint32_t row_words, flip;
    if (on_patch) {
        const PatchWin pw = A.patch.pwin[win];
        if (slice * 64 * GW >= pw.npw) return;           // nothing of this window's patch planes left for the wave
        base = A.patch.pplanes + pw.poff;
        mask = A.patch.pvalid + pw.voff;
        row_words = (uint32_t)pw.npw;
        flip = 0u;
    } else {
        base = A.cols32 + (size_t)(A.p0 + win) * 4 * (size_t)A.nw32;
        mask = A.excl32 + (size_t)win * (size_t)A.nw32;
        row_words = (uint32_t)A.nw32;
        flip = 0xFFFFFFFFu;
    }
    Lane<GW> L;
    L.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(base), 0, 0x7FFFFFFF, 0x00020000);
    L.row_bytes = row_words * 4u;
    uint32_t valid[GW];
#pragma unroll
    for (int g = 0; g < NG; g++) {                        // a group of (up to) 8 words is inside the row or past its end
        constexpr int GN = GW < 8 ? GW : 8;
        const int w0 = word0 + 8 * g;
        const bool live = w0 < (int)row_words;
        const int w_safe = live ? w0 : 0;
        L.voff[g] = w_safe * 4;
        uint32_t raw[GN];
#pragma unroll
        for (int i = 0; i < GN; i++) raw[i] = mask[w_safe + i];
#pragma unroll
        for (int i = 0; i < GN; i++) valid[8 * g + i] = live ? (raw[i] ^ flip) : 0u;
    }
    uint32_t acc[2 * CC];
#pragma unroll
    for (int c = 0; c < 2 * CC; c++) acc[c] = 0;
    uint32_t *keep = s_keep + (KS > 0 ? (threadIdx.x >> 6) * (KS * 64 * GW) : 0);
    if (lane_of(P.r[0], 40)) run_item<LV, GW, D, true, KS>(P, L, valid, acc, keep);
    else run_item<LV, GW, D, false, KS>(P, L, valid, acc, keep);
    // commit: wave totals by DPP (bitslice.hpp), the output slots of the 24 counters come from lanes 0-23 of program register 0
    uint32_t tot[3 * CC / 2];
    if constexpr (kPacked) {
#pragma unroll
        for (int q = 0; q < 3 * CC / 2; q++) {
            const int a = 2 * q, b = 2 * q + 1;
            const uint32_t va = (acc[a / 3] >> (10 * (a % 3))) & 1023u, vb = (acc[b / 3] >> (10 * (b % 3))) & 1023u;
            tot[q] = wave_sum_lane63(va | (vb << 16));
        }
    } else {                                               // 16 words per lane: 512 per count and lane, 32768 per wave: 16-bit halves
#pragma unroll
        for (int q = 0; q < 3 * CC / 2; q++) {
            const int a = 2 * q, b = 2 * q + 1;
            auto val = [&](int t) { const int c = t / 3, r = t % 3; return r == 0 ? (acc[c] & 0xFFFFu) : (r == 1 ? (acc[c] >> 16) : acc[CC + c]); };
            tot[q] = wave_sum_lane63(val(a) | (val(b) << 16));
        }
    }
    const int wv = (int)(threadIdx.x >> 6);
    if (lane == 63) {
#pragma unroll
        for (int q = 0; q < 3 * CC / 2; q++) s_part[wv][q] = tot[q];
    }
    if (per_wave) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 3 * CC) {
            const int c = lane / 3, rr = lane % 3;
            const uint32_t mine = (s_part[wv][lane >> 1] >> (16 * (lane & 1))) & 0xFFFFu;
            const uint32_t perfect = (s_part[wv][(3 * c) >> 1] >> (16 * ((3 * c) & 1))) & 0xFFFFu;
            const uint32_t val = rr ? mine - perfect : mine;
            const int oc = (int)P.r[0];
            if (oc >= 0 && val) atomicAdd(&A.out[(size_t)oc * 3 + rr], (unsigned long long)val);
        }
    } else {
        __syncthreads();
        if (threadIdx.x < 3 * CC) {                          // wave 0: its lanes 0-23 hold the slots
            const int c = lane / 3, rr = lane % 3;
            uint32_t mine = 0, perfect = 0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; w++) {
                mine += (s_part[w][lane >> 1] >> (16 * (lane & 1))) & 0xFFFFu;
                perfect += (s_part[w][(3 * c) >> 1] >> (16 * ((3 * c) & 1))) & 0xFFFFu;
            }
            const uint32_t val = rr ? mine - perfect : mine;
            const int oc = (int)P.r[0];
            if (oc >= 0 && val) atomicAdd(&A.out[(size_t)oc * 3 + rr], (unsigned long long)val);
        }
    }
}

typedef void (*ProgFn)(const EvalProgArgs);

}  // namespace

namespace mp {

// One fixed block of kProgRegs x 64 entries per chain item (layout: evalprog.hpp).  Chains of at most 8 members only: 4 k <= 124
// first-pass entries and 3 k <= 93 events are the most k <= 31 allows, each within its two registers.
void build_eval_programs(const std::vector<ChainItem> &chains, const std::vector<uint32_t> &events, const std::vector<int32_t> &cand_out,
                         int k, uint32_t sF, uint32_t sR, int keep_slots, std::vector<uint32_t> &prog) {
    prog.assign(chains.size() * (size_t)(kProgRegs * 64), 0u);
    const uint32_t any_strict = sF | sR;
    for (size_t i = 0; i < chains.size(); i++) {
        const ChainItem &ch = chains[i];
        auto entry = [&](int j, int base) {
            return (uint32_t)(j * 4 + base) | (((sF >> j) & 1u) ? kStrictF : 0u) | (((sR >> j) & 1u) ? kStrictR : 0u);
        };
        auto sym = [&](int j) { return (ch.sym[j >> 3] >> (4 * (j & 7))) & 15u; };
        uint32_t *blk = prog.data() + i * (size_t)(kProgRegs * 64);
        // the first `keep_slots` events get a slot of the wave's LDS scratch: slot_of[position * 4 + base] = slot + 1
        int slot_of[MP_MAX_K * 4 + 4] = {0};
        for (int q = 0; q < ch.n_ev && q < keep_slots; q++) {
            const uint32_t ev = events[(size_t)ch.ev0 + (size_t)q];
            slot_of[(ev & 255u) * 4 + (uint32_t)__builtin_ctz((ev >> 8) & 15u)] = q + 1;
        }
        int n_fp = 0, n[5] = {0, 0, 0, 0, 0};
        for (int list = 0; list < 5; list++)              // A, B: one base without / with a strict position; C, D: two bases; E: the rest
            for (int j = 0; j < k; j++) {
                const uint32_t sy = sym(j);
                const int nb = __builtin_popcount(sy);
                const bool st = (any_strict >> j) & 1u;
                const int mine = nb == 1 ? (st ? 1 : 0) : (nb == 2 ? (st ? 3 : 2) : 4);
                if (mine != list || nb == 0) continue;
                int bases[4], nbase = 0;
                for (uint32_t rest = sy; rest; rest &= rest - 1u) bases[nbase++] = __builtin_ctz(rest);
                if (nb == 2 && slot_of[j * 4 + bases[0]]) std::swap(bases[0], bases[1]);      // a pair's parked base goes second
                for (int t = 0; t < nbase; t++) {
                    uint32_t e = entry(j, bases[t]) | (t + 1 < nbase ? kMore : 0u);
                    if (nb == 2 && t == 0 && slot_of[j * 4 + bases[1]]) e |= kKeepNext | ((uint32_t)(slot_of[j * 4 + bases[1]] - 1) << 8);
                    if (nb > 2 && slot_of[j * 4 + bases[t]]) e |= kKeepThis | ((uint32_t)(slot_of[j * 4 + bases[t]] - 1) << 8);
                    blk[64 + n_fp++] = e;
                }
                n[list]++;
            }
        int n_e_entries = 0;
        for (int j = 0; j < k; j++) if (__builtin_popcount(sym(j)) > 2) n_e_entries += __builtin_popcount(sym(j));
        for (int q = 0; q < ch.n_ev; q++) {
            const uint32_t ev = events[(size_t)ch.ev0 + (size_t)q];          // position | lost base (one-hot) << 8 | step << 16
            blk[192 + q] = entry((int)(ev & 255u), __builtin_ctz((ev >> 8) & 15u)) | ((ev >> 16) << 24) |
                           (q < keep_slots ? kKeepThis | ((uint32_t)q << 8) : 0u);
        }
        for (int t = 0; t < 24; t++) blk[t] = (uint32_t)cand_out[(size_t)ch.cand0 + (size_t)(t / 3)];
        const uint32_t head[9] = {(uint32_t)ch.win, (uint32_t)ch.n_steps, (uint32_t)ch.n_ev, (uint32_t)n[0], (uint32_t)n[1], (uint32_t)n[2],
                                  (uint32_t)n[3], (uint32_t)n_e_entries, (uint32_t)(n_fp > 64 || ch.n_ev > 64)};
        for (int q = 0; q < 9; q++) blk[32 + q] = head[q];
    }
}

// shape 0-8: words per thread x fetches per group as eval_chain_kernel's MP_EVAL_CHAIN table; 9 = 16 words per thread; 10-13 park the
// event planes in LDS (kProgKeep[shape] slots per wave): 4 words x 8 slots, 4 x 4 (the default from 393 216 rows up), 8 x 4, 4 x 4 with
// two fetches in flight (eight more shapes were measured and dropped: profiles/r03_prog_keep.txt)
int launch_eval_prog(mp_ctx *c, int shape, const BlockMap &bm, const PatchArgs &pa, unsigned grid, unsigned long long *device_out) {
#define PROG_ROW(LV) {eval_prog_kernel<LV, 2, 6, 0>, eval_prog_kernel<LV, 2, 3, 0>, eval_prog_kernel<LV, 2, 8, 0>, eval_prog_kernel<LV, 4, 3, 0>, \
                      eval_prog_kernel<LV, 4, 6, 0>, eval_prog_kernel<LV, 1, 6, 0>, eval_prog_kernel<LV, 8, 2, 0>, eval_prog_kernel<LV, 8, 4, 0>, \
                      eval_prog_kernel<LV, 8, 1, 0>, eval_prog_kernel<LV, 16, 2, 0>, eval_prog_kernel<LV, 4, 4, 8>, eval_prog_kernel<LV, 4, 4, 4>, \
                      eval_prog_kernel<LV, 8, 4, 4>, eval_prog_kernel<LV, 4, 2, 4>}
    static const ProgFn fn[4][kProgShapes] = {PROG_ROW(1), PROG_ROW(2), PROG_ROW(3), PROG_ROW(4)};
#undef PROG_ROW
    EvalProgArgs a{reinterpret_cast<const uint32_t *>(c->cols), reinterpret_cast<const uint32_t *>(c->excl), c->n_pad / 32, c->p0,
                   c->chain_prog, device_out, bm, pa};
    hipLaunchKernelGGL(fn[c->v][shape], dim3(grid + (unsigned)pa.n_blocks), dim3(kBlock), 0, c->stream, a);
    HIPCK(c, hipGetLastError());
    return MP_OK;
}

}  // namespace mp
