// chainbody.hpp — the nested-chain evaluation of one workgroup (eval_chain_kernel's body) as a device function, shared by eval.hip
// (the kernel proper) and evalslide.hip (whose launch runs the patch units of the same step in the tail of its own grid).
#pragma once

#include "common.hpp"
#include "bitslice.hpp"

namespace mp {

// ----------------------------------------------------------------------------------------------
// (4c) bit-sliced evaluation of a NESTED chain: the candidates of the item, ordered from the most degenerate
// one down, each accept a subset of what the previous one accepts (a refinement chain read backwards).  Then
// the mismatch counters of candidate s are those of candidate s-1 plus, for every position whose symbol lost
// bases in between, one bit plane d = "the sequence's base is one of the lost ones" — for one lost base that is
// the base's column plane as loaded.  Counters only ever grow, so saturating bit-sliced counters stay exact.
// Work per 32 sequences at v = 1: k positions once (1 load + 2-4 VALU each, for the first candidate) + 1 load and
// 2-4 VALU per event + 6 VALU per candidate, against 4 loads and ~45 VALU per position in the symbol-table kernel.
// ----------------------------------------------------------------------------------------------
struct EvalChainArgs {
    const unsigned long long *cols;    // [n_cols][4][nw]
    const unsigned long long *excl;    // [W][nw]
    int nw, p0, k, v;
    const ChainItem *items;
    const uint32_t *events;            // position | lost base (one-hot) << 8 | step << 16, ascending by step
    const int32_t *cand_out;
    uint32_t sF, sR;
    unsigned long long *out;
    BlockMap map;
    PatchArgs patch;                   // patch / IUPAC rows: the first patch.n_blocks workgroups run on their planes
    const ChainItem *neg_items;        // the items of the subtracting run (patch.neg_blocks workgroups), n_neg of them
    int n_neg;
    unsigned long long *clear;         // mp_eval_launch_rotating: the NEXT launch's counter block, zeroed by this launch's grid (may be null)
    uint32_t n_clear;
};

// The side job of a rotating launch: the counter block the next launch will add to is set to zero by THIS launch's own workgroups
// (one store per thread at the bench sizes) instead of by a fill dispatch between two evaluations — a dispatch of its own costs 4-5 us
// on the stream, a sixth of the evaluation of a 131072-row shard.
__device__ __forceinline__ void clear_counters(unsigned long long *__restrict__ p, uint32_t n, unsigned bid, unsigned n_blocks) {
    if (!p) return;
    for (uint32_t i = bid * kBlock + threadIdx.x; i < n; i += n_blocks * kBlock) p[i] = 0ull;
}

// One pass of the first candidate over the positions in `rem` whose symbol has NB bases (NB = 4: three or four,
// all planes loaded and masked).  D positions of loads are requested before the first one is consumed; no other
// load sits in between, so each position waits for exactly its own words.
template <int LV, int GW, int D, int NB>
__device__ __forceinline__ void chain_first_pass(uint32_t rem, const uint32_t *Pw, size_t nw32, unsigned long long sy_lo,
                                                 unsigned long long sy_hi, uint32_t sF, uint32_t sR, uint32_t (&t1)[GW],
                                                 uint32_t (&t2)[GW], uint32_t (&t3)[GW], uint32_t (&t4)[GW], uint32_t (&sf)[GW],
                                                 uint32_t (&sr)[GW]) {
#pragma unroll 1
    while (rem) {
        int js[D]; bool has[D];
        uint32_t ld[D][NB][GW], sys[D];
#pragma unroll
        for (int u = 0; u < D; u++) {
            has[u] = rem != 0u;
            js[u] = has[u] ? __builtin_ctz(rem) : js[0];
            rem &= rem - 1u;
            const int j = js[u];
            const uint32_t sy = (uint32_t)((j < 16 ? sy_lo : sy_hi) >> (4 * (j & 15))) & 15u;
            sys[u] = sy;
            const uint32_t *P = Pw + (size_t)j * 4 * nw32;
            if (NB == 4) {
#pragma unroll
                for (int b = 0; b < 4; b++)
#pragma unroll
                    for (int i = 0; i < GW; i++) ld[u][b][i] = P[b * nw32 + i];
            } else {
                const uint32_t second = sy & (sy - 1u);
                const size_t b0 = (size_t)__builtin_ctz(sy | 16u), b1 = (size_t)__builtin_ctz(second | 16u);
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    ld[u][0][i] = P[b0 * nw32 + i];
                    if (NB == 2) ld[u][1][i] = P[b1 * nw32 + i];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < D; u++) {
            if (!has[u]) break;
            const int j = js[u];
            uint32_t m[GW];
            if (NB == 4) {
                const uint32_t sy = sys[u];
                const uint32_t kA = (sy & 1u) ? 0xFFFFFFFFu : 0u, kC = (sy & 2u) ? 0xFFFFFFFFu : 0u;
                const uint32_t kG = (sy & 4u) ? 0xFFFFFFFFu : 0u, kT = (sy & 8u) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    m[i] = ld[u][0][i] & kA;
                    m[i] = __builtin_amdgcn_bitop3_b32(ld[u][1][i], kC, m[i], kLutAndOr);
                    m[i] = __builtin_amdgcn_bitop3_b32(ld[u][2][i], kG, m[i], kLutAndOr);
                    m[i] = __builtin_amdgcn_bitop3_b32(ld[u][3][i], kT, m[i], kLutAndOr);
                }
            } else {
#pragma unroll
                for (int i = 0; i < GW; i++) m[i] = NB == 2 ? (ld[u][0][i] | ld[u][1][i]) : ld[u][0][i];
            }
#pragma unroll
            for (int i = 0; i < GW; i++) count_unmatched<LV>(t1[i], t2[i], t3[i], t4[i], m[i]);
            if (((sF | sR) >> j) & 1u) {
                const uint32_t fF = ((sF >> j) & 1u) ? 0xFFFFFFFFu : 0u, fR = ((sR >> j) & 1u) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    sf[i] = __builtin_amdgcn_bitop3_b32(sf[i], m[i], fF, kLutOrNotAnd);
                    sr[i] = __builtin_amdgcn_bitop3_b32(sr[i], m[i], fR, kLutOrNotAnd);
                }
            }
        }
    }
}

// One workgroup of the nested-chain evaluation (`bid` = its index in the launch): eval_chain_kernel below, and — for the patch
// units of a sliding launch — the tail of eval_slide_kernel's grid (evalslide.hip).
template <int LV, int GW, int D>
__device__ __forceinline__ void eval_chain_block(const EvalChainArgs &A, uint32_t (&s_part)[kBlock / 64][12], const unsigned bid) {
    static_assert(GW <= 8, "plane rows are padded to multiples of 8 words");
    constexpr int CC = 8;
    // the patch units come first; a second run of them (sliding evaluation) subtracts the plain slices of the items that slide
    const int patch_blocks = A.patch.n_blocks + A.patch.neg_blocks;
    const bool on_patch = (int)bid < patch_blocks;
    const bool negative = on_patch && (int)bid >= A.patch.n_blocks;
    int slice, item, word0;
    if (on_patch) {                                    // a wave per patch unit: everything below is wave-uniform
        const int pb = (int)bid - (negative ? A.patch.n_blocks : 0);
        const int unit = __builtin_amdgcn_readfirstlane(pb * (kBlock / 64) + (int)(threadIdx.x >> 6));
        item = unit / A.patch.per_item;
        slice = unit % A.patch.per_item;
        if (item >= (negative ? A.n_neg : A.map.n_items)) return;
        word0 = (slice * 64 + (int)(threadIdx.x & 63)) * GW;
    } else {
        if (!map_block(A.map, bid - (unsigned)patch_blocks, slice, item)) return;
        word0 = (slice * kBlock + threadIdx.x) * GW;
    }
    const ChainItem it = negative ? A.neg_items[item] : A.items[item];
    const WordTile T = !on_patch ? column_tile(A.cols, A.excl, A.nw, A.p0, it.win, word0)
                                 : (negative ? plain_tile(A.patch, it.win, word0) : patch_tile(A.patch, it.win, word0));
    if (on_patch && slice * 64 * GW >= (int)T.stride) return;               // nothing of this window's patch planes left for the wave
    const size_t nw32 = T.stride;
    const bool live = T.live;
    // the three counts of a chain member are written once (when the walk reaches it) and are at most 32 * GW <= 256 per thread:
    // one register per member (10 bits each) instead of three keeps the kernel at 8 waves per SIMD
    static_assert(32 * GW < 1024, "three counts per register need 10 bits each");
    uint32_t acc[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) acc[c] = 0;
    if (live) {
        uint32_t t1[GW], t2[GW], t3[GW], t4[GW], sf[GW], sr[GW];
#pragma unroll
        for (int i = 0; i < GW; i++) t1[i] = t2[i] = t3[i] = t4[i] = sf[i] = sr[i] = 0;
        const uint32_t *Pw = T.planes;
        const unsigned long long sy_lo = it.sym[0] | ((unsigned long long)it.sym[1] << 32);
        const unsigned long long sy_hi = it.sym[2] | ((unsigned long long)it.sym[3] << 32);
        // (1) the first candidate over all k positions, grouped by the number of bases of its symbol there
        chain_first_pass<LV, GW, D, 1>(it.pos1, Pw, nw32, sy_lo, sy_hi, A.sF, A.sR, t1, t2, t3, t4, sf, sr);
        chain_first_pass<LV, GW, (D + 1) / 2, 2>(it.pos2, Pw, nw32, sy_lo, sy_hi, A.sF, A.sR, t1, t2, t3, t4, sf, sr);
        chain_first_pass<LV, GW, (D + 3) / 4, 4>(it.pos4, Pw, nw32, sy_lo, sy_hi, A.sF, A.sR, t1, t2, t3, t4, sf, sr);
        uint32_t valid[GW];
        {
#pragma unroll
            for (int i = 0; i < GW; i++) valid[i] = T.mask[i] ^ T.mask_flip;
        }
        // (2) walk down the chain: apply the events of step s, then count candidate s
        const uint32_t *ev = A.events + it.ev0;
        int e = 0;
        uint32_t evw = it.n_ev ? ev[0] : (1u << 8);
        uint32_t cur[GW];
        {
            const uint32_t *P = Pw + ((size_t)(evw & 255u) * 4 + (size_t)__builtin_ctz(((evw >> 8) & 15u) | 16u)) * nw32;
#pragma unroll
            for (int i = 0; i < GW; i++) cur[i] = P[i];
        }
#pragma unroll
        for (int s = 0; s < CC; s++) {
            if (s >= it.n_steps) break;
            if (s > 0) {
#pragma unroll 1
                while (e < it.n_ev && (int)(evw >> 16) == s) {
                    e++;
                    const uint32_t evn = e < it.n_ev ? ev[e] : evw;          // the plane of the next event is on its way
                    uint32_t nxt[GW];
                    {
                        const uint32_t *P = Pw + ((size_t)(evn & 255u) * 4 + (size_t)__builtin_ctz(((evn >> 8) & 15u) | 16u)) * nw32;
#pragma unroll
                        for (int i = 0; i < GW; i++) nxt[i] = P[i];
                    }
                    const uint32_t j = evw & 255u;
#pragma unroll
                    for (int i = 0; i < GW; i++) count_plane<LV>(t1[i], t2[i], t3[i], t4[i], cur[i]);
                    if (((A.sF | A.sR) >> j) & 1u) {
                        const uint32_t fF = ((A.sF >> j) & 1u) ? 0xFFFFFFFFu : 0u, fR = ((A.sR >> j) & 1u) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                        for (int i = 0; i < GW; i++) {
                            sf[i] = __builtin_amdgcn_bitop3_b32(sf[i], cur[i], fF, kLutOrAnd);
                            sr[i] = __builtin_amdgcn_bitop3_b32(sr[i], cur[i], fR, kLutOrAnd);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < GW; i++) cur[i] = nxt[i];
                    evw = evn;
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
            acc[s] = nP | (nF << 10) | (nR << 20);
        }
    }
    uint32_t accP[CC], accF[CC], accR[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) { accP[c] = acc[c] & 1023u; accF[c] = (acc[c] >> 10) & 1023u; accR[c] = acc[c] >> 20; }
    if (on_patch) wave_commit<GW>(accP, accF, accR, s_part[threadIdx.x >> 6], A.cand_out + it.cand0, A.out, negative);
    else block_commit<GW>(accP, accF, accR, s_part, A.cand_out + it.cand0, A.out);
}

// One PATCH UNIT per wave with ONE word per lane (64 words = 2048 rows of a window's patch planes) — the form the sliding launch runs
// its patch units in.  A unit of eval_chain_block<LV, 8, 4> took ~10 us at the 131072-row shard whatever little it had to do (23 words
// of patch rows, 3 lanes of 64 busy): its walk fetches an event word, then that event's plane, one event after the other — fourteen
// dependent memory round trips — and those units are the tail of the launch, nothing hides them (tools/slide_stamps.py: slide
// workgroups done at 25 us, patch units until 30).  Here everything a unit needs is requested in three rounds: the item, then its
// event words (one per lane, read back with v_readlane) together with the first-pass planes (16 positions in flight), then ALL event
// planes at once into the wave's stash in LDS (kPatchEvents of them; a chain with more events falls back to one-ahead fetches from
// there on); the walk then reads the stash.  `stash`: 64 x kPatchEvents words of LDS of this wave; `row`: its 12 words for the sums.
constexpr int kPatchEvents = 16;
template <int LV>
__device__ __forceinline__ void eval_patch_wave(const EvalChainArgs &A, uint32_t *row, uint32_t *stash, const unsigned unit, const bool negative) {
    constexpr int CC = 8, GW = 1;
    const int lane = (int)(threadIdx.x & 63);
    const int item = __builtin_amdgcn_readfirstlane((int)(unit / (unsigned)A.patch.per_item)), slice = __builtin_amdgcn_readfirstlane((int)(unit % (unsigned)A.patch.per_item));
    if (item >= (negative ? A.n_neg : A.map.n_items)) return;
    const int word0 = slice * 64 + lane;
    const ChainItem it = negative ? A.neg_items[item] : A.items[item];
    const WordTile T = negative ? plain_tile(A.patch, it.win, word0) : patch_tile(A.patch, it.win, word0);
    if (slice * 64 >= (int)T.stride) return;                               // nothing of this window's patch planes left for the wave
    const size_t nw32 = T.stride;
    const uint32_t *Pw = T.planes;
    const uint32_t *ev = A.events + it.ev0;
    const int n_ev = __builtin_amdgcn_readfirstlane((int)it.n_ev), n_steps = __builtin_amdgcn_readfirstlane((int)it.n_steps);
    // round 2: the event words (lane e holds event e) ...
    const uint32_t evv = lane < n_ev ? ev[lane] : 0u;
    uint32_t acc[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) acc[c] = 0;
    // (a lane past the window's patch words computes on word 0 of the planes — inside the allocation — and is masked at the counts)
    const uint32_t *Pl = T.live ? Pw : Pw - (word0 - slice * 64) ;
    uint32_t t1[GW] = {0}, t2[GW] = {0}, t3[GW] = {0}, t4[GW] = {0}, sf[GW] = {0}, sr[GW] = {0};
    const unsigned long long sy_lo = it.sym[0] | ((unsigned long long)it.sym[1] << 32);
    const unsigned long long sy_hi = it.sym[2] | ((unsigned long long)it.sym[3] << 32);
    // ... and the first candidate over all k positions, 16 / 8 / 4 positions in flight
    chain_first_pass<LV, GW, 16, 1>(it.pos1, Pl, nw32, sy_lo, sy_hi, A.sF, A.sR, t1, t2, t3, t4, sf, sr);
    chain_first_pass<LV, GW, 8, 2>(it.pos2, Pl, nw32, sy_lo, sy_hi, A.sF, A.sR, t1, t2, t3, t4, sf, sr);
    chain_first_pass<LV, GW, 4, 4>(it.pos4, Pl, nw32, sy_lo, sy_hi, A.sF, A.sR, t1, t2, t3, t4, sf, sr);
    const uint32_t valid = T.live ? (T.mask[0] ^ T.mask_flip) : 0u;
    // round 3: every event plane at once (an event the item does not have repeats its last one)
    auto plane_of = [&](uint32_t evw) { return Pl + ((size_t)(evw & 255u) * 4 + (size_t)__builtin_ctz(((evw >> 8) & 15u) | 16u)) * nw32; };
    {
        uint32_t pl[kPatchEvents];
#pragma unroll
        for (int u = 0; u < kPatchEvents; u++) {
            const int e = n_ev ? (u < n_ev ? u : n_ev - 1) : 0;
            const uint32_t evw = n_ev ? (uint32_t)__builtin_amdgcn_readlane((int)evv, e) : (1u << 8);
            pl[u] = plane_of(evw)[0];
        }
#pragma unroll
        for (int u = 0; u < kPatchEvents; u++) stash[u * 64 + lane] = pl[u];           // (a lane reads back its own words only: no barrier)
    }
    // the walk down the chain: the events of step s, then member s is counted (as in eval_chain_block)
    int e = 0;
#pragma unroll
    for (int s = 0; s < CC; s++) {
        if (s >= n_steps) break;
        if (s > 0) {
#pragma unroll 1
            while (e < n_ev) {
                const uint32_t evw = e < 64 ? (uint32_t)__builtin_amdgcn_readlane((int)evv, e) : ev[e];
                if ((int)(evw >> 16) != s) break;
                const uint32_t cur = e < kPatchEvents ? stash[e * 64 + lane] : plane_of(evw)[0];
                const uint32_t j = evw & 255u;
                count_plane<LV>(t1[0], t2[0], t3[0], t4[0], cur);
                if (((A.sF | A.sR) >> j) & 1u) {
                    const uint32_t fF = ((A.sF >> j) & 1u) ? 0xFFFFFFFFu : 0u, fR = ((A.sR >> j) & 1u) ? 0xFFFFFFFFu : 0u;
                    sf[0] = __builtin_amdgcn_bitop3_b32(sf[0], cur, fF, kLutOrAnd);
                    sr[0] = __builtin_amdgcn_bitop3_b32(sr[0], cur, fR, kLutOrAnd);
                }
                e++;
            }
        }
        const uint32_t far = LV == 1 ? t1[0] : (LV == 2 ? t2[0] : (LV == 3 ? t3[0] : t4[0]));
        const uint32_t nP = (uint32_t)__popc(valid & ~t1[0]);
        const uint32_t nF = (uint32_t)__popc(__builtin_amdgcn_bitop3_b32(valid, far, sf[0], kLutAndNotNot));
        const uint32_t nR = (uint32_t)__popc(__builtin_amdgcn_bitop3_b32(valid, far, sr[0], kLutAndNotNot));
        acc[s] = nP | (nF << 10) | (nR << 20);
    }
    uint32_t accP[CC], accF[CC], accR[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) { accP[c] = acc[c] & 1023u; accF[c] = (acc[c] >> 10) & 1023u; accR[c] = acc[c] >> 20; }
    wave_commit<GW>(accP, accF, accR, row, A.cand_out + it.cand0, A.out, negative);
}

}  // namespace mp
