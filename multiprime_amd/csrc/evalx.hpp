// evalx.hpp — part of eval.hip: the nested-chain evaluation for what the tuned kernels of rounds 1-5 do not hold — primers of 32..63
// bases (their items hold 32 positions) and more than three tolerated mismatches (they carry four counter levels).  [r6]
//
// Same formulation as eval_chain_kernel (chainbody.hpp): a workgroup owns (chain item, row slice); the most degenerate member of the
// chain is scored once over the k one-hot column planes of its window — saturating thermometer counters, one v_bitop3 per level and
// 32-row word —, every later member adds one column plane per base it loses (an event), and a member's three counts are popcounts.
// What is general here: LV = v + 1 counter levels up to 6 as an array, 64 positions (position masks and strict masks are 64-bit, the
// symbols of the first member sit in eight words read through scalar loads), chains of at most 8 members (a longer refinement run is
// cut into several items by the host).  Patch rows (edge-gap repaired, IUPAC expansions) ride in the same launch on their patch planes,
// one wave per unit, as there.  Until round 6 these cases ran on eval_kernel — one row per lane, window words derived from the planes,
// every candidate compared symbol by symbol: ~40 instructions per (candidate, row) against ~0.5 here.
#pragma once

#include "bitslice.hpp"
#include "common.hpp"

namespace mp {

struct ChainItemX {
    int32_t win, cand0, n_steps, ev0, n_ev, pad;
    uint32_t sym[8];                   // nibble j & 7 of word j >> 3 = symbol of the first (most degenerate) member at position j
    uint64_t pos1, pos2, pos4;         // positions where that symbol has one / two / more bases
};

struct EvalXArgs {
    const unsigned long long *cols;    // [n_cols][4][nw]
    const unsigned long long *excl;    // [W][nw]
    int nw, p0, k, v;
    const ChainItemX *items;
    const uint32_t *events;            // position | lost base (one-hot) << 8 | step << 16, ascending by step
    const int32_t *cand_out;
    uint64_t sF, sR;
    unsigned long long *out;
    BlockMap map;
    PatchArgs patch;                   // the first patch.n_blocks workgroups run the patch units (a wave each)
};

template <int LV, int GW>
__device__ __forceinline__ void x_count_unmatched(uint32_t (&t)[LV][GW], const uint32_t (&m)[GW]) {
#pragma unroll
    for (int i = 0; i < GW; i++) {
#pragma unroll
        for (int l = LV - 1; l >= 1; l--) t[l][i] = __builtin_amdgcn_bitop3_b32(t[l][i], t[l - 1][i], m[i], kLutOrAndNot);
        t[0][i] = __builtin_amdgcn_bitop3_b32(t[0][i], m[i], m[i], kLutOrNot);
    }
}
template <int LV, int GW>
__device__ __forceinline__ void x_count_plane(uint32_t (&t)[LV][GW], const uint32_t (&d)[GW]) {
#pragma unroll
    for (int i = 0; i < GW; i++) {
#pragma unroll
        for (int l = LV - 1; l >= 1; l--) t[l][i] = __builtin_amdgcn_bitop3_b32(t[l][i], t[l - 1][i], d[i], kLutOrAnd);
        t[0][i] |= d[i];
    }
}

// the first member over the positions in `rem` (NB = bases of its symbol there: 1, 2, or 4 = three or four), D positions in flight
template <int LV, int GW, int D, int NB>
__device__ __forceinline__ void x_first_pass(uint64_t rem, const uint32_t *Pw, size_t nw32, const uint32_t *sym, uint64_t sF, uint64_t sR,
                                             uint32_t (&t)[LV][GW], uint32_t (&sf)[GW], uint32_t (&sr)[GW]) {
#pragma unroll 1
    while (rem) {
        int js[D]; bool has[D];
        uint32_t ld[D][NB][GW], sys[D];
#pragma unroll
        for (int u = 0; u < D; u++) {
            has[u] = rem != 0ull;
            js[u] = has[u] ? (int)__builtin_ctzll(rem) : js[0];
            rem &= rem - 1ull;
            const int j = js[u];
            const uint32_t sy = (sym[j >> 3] >> (4 * (j & 7))) & 15u;
            sys[u] = sy;
            const uint32_t *P = Pw + (size_t)j * 4 * nw32;
            if constexpr (NB == 4) {
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
                    if constexpr (NB == 2) ld[u][1][i] = P[b1 * nw32 + i];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < D; u++) {
            if (!has[u]) break;
            const int j = js[u];
            uint32_t m[GW];
            if constexpr (NB == 4) {
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
                for (int i = 0; i < GW; i++) {
                    if constexpr (NB == 2) m[i] = ld[u][0][i] | ld[u][1][i];
                    else m[i] = ld[u][0][i];
                }
            }
            x_count_unmatched<LV, GW>(t, m);
            if (((sF | sR) >> j) & 1ull) {
                const uint32_t fF = ((sF >> j) & 1ull) ? 0xFFFFFFFFu : 0u, fR = ((sR >> j) & 1ull) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    sf[i] = __builtin_amdgcn_bitop3_b32(sf[i], m[i], fF, kLutOrNotAnd);
                    sr[i] = __builtin_amdgcn_bitop3_b32(sr[i], m[i], fR, kLutOrNotAnd);
                }
            }
        }
    }
}

template <int LV, int GW, int D>
__global__ __launch_bounds__(kBlock) void eval_chain_x_kernel(const EvalXArgs A) {
    static_assert(GW <= 8 && 32 * GW < 1024, "plane rows are padded to multiples of 8 words; three 10-bit counts per register");
    constexpr int CC = 8;
    __shared__ uint32_t s_part[kBlock / 64][12];
    const unsigned bid = blockIdx.x;
    const bool on_patch = (int)bid < A.patch.n_blocks;
    int slice, item, word0;
    if (on_patch) {                                    // a wave per patch unit: everything below is wave-uniform
        const int unit = __builtin_amdgcn_readfirstlane((int)bid * (kBlock / 64) + (int)(threadIdx.x >> 6));
        item = unit / A.patch.per_item;
        slice = unit % A.patch.per_item;
        if (item >= A.map.n_items) return;
        word0 = (slice * 64 + (int)(threadIdx.x & 63)) * GW;
    } else {
        if (!map_block(A.map, bid - (unsigned)A.patch.n_blocks, slice, item)) return;
        word0 = (slice * kBlock + threadIdx.x) * GW;
    }
    const ChainItemX *it = A.items + item;             // (uniform address: scalar loads)
    const int win = it->win, n_steps = it->n_steps, n_ev = it->n_ev;
    const WordTile T = !on_patch ? column_tile(A.cols, A.excl, A.nw, A.p0, win, word0) : patch_tile(A.patch, win, word0);
    if (on_patch && slice * 64 * GW >= (int)T.stride) return;               // nothing of this window's patch planes left for the wave
    const size_t nw32 = T.stride;
    uint32_t acc[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) acc[c] = 0;
    if (T.live) {
        uint32_t t[LV][GW], sf[GW], sr[GW];
#pragma unroll
        for (int i = 0; i < GW; i++) {
            sf[i] = sr[i] = 0;
#pragma unroll
            for (int l = 0; l < LV; l++) t[l][i] = 0;
        }
        const uint32_t *Pw = T.planes;
        x_first_pass<LV, GW, D, 1>(it->pos1, Pw, nw32, it->sym, A.sF, A.sR, t, sf, sr);
        x_first_pass<LV, GW, (D + 1) / 2, 2>(it->pos2, Pw, nw32, it->sym, A.sF, A.sR, t, sf, sr);
        x_first_pass<LV, GW, (D + 3) / 4, 4>(it->pos4, Pw, nw32, it->sym, A.sF, A.sR, t, sf, sr);
        uint32_t valid[GW];
#pragma unroll
        for (int i = 0; i < GW; i++) valid[i] = T.mask[i] ^ T.mask_flip;
        // walk down the chain: the events of step s, then member s is counted
        const uint32_t *ev = A.events + it->ev0;
        int e = 0;
        uint32_t evw = n_ev ? ev[0] : (1u << 8);
        uint32_t cur[GW];
        {
            const uint32_t *P = Pw + ((size_t)(evw & 255u) * 4 + (size_t)__builtin_ctz(((evw >> 8) & 15u) | 16u)) * nw32;
#pragma unroll
            for (int i = 0; i < GW; i++) cur[i] = P[i];
        }
#pragma unroll
        for (int s = 0; s < CC; s++) {
            if (s >= n_steps) break;
            if (s > 0) {
#pragma unroll 1
                while (e < n_ev && (int)(evw >> 16) == s) {
                    e++;
                    const uint32_t evn = e < n_ev ? ev[e] : evw;             // the plane of the next event is on its way
                    uint32_t nxt[GW];
                    {
                        const uint32_t *P = Pw + ((size_t)(evn & 255u) * 4 + (size_t)__builtin_ctz(((evn >> 8) & 15u) | 16u)) * nw32;
#pragma unroll
                        for (int i = 0; i < GW; i++) nxt[i] = P[i];
                    }
                    const uint32_t j = evw & 255u;
                    x_count_plane<LV, GW>(t, cur);
                    if (((A.sF | A.sR) >> j) & 1ull) {
                        const uint32_t fF = ((A.sF >> j) & 1ull) ? 0xFFFFFFFFu : 0u, fR = ((A.sR >> j) & 1ull) ? 0xFFFFFFFFu : 0u;
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
                const uint32_t far = t[LV - 1][i];                           // more than v mismatches
                nP += __popc(valid[i] & ~t[0][i]);
                nF += __popc(__builtin_amdgcn_bitop3_b32(valid[i], far, sf[i], kLutAndNotNot));
                nR += __popc(__builtin_amdgcn_bitop3_b32(valid[i], far, sr[i], kLutAndNotNot));
            }
            acc[s] = nP | (nF << 10) | (nR << 20);
        }
    }
    uint32_t accP[CC], accF[CC], accR[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) { accP[c] = acc[c] & 1023u; accF[c] = (acc[c] >> 10) & 1023u; accR[c] = acc[c] >> 20; }
    if (on_patch) wave_commit<GW>(accP, accF, accR, s_part[threadIdx.x >> 6], A.cand_out + it->cand0, A.out, false);
    else block_commit<GW>(accP, accF, accR, s_part, A.cand_out + it->cand0, A.out);
}

}  // namespace mp
