// bitslice.hpp — device helpers shared by the bit-sliced evaluation kernels (eval.hip, evaltile.hip): v_bitop3 truth tables,
// saturating bit-sliced mismatch counters, DPP wave sums and the commit of a workgroup's / wave's 8 x 3 popcount totals.
#pragma once

#include "common.hpp"

namespace mp {

// truth table of v_bitop3_b32 D = f(S0,S1,S2): bit (S0<<2 | S1<<1 | S2) of the immediate
template <typename F>
constexpr int make_lut(F f) {
    int t = 0;
    for (int i = 0; i < 8; i++)
        if (f((i >> 2) & 1, (i >> 1) & 1, i & 1)) t |= 1 << i;
    return t;
}
constexpr int kLutOrAnd = make_lut([](int a, int b, int c) { return a | (b & c); });            // S0 | (S1 & S2)
constexpr int kLutOrAndNot = make_lut([](int a, int b, int c) { return a | (b & (c ^ 1)); });   // S0 | (S1 & ~S2)
constexpr int kLutOrNot = make_lut([](int a, int b, int) { return a | (b ^ 1); });              // S0 | ~S1
constexpr int kLutOrNotAnd = make_lut([](int a, int b, int c) { return a | ((b ^ 1) & c); });   // S0 | (~S1 & S2)
constexpr int kLutAndNotNot = make_lut([](int a, int b, int c) { return a & (b ^ 1) & (c ^ 1); });   // S0 & ~S1 & ~S2
constexpr int kLutOr3 = make_lut([](int a, int b, int c) { return a | b | c; });
constexpr int kLutAndOr = make_lut([](int a, int b, int c) { return (a & b) | c; });            // (S0 & S1) | S2
static_assert(kLutOrAnd == 0xF8 && kLutAndNotNot == 0x10, "v_bitop3 truth tables");

// add one "mismatch where NOT m" plane to saturating counters
template <int LV>
__device__ __forceinline__ void count_unmatched(uint32_t &t1, uint32_t &t2, uint32_t &t3, uint32_t &t4, uint32_t m) {
    if (LV >= 4) t4 = __builtin_amdgcn_bitop3_b32(t4, t3, m, kLutOrAndNot);
    if (LV >= 3) t3 = __builtin_amdgcn_bitop3_b32(t3, t2, m, kLutOrAndNot);
    if (LV >= 2) t2 = __builtin_amdgcn_bitop3_b32(t2, t1, m, kLutOrAndNot);
    t1 = __builtin_amdgcn_bitop3_b32(t1, m, m, kLutOrNot);
}
// add one "mismatch where d" plane
template <int LV>
__device__ __forceinline__ void count_plane(uint32_t &t1, uint32_t &t2, uint32_t &t3, uint32_t &t4, uint32_t d) {
    if (LV >= 4) t4 = __builtin_amdgcn_bitop3_b32(t4, t3, d, kLutOrAnd);
    if (LV >= 3) t3 = __builtin_amdgcn_bitop3_b32(t3, t2, d, kLutOrAnd);
    if (LV >= 2) t2 = __builtin_amdgcn_bitop3_b32(t2, t1, d, kLutOrAnd);
    t1 |= d;
}

// Sum over the 64 lanes of a wave with DPP adds only (no LDS); the total ends up in lane 63.  All lanes active.
__device__ __forceinline__ uint32_t wave_sum_lane63(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true);    // row_half_mirror
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, true);    // row_mirror: every lane = its row's sum
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast31 into rows 2 and 3
    return x;
}

// Block totals of the 8 x 3 per-thread popcounts without LDS round trips: two 16-bit counts per word (a wave's
// sum is at most 64 * 32 * GW), six DPP adds per word leave the wave total in lane 63, which parks it in LDS for
// the final 24 threads; those add the block's share to the global counters (F_mis = F_raw - perfect).
template <int GW>
__device__ __forceinline__ void block_commit(const uint32_t (&accP)[8], const uint32_t (&accF)[8], const uint32_t (&accR)[8],
                                             uint32_t (&s_part)[kBlock / 64][12], const int32_t *cand_out, unsigned long long *out) {
    constexpr int CC = 8;
    static_assert(64 * 32 * GW < 65536, "packed wave sums must fit 16 bits");
    uint32_t vals[3 * CC];
#pragma unroll
    for (int c = 0; c < CC; c++) { vals[3 * c] = accP[c]; vals[3 * c + 1] = accF[c]; vals[3 * c + 2] = accR[c]; }
    uint32_t tot[3 * CC / 2];
#pragma unroll
    for (int q = 0; q < 3 * CC / 2; q++) tot[q] = wave_sum_lane63(vals[2 * q] | (vals[2 * q + 1] << 16));
    if ((threadIdx.x & 63) == 63) {
#pragma unroll
        for (int q = 0; q < 3 * CC / 2; q++) s_part[threadIdx.x >> 6][q] = tot[q];
    }
    __syncthreads();
    if (threadIdx.x < 3 * CC) {
        const int c = threadIdx.x / 3, r = threadIdx.x % 3;
        uint32_t mine = 0, perfect = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) {
            mine += (s_part[w][threadIdx.x >> 1] >> (16 * (threadIdx.x & 1))) & 0xFFFFu;
            perfect += (s_part[w][(3 * c) >> 1] >> (16 * ((3 * c) & 1))) & 0xFFFFu;
        }
        const uint32_t val = r ? mine - perfect : mine;
        const int oc = cand_out[c];
        if (oc >= 0 && val) atomicAdd(&out[(size_t)oc * 3 + r], (unsigned long long)val);
    }
}

// The same for ONE wave (a patch unit, a wave of the tile kernel): lane 63's totals go through the wave's own LDS row
// (`row`, 12 words), lanes 0..23 add them.
template <int GW>
__device__ __forceinline__ void wave_commit(const uint32_t (&accP)[8], const uint32_t (&accF)[8], const uint32_t (&accR)[8],
                                            uint32_t *row, const int32_t *cand_out, unsigned long long *out, bool negative = false) {
    constexpr int CC = 8;
    uint32_t vals[3 * CC];
#pragma unroll
    for (int c = 0; c < CC; c++) { vals[3 * c] = accP[c]; vals[3 * c + 1] = accF[c]; vals[3 * c + 2] = accR[c]; }
    uint32_t tot[3 * CC / 2];
#pragma unroll
    for (int q = 0; q < 3 * CC / 2; q++) tot[q] = wave_sum_lane63(vals[2 * q] | (vals[2 * q + 1] << 16));
    const int lane = threadIdx.x & 63;
    if (lane == 63) {
#pragma unroll
        for (int q = 0; q < 3 * CC / 2; q++) row[q] = tot[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 3 * CC) {
        const int c = lane / 3, r = lane % 3;
        const uint32_t mine = (row[lane >> 1] >> (16 * (lane & 1))) & 0xFFFFu;
        const uint32_t perfect = (row[(3 * c) >> 1] >> (16 * ((3 * c) & 1))) & 0xFFFFu;
        const uint32_t val = r ? mine - perfect : mine;
        const int oc = cand_out[c];
        if (oc >= 0 && val) atomicAdd(&out[(size_t)oc * 3 + r], negative ? 0ull - (unsigned long long)val : (unsigned long long)val);
    }
}

// Patch planes and the (item, row slice) -> workgroup mapping shared by eval.hip and evalprog.hip (see eval.hip)
struct PatchArgs {
    const uint32_t *pplanes;
    const uint32_t *pvalid;
    const PatchWin *pwin;      // [W]
    int per_item;              // patch units per item (0: no patch rows anywhere); a unit = one WAVE in the evaluation kernels
                               // (4 items' patch planes per workgroup: they are a few hundred rows each), one workgroup in window_stats
    int n_blocks;              // workgroups holding the units, rounded up to a multiple of 8; they come first in the grid
    // Sliding evaluation (evalslide.hip): its column pass counts EVERY row of a window (no exclusion words), so the rows of the patch
    // list are counted there as plain column slices although their k-mer is the repaired one.  A second run of neg_blocks workgroups
    // (blockIdx in [n_blocks, n_blocks + neg_blocks)) evaluates exactly those plain slices, on planes of their own, for the items that
    // slide (EvalChainArgs::neg_items) and SUBTRACTS what it finds.
    const uint32_t *qplanes;
    const uint32_t *qvalid;
    int neg_blocks;            // 0: no second run
};

struct WordTile {
    const uint32_t *planes;    // plane (j, base) of the tile's first word: planes + (j * 4 + base) * stride
    const uint32_t *mask;      // validity words: valid = mask[i] ^ mask_flip
    uint32_t stride, mask_flip;
    bool live;
};

// words [word0, word0 + GW) of the column planes of window `win` (valid = not excluded) ...
__device__ __forceinline__ WordTile column_tile(const unsigned long long *cols, const unsigned long long *excl, int nw, int p0,
                                                int win, int word0) {
    const size_t nw32 = (size_t)nw * 2;
    WordTile t;
    t.planes = reinterpret_cast<const uint32_t *>(cols) + ((size_t)(p0 + win) * 4) * nw32 + word0;
    t.mask = reinterpret_cast<const uint32_t *>(excl) + (size_t)win * nw32 + word0;
    t.stride = (uint32_t)nw32;
    t.mask_flip = 0xFFFFFFFFu;
    t.live = word0 < (int)nw32;                        // nw32 % GW == 0 (n_pad % 256 == 0, GW <= 8)
    return t;
}
// ... or of the window's patch planes
__device__ __forceinline__ WordTile patch_tile(const PatchArgs &P, int win, int word0) {
    const PatchWin pw = P.pwin[win];
    WordTile t;
    t.planes = P.pplanes + pw.poff + word0;
    t.mask = P.pvalid + pw.voff + word0;
    t.stride = (uint32_t)pw.npw;
    t.mask_flip = 0u;
    t.live = word0 < pw.npw;                           // npw % 8 == 0
    return t;
}

// ... or of the plain column slices of the window's patch-list rows (the subtracting run)
__device__ __forceinline__ WordTile plain_tile(const PatchArgs &P, int win, int word0) {
    const PatchWin pw = P.pwin[win];
    WordTile t;
    t.planes = P.qplanes + pw.poff + word0;
    t.mask = P.qvalid + pw.voff + word0;
    t.stride = (uint32_t)pw.npw;
    t.mask_flip = 0u;
    t.live = word0 < pw.npw;
    return t;
}

struct BlockMap { int ny, ny_pad, n_items, per_band; };       // see map_block

// XCD-aware block mapping shared by both kernels: workgroup b runs on XCD b % 8 (observed dispatch order) and every
// XCD has its own L2, so the (item, row slice) grid is laid out to let consecutive windows re-read their k-1 shared
// columns from ONE L2.  With 8 or more slices (ny_pad a multiple of 8) slice = b % ny_pad: an XCD owns slices.  With
// fewer (ny_pad = 1, 2, 4) an XCD owns one slice and one of 8 / ny_pad contiguous BANDS of items — otherwise two
// XCDs would walk the same slice with interleaved windows and each pull every column from HBM.
__device__ __forceinline__ bool map_block(const BlockMap &M, unsigned b, int &slice, int &idx) {
    if (M.ny_pad >= 8) {
        slice = (int)(b % (unsigned)M.ny_pad);
        idx = (int)(b / (unsigned)M.ny_pad);
    } else {
        const int xcd = (int)(b & 7u);
        slice = xcd % M.ny_pad;
        idx = (xcd / M.ny_pad) * M.per_band + (int)(b >> 3);
        if ((int)(b >> 3) >= M.per_band) return false;
    }
    return slice < M.ny && idx < M.n_items;
}

}  // namespace mp
