// evalprog.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of include/mprime.h.
// Candidate x sequence coverage evaluation of nested refinement chains (mis_primer_check + Y_distance, V20:1103-1130, 229-233):
// eval_prog_kernel — eval_chain_kernel's arithmetic (eval.hip: saturating bit-sliced mismatch counters over the one-hot column
// planes, one pass for the most degenerate member, one plane per refinement step) driven by a per-item FETCH PROGRAM the host
// writes at upload time instead of by symbol words decoded on the scalar unit:
//   * one 32-bit entry per plane fetch — plane row (window position * 4 + base), strict-position flags, the chain step of an event —
//     one entry per lane of a few registers, broadcast with v_readlane when its turn comes;
//   * the fetch itself is a buffer load: resource = the item's first plane row, scalar offset = entry row * row bytes, vector
//     offset = the lane's word offset, a constant of the wave.  No vector address arithmetic (eval_chain_kernel: two 64-bit adds
//     per load instruction) and two scalar instructions per fetch instead of ten;
//   * the 24 output slots of an item sit in the program too, in the lanes that commit them.
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

// GW consecutive words of plane row `entry & kRowMask` at the lane's word offset
template <int GW>
__device__ __forceinline__ void fetch_row(__amdgpu_buffer_rsrc_t rsrc, int voff, uint32_t row_bytes, uint32_t entry, uint32_t (&d)[GW]) {
    const int soff = (int)((entry & kRowMask) * row_bytes);
    if constexpr (GW >= 4) {
#pragma unroll
        for (int q = 0; q < GW / 4; q++) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 16 * q, soff, 0);
            d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
        }
    } else if constexpr (GW == 2) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
        d[0] = v.x; d[1] = v.y;
    } else {
        d[0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0);
    }
}

template <int LV, int GW, int D>
__global__ __launch_bounds__(kBlock) void eval_prog_kernel(const EvalProgArgs A) {
    static_assert(GW <= 8, "plane rows are padded to multiples of 8 words");
    constexpr int CC = 8;
    __shared__ uint32_t s_part[kBlock / 64][3 * CC / 2];
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
    const bool live = word0 < (int)row_words;             // rows are padded to multiples of 8 words, GW divides 8
    const int w_safe = live ? word0 : 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(base), 0, 0x7FFFFFFF, 0x00020000);
    const int voff = w_safe * 4;
    const uint32_t row_bytes = row_words * 4u;
    const int n_steps = (int)lane_of(P.r[0], 34), n_ev = (int)lane_of(P.r[0], 35), n_fp = (int)lane_of(P.r[0], 36);
    uint32_t valid[GW];
    {
        uint32_t raw[GW];
#pragma unroll
        for (int i = 0; i < GW; i++) raw[i] = mask[w_safe + i];
#pragma unroll
        for (int i = 0; i < GW; i++) valid[i] = live ? (raw[i] ^ flip) : 0u;
    }
    auto fetch = [&](uint32_t entry, uint32_t (&d)[GW]) { fetch_row<GW>(rsrc, voff, row_bytes, entry, d); };
    auto pass_entry = [&](int q) -> uint32_t { const uint32_t a = lane_of(P.r[1], q & 63), b = lane_of(P.r[2], q & 63); return q < 64 ? a : b; };
    auto event_entry = [&](int q) -> uint32_t { const uint32_t a = lane_of(P.r[3], q & 63), b = lane_of(P.r[4], q & 63); return q < 64 ? a : b; };
    uint32_t t1[GW], t2[GW], t3[GW], t4[GW], sf[GW], sr[GW];
#pragma unroll
    for (int i = 0; i < GW; i++) t1[i] = t2[i] = t3[i] = t4[i] = sf[i] = sr[i] = 0;
    // (1) the first (most degenerate) member over all k positions: one fetch per base of a position's symbol, D fetches in flight
    // ALL the time — a consumed buffer is refilled with the fetch D entries ahead at once, so only the first fetch of an item
    // pays a full round trip (eval_chain_kernel drains its D loads before it asks for the next D).
    {
        uint32_t en[D], ld[D][GW], hold[GW];
#pragma unroll
        for (int i = 0; i < GW; i++) hold[i] = 0u;
#pragma unroll
        for (int u = 0; u < D; u++) {
            en[u] = pass_entry(min(u, n_fp - 1));
            fetch(en[u], ld[u]);
        }
#pragma unroll 1
        for (int q0 = 0; q0 < n_fp; q0 += D) {
#pragma unroll
            for (int u = 0; u < D; u++) {
                if (q0 + u >= n_fp) break;
                const uint32_t e = en[u];
                if (e & (kMore | kCont)) {                  // a position with several bases: their planes are OR-ed first
#pragma unroll
                    for (int i = 0; i < GW; i++) hold[i] = (e & kCont) ? (hold[i] | ld[u][i]) : ld[u][i];
                }
                if (!(e & kMore)) {
                    if (e & kCont) {
#pragma unroll
                        for (int i = 0; i < GW; i++) ld[u][i] = hold[i];
                    }
#pragma unroll
                    for (int i = 0; i < GW; i++) count_unmatched<LV>(t1[i], t2[i], t3[i], t4[i], ld[u][i]);
                    if (e & (kStrictF | kStrictR)) {
                        const uint32_t fF = (e & kStrictF) ? 0xFFFFFFFFu : 0u, fR = (e & kStrictR) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                        for (int i = 0; i < GW; i++) {
                            sf[i] = __builtin_amdgcn_bitop3_b32(sf[i], ld[u][i], fF, kLutOrNotAnd);
                            sr[i] = __builtin_amdgcn_bitop3_b32(sr[i], ld[u][i], fR, kLutOrNotAnd);
                        }
                    }
                }
                en[u] = pass_entry(min(q0 + u + D, n_fp - 1));      // past the end: the last entry again, dropped
                fetch(en[u], ld[u]);
            }
        }
    }
    // (2) walk down the chain: the events of step s (one lost base each: its plane IS the increment), then member s is counted.
    // The three counts of a member are at most 32 * GW <= 256 per thread: one register per member, 10 bits each.
    static_assert(32 * GW < 1024, "three counts per register need 10 bits each");
    uint32_t acc[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) acc[c] = 0;
    int e = 0;
    uint32_t evw = n_ev ? event_entry(0) : 0u;
    uint32_t cur[GW];
    fetch(evw, cur);
#pragma unroll
    for (int s = 0; s < CC; s++) {
        if (s >= n_steps) break;
        if (s > 0) {
#pragma unroll 1
            while (e < n_ev && (int)((evw >> 24) & 15u) == s) {
                e++;
                const uint32_t evn = e < n_ev ? event_entry(e) : evw;          // the plane of the next event is on its way
                uint32_t nxt[GW];
                fetch(evn, nxt);
#pragma unroll
                for (int i = 0; i < GW; i++) count_plane<LV>(t1[i], t2[i], t3[i], t4[i], cur[i]);
                if (evw & (kStrictF | kStrictR)) {
                    const uint32_t fF = (evw & kStrictF) ? 0xFFFFFFFFu : 0u, fR = (evw & kStrictR) ? 0xFFFFFFFFu : 0u;
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
    // commit: wave totals by DPP (bitslice.hpp), the output slots of the 24 counters come from lanes 0-23 of program register 0
    uint32_t tot[3 * CC / 2];
#pragma unroll
    for (int q = 0; q < 3 * CC / 2; q++) {
        const int a = 2 * q, b = 2 * q + 1;
        const uint32_t va = (acc[a / 3] >> (10 * (a % 3))) & 1023u, vb = (acc[b / 3] >> (10 * (b % 3))) & 1023u;
        tot[q] = wave_sum_lane63(va | (vb << 16));
    }
    const int wv = (int)(threadIdx.x >> 6);
    if (lane == 63) {
#pragma unroll
        for (int q = 0; q < 3 * CC / 2; q++) s_part[wv][q] = tot[q];
    }
    if (on_patch) {
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
                         int k, uint32_t sF, uint32_t sR, std::vector<uint32_t> &prog) {
    prog.assign(chains.size() * (size_t)(kProgRegs * 64), 0u);
    for (size_t i = 0; i < chains.size(); i++) {
        const ChainItem &ch = chains[i];
        auto entry = [&](int j, int base) {
            return (uint32_t)(j * 4 + base) | (((sF >> j) & 1u) ? kStrictF : 0u) | (((sR >> j) & 1u) ? kStrictR : 0u);
        };
        auto sym = [&](int j) { return (ch.sym[j >> 3] >> (4 * (j & 7))) & 15u; };
        uint32_t *blk = prog.data() + i * (size_t)(kProgRegs * 64);
        int n_fp = 0;
        for (int pass = 1; pass <= 3; pass++)             // single-base positions first, then two bases, then the rest
            for (int j = 0; j < k; j++) {
                const uint32_t mask = pass == 1 ? ch.pos1 : (pass == 2 ? ch.pos2 : ch.pos4);
                if (!((mask >> j) & 1u)) continue;
                uint32_t sy = sym(j);
                bool first = true;
                while (sy) {
                    const int base = __builtin_ctz(sy);
                    sy &= sy - 1u;
                    blk[64 + n_fp++] = entry(j, base) | (sy ? kMore : 0u) | (first ? 0u : kCont);
                    first = false;
                }
            }
        for (int q = 0; q < ch.n_ev; q++) {
            const uint32_t ev = events[(size_t)ch.ev0 + (size_t)q];          // position | lost base (one-hot) << 8 | step << 16
            blk[192 + q] = entry((int)(ev & 255u), __builtin_ctz((ev >> 8) & 15u)) | ((ev >> 16) << 24);
        }
        for (int t = 0; t < 24; t++) blk[t] = (uint32_t)cand_out[(size_t)ch.cand0 + (size_t)(t / 3)];
        const uint32_t head[5] = {(uint32_t)ch.win, (uint32_t)ch.cand0, (uint32_t)ch.n_steps, (uint32_t)ch.n_ev, (uint32_t)n_fp};
        for (int q = 0; q < 5; q++) blk[32 + q] = head[q];
    }
}

// shape: words per thread x fetches in flight, as eval_chain_kernel's MP_EVAL_CHAIN table
int launch_eval_prog(mp_ctx *c, int shape, const BlockMap &bm, const PatchArgs &pa, unsigned grid, unsigned long long *device_out) {
#define PROG_ROW(LV) {eval_prog_kernel<LV, 2, 6>, eval_prog_kernel<LV, 2, 3>, eval_prog_kernel<LV, 2, 9>, eval_prog_kernel<LV, 4, 3>, \
                      eval_prog_kernel<LV, 4, 6>, eval_prog_kernel<LV, 1, 6>, eval_prog_kernel<LV, 8, 2>, eval_prog_kernel<LV, 8, 4>, \
                      eval_prog_kernel<LV, 8, 1>}
    static const ProgFn fn[4][9] = {PROG_ROW(1), PROG_ROW(2), PROG_ROW(3), PROG_ROW(4)};
#undef PROG_ROW
    EvalProgArgs a{reinterpret_cast<const uint32_t *>(c->cols), reinterpret_cast<const uint32_t *>(c->excl), c->n_pad / 32, c->p0,
                   c->chain_prog, device_out, bm, pa};
    hipLaunchKernelGGL(fn[c->v][shape], dim3(grid + (unsigned)pa.n_blocks), dim3(kBlock), 0, c->stream, a);
    HIPCK(c, hipGetLastError());
    return MP_OK;
}

}  // namespace mp
