// evaltile.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of include/mprime.h.
// Candidate x sequence coverage evaluation of nested refinement chains (mis_primer_check + Y_distance, V20:1103-1130, 229-233)
// with the column planes staged in LDS: eval_tile_kernel.
//
// eval_chain_kernel (eval.hip) gives every (chain item, row slice) its own workgroup and reads the k column planes of the item's
// window from L2; consecutive windows share k - 1 of their k columns, so every plane word travels from L2 to a CU k times
// (579 MB of L2 reads for 66 MB of planes at the bench shard, profiles/r02_counters.json).  Here a workgroup owns a TILE of
// 64 * GW row words (32 sequences each) and sweeps a BAND of consecutive chain items over it:
//   * the four base planes of the columns the band is working on live in a ring in LDS, [ring slot][base][tile word]; every column
//     is fetched from L2 / HBM ONCE per band (a whole 1 KiB line run per wave-load) instead of once per covering window;
//   * 16 waves work on 16 different items (windows) at once, all of them on the same row words: a lane keeps GW words of saturating
//     bit-sliced mismatch counters in registers, exactly as in eval_chain_kernel, and fetches "the sequences that carry base b in
//     window position j" with one ds_read_b128 whose address picks ring slot and base — the symbol decoding stays free;
//   * a ROUND = up to 16 items; between two rounds the columns that drop out of the sweep are overwritten by the ones that join it.
//     The global loads of the next round's columns are issued before the round's arithmetic and land in registers meanwhile; only
//     the LDS store between the two barriers is exposed.
// The host (plan_tiles) cuts the item list into bands and rounds so that a round's columns fit the ring; rows the column planes
// do not cover (edge-gap repair, ragged ends, IUPAC expansions: the patch planes) keep going through eval_chain_kernel's patch units.
#include "common.hpp"
#include "bitslice.hpp"
#include "evaltile.hpp"

using namespace mp;

namespace {

constexpr int kTileWaves = 16;
constexpr int kTileThreads = 64 * kTileWaves;
constexpr int kLdsBytes = 160 * 1024;
constexpr int kProgRegs = 7;                 // registers (64 entries each) of one item's fetch program
constexpr int kBatchCols = 16;                 // columns per staged load batch: 16 x 4 planes x tile words = GW 16-byte chunks per thread

constexpr uint32_t kOffMask = 0xFFFFFu, kStrictF = 1u << 20, kStrictR = 1u << 21, kLast = 1u << 28;     // fetch-program entry

struct TileArgs {
    const uint32_t *cols32;            // [n_cols][4][nw32] one-hot column planes
    const uint32_t *excl32;            // [W][nw32]
    int nw32;
    int ring_cols;                     // ring slots (columns)
    const uint32_t *prog;              // [chain item][kProgRegs][64] fetch programs
    unsigned long long *out;
    const TileRound *rounds;
    const TileBand *bands;
    int n_slices;
    unsigned long long *prof;          // MP_TILE_PROF: [workgroup][wave][kProfSlots] shader-clock stamps (null: none)
};
constexpr int kProfSlots = 64;

// GW consecutive words through one 4 / 8 / 16-byte access (ring reads: ds_read_b128 at GW = 4)
template <int GW>
__device__ __forceinline__ void vec_get(const char *p, uint32_t (&d)[GW]) {
    if constexpr (GW == 4) {
        const uint4 q = *reinterpret_cast<const uint4 *>(p);
        d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
    } else if constexpr (GW == 2) {
        const uint2 q = *reinterpret_cast<const uint2 *>(p);
        d[0] = q.x; d[1] = q.y;
    } else {
        d[0] = *reinterpret_cast<const uint32_t *>(p);
    }
}

// One wave, one chain item, GW words per lane; every plane comes out of the ring.  What to read is a per-item PROGRAM the host
// wrote (plan_tiles): one 32-bit entry per plane fetch — ring byte offset of (column slot, base), strict-position flags, the
// chain step for event entries.  A program is a fixed block of kProgRegs x 64 words, one register per 64 words, one entry per lane,
// broadcast with v_readlane when its turn comes: the wave spends no scalar work on decoding symbols, wrapping ring slots or
// shifting masks (eval_chain_kernel: ~10 SALU per fetch), and the block of the NEXT round's item is requested before this round's
// arithmetic, so a round starts with everything it needs in registers.
//   register 0: lanes 0-7 header {win, cand0, n_steps, n_ev, n1, n2, nq, -}; lanes 32-62 the n1 single-base positions
//   register 1: the n2 two-base positions, two entries each          registers 2, 3: the other positions, one entry per base
//   registers 4, 5: the events (one lost base each), ascending by step
//   register 6: lanes 0-23 the output slot (candidate index or -1) of counter lane / 3 — the commit's 24 lanes find theirs in place
struct Prog { uint32_t r[kProgRegs]; };

// (the program array ends with 16 empty blocks: the waves of a short last round read those)
__device__ __forceinline__ Prog prog_load(const uint32_t *prog, int item) {
    Prog P;
    const uint32_t *src = prog + (size_t)item * (kProgRegs * 64) + (threadIdx.x & 63);
#pragma unroll
    for (int q = 0; q < kProgRegs; q++) P.r[q] = src[q * 64];
    return P;
}
__device__ __forceinline__ uint32_t lane_of(uint32_t reg, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)reg, lane); }

template <int LV, int GW>
__device__ __forceinline__ void tile_item(const TileArgs &A, const char *ring, uint32_t *part_row, const Prog &P, const uint32_t (&valid)[GW]) {
    constexpr int CC = 8, D = 4;
    const int lane = threadIdx.x & 63;
    const int n_steps = (int)lane_of(P.r[0], 2), n_ev = (int)lane_of(P.r[0], 3);
    const int n1 = (int)lane_of(P.r[0], 4), n2 = (int)lane_of(P.r[0], 5), nq = (int)lane_of(P.r[0], 6);
    const char *mine = ring + lane * (GW * 4);
    auto fetch = [&](uint32_t entry, uint32_t (&d)[GW]) { vec_get<GW>(mine + (entry & kOffMask), d); };
    auto quad_entry = [&](int q) -> uint32_t { const uint32_t a = lane_of(P.r[2], q & 63), b = lane_of(P.r[3], q & 63); return q < 64 ? a : b; };
    auto event_entry = [&](int q) -> uint32_t { const uint32_t a = lane_of(P.r[4], q & 63), b = lane_of(P.r[5], q & 63); return q < 64 ? a : b; };
    uint32_t t1[GW], t2[GW], t3[GW], t4[GW], sf[GW], sr[GW];
#pragma unroll
    for (int i = 0; i < GW; i++) t1[i] = t2[i] = t3[i] = t4[i] = sf[i] = sr[i] = 0;
    auto strict_unmatched = [&](uint32_t entry, const uint32_t (&m)[GW]) {
        if (entry & (kStrictF | kStrictR)) {
            const uint32_t fF = (entry & kStrictF) ? 0xFFFFFFFFu : 0u, fR = (entry & kStrictR) ? 0xFFFFFFFFu : 0u;
#pragma unroll
            for (int i = 0; i < GW; i++) {
                sf[i] = __builtin_amdgcn_bitop3_b32(sf[i], m[i], fF, kLutOrNotAnd);
                sr[i] = __builtin_amdgcn_bitop3_b32(sr[i], m[i], fR, kLutOrNotAnd);
            }
        }
    };
    // (1) the first (most degenerate) member over all k positions: those whose symbol is one base, two bases, more
#pragma unroll 1
    for (int q0 = 0; q0 < n1; q0 += D) {
        uint32_t en[D], ld[D][GW];
#pragma unroll
        for (int u = 0; u < D; u++) {
            en[u] = lane_of(P.r[0], 32 + min(q0 + u, n1 - 1));
            fetch(en[u], ld[u]);
        }
#pragma unroll
        for (int u = 0; u < D; u++) {
            if (q0 + u >= n1) break;
#pragma unroll
            for (int i = 0; i < GW; i++) count_unmatched<LV>(t1[i], t2[i], t3[i], t4[i], ld[u][i]);
            strict_unmatched(en[u], ld[u]);
        }
    }
    {
        constexpr int D2 = 2;
#pragma unroll 1
        for (int q0 = 0; q0 < n2; q0 += D2) {
            uint32_t en[D2], la[D2][GW], lb[D2][GW];
#pragma unroll
            for (int u = 0; u < D2; u++) {
                const int q = 2 * min(q0 + u, n2 - 1);
                en[u] = lane_of(P.r[1], q);
                fetch(en[u], la[u]);
                fetch(lane_of(P.r[1], q + 1), lb[u]);
            }
#pragma unroll
            for (int u = 0; u < D2; u++) {
                if (q0 + u >= n2) break;
                uint32_t m[GW];
#pragma unroll
                for (int i = 0; i < GW; i++) m[i] = la[u][i] | lb[u][i];
#pragma unroll
                for (int i = 0; i < GW; i++) count_unmatched<LV>(t1[i], t2[i], t3[i], t4[i], m[i]);
                strict_unmatched(en[u], m);
            }
        }
    }
#pragma unroll 1
    for (int q = 0; q < nq;) {
        uint32_t m[GW], en;
#pragma unroll
        for (int i = 0; i < GW; i++) m[i] = 0u;
#pragma unroll 1
        do {
            en = quad_entry(q);
            q++;
            uint32_t pl[GW];
            fetch(en, pl);
#pragma unroll
            for (int i = 0; i < GW; i++) m[i] |= pl[i];
        } while (!(en & kLast) && q < nq);
#pragma unroll
        for (int i = 0; i < GW; i++) count_unmatched<LV>(t1[i], t2[i], t3[i], t4[i], m[i]);
        strict_unmatched(en, m);
    }
    // (2) walk down the chain: the events of step s (one lost base each: its plane IS the increment), then member s is counted
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
    uint32_t accP[CC], accF[CC], accR[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) { accP[c] = acc[c] & 1023u; accF[c] = (acc[c] >> 10) & 1023u; accR[c] = acc[c] >> 20; }
    // wave_commit (bitslice.hpp) with the output slots taken from the program instead of a late load
    {
        uint32_t tot[3 * CC / 2];
#pragma unroll
        for (int q = 0; q < 3 * CC / 2; q++) {
            const int a = 2 * q, b = 2 * q + 1;
            const uint32_t va = a % 3 == 0 ? accP[a / 3] : (a % 3 == 1 ? accF[a / 3] : accR[a / 3]);
            const uint32_t vb = b % 3 == 0 ? accP[b / 3] : (b % 3 == 1 ? accF[b / 3] : accR[b / 3]);
            tot[q] = wave_sum_lane63(va | (vb << 16));
        }
        if (lane == 63) {
#pragma unroll
            for (int q = 0; q < 3 * CC / 2; q++) part_row[q] = tot[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 3 * CC) {
            const int c = lane / 3, rr = lane % 3;
            const uint32_t mine_v = (part_row[lane >> 1] >> (16 * (lane & 1))) & 0xFFFFu;
            const uint32_t perfect = (part_row[(3 * c) >> 1] >> (16 * ((3 * c) & 1))) & 0xFFFFu;
            const uint32_t val = rr ? mine_v - perfect : mine_v;
            const int oc = (int)P.r[6];
            if (oc >= 0 && val) atomicAdd(&A.out[(size_t)oc * 3 + rr], (unsigned long long)val);
        }
    }
}

template <int LV, int GW>
__global__ __launch_bounds__(kTileThreads) void eval_tile_kernel(const TileArgs A) {
    constexpr int TW = 64 * GW, CS = 16 * TW, PS = 4 * TW;                  // tile words; bytes per ring column / per base plane
    constexpr int CPP = TW / 4;                                            // 16-byte chunks per (column, base)
    extern __shared__ __align__(16) char lds[];
    const int RC = A.ring_cols;
    uint32_t *s_part = reinterpret_cast<uint32_t *>(lds + (size_t)RC * CS);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    const int band_i = (int)blockIdx.x / A.n_slices, slice = (int)blockIdx.x % A.n_slices;
    const TileBand band = A.bands[band_i];
    const int word_base = slice * TW;
    const int w0 = word_base + lane * GW;
    const bool live = w0 < A.nw32;
    // a batch = kBatchCols columns x 4 bases x CPP chunks = GW chunks per thread
    uint4 st[GW];
    auto batch_load = [&](uint4 (&reg)[GW], int c0, int n) {
#pragma unroll
        for (int i = 0; i < GW; i++) {
            const int q = i * kTileThreads + (int)threadIdx.x, cp = q / CPP, ch = q % CPP;
            const int col = cp >> 2, base = cp & 3, w = word_base + ch * 4;
            reg[i] = uint4{0u, 0u, 0u, 0u};
            if (col < n && w < A.nw32)
                reg[i] = *reinterpret_cast<const uint4 *>(A.cols32 + ((size_t)(c0 + col) * 4 + (size_t)base) * (size_t)A.nw32 + (size_t)w);
        }
    };
    auto batch_store = [&](const uint4 (&reg)[GW], int slot_first, int n) {
#pragma unroll
        for (int i = 0; i < GW; i++) {
            const int q = i * kTileThreads + (int)threadIdx.x, cp = q / CPP, ch = q % CPP;
            const int col = cp >> 2, base = cp & 3;
            int s = slot_first + col;
            s = s >= RC ? s - RC : s;
            if (col < n) *reinterpret_cast<uint4 *>(lds + s * CS + base * PS + ch * 16) = reg[i];
        }
    };
    auto slot_after = [&](int slot, int n) { slot += n; return slot >= RC ? slot - RC : slot; };      // n <= kBatchCols <= RC
    // round descriptors travel through a register too (lane f = field f), two rounds ahead of the arithmetic
    auto round_load = [&](int r) -> uint32_t {
        return r < band.n_rounds && lane < 6 ? reinterpret_cast<const uint32_t *>(A.rounds + band.round0 + r)[lane] : 0u;
    };
    auto round_of = [&](uint32_t reg) -> TileRound {
        return TileRound{(int32_t)lane_of(reg, 0), (int32_t)lane_of(reg, 1), (int32_t)lane_of(reg, 2), (int32_t)lane_of(reg, 3),
                         (int32_t)lane_of(reg, 4), 0};
    };
    unsigned long long *prof = A.prof ? A.prof + ((size_t)blockIdx.x * kTileWaves + (size_t)wv) * kProfSlots : nullptr;
    int n_stamp = 0;
    auto stamp = [&]() { if (prof && lane == 0 && n_stamp < kProfSlots) prof[n_stamp] = clock64(); n_stamp++; };
    stamp();
    TileRound R = round_of(round_load(0)), Rn = round_of(round_load(1));
    Prog P = prog_load(A.prog, R.item0 + wv);
#pragma unroll 1
    for (int r = 0; r < band.n_rounds; r++) {
        stamp();                                           // [1 + 4r] arithmetic of the previous round done
        __syncthreads();                                   // every wave is done with the columns that drop out of the sweep
        stamp();                                           // [2 + 4r] through the barrier
        int c = R.new_c0, slot = R.new_slot0;
        if (r > 0 && c < R.new_c1) {                       // the batch requested before the previous round's arithmetic
            batch_store(st, slot, min(kBatchCols, R.new_c1 - c));
            c += kBatchCols;
            slot = slot_after(slot, kBatchCols);
        }
#pragma unroll 1
        while (c < R.new_c1) {                             // whatever that batch does not cover (a band's first round, a jump):
            uint4 pb[3][GW];                               // three batches in flight, their latencies overlap
#pragma unroll
            for (int b = 0; b < 3; b++) batch_load(pb[b], c + b * kBatchCols, max(0, min(kBatchCols, R.new_c1 - c - b * kBatchCols)));
#pragma unroll
            for (int b = 0; b < 3; b++) {
                batch_store(pb[b], slot, max(0, min(kBatchCols, R.new_c1 - c - b * kBatchCols)));
                slot = slot_after(slot, kBatchCols);
            }
            c += 3 * kBatchCols;
        }
        stamp();                                           // [3 + 4r] columns stored
        __syncthreads();
        stamp();                                           // [4 + 4r] through the second barrier
        // this item's validity words, then everything the NEXT round needs: its columns, its item's program, the descriptor after it
        const bool mine = wv < R.n_items;
        uint32_t valid[GW];
        {
            // one unconditional vector load from an address that always exists (idle waves and lanes past the last row word read
            // window 0 / word 0 and drop the result): a conditional load would be waited for on the spot
            const int win = (int)lane_of(P.r[0], 0);
            uint32_t raw[GW];
            vec_get<GW>(reinterpret_cast<const char *>(A.excl32 + (size_t)win * (size_t)A.nw32 + (size_t)(live ? w0 : 0)), raw);
            const uint32_t keep = (mine && live) ? 0xFFFFFFFFu : 0u;
#pragma unroll
            for (int i = 0; i < GW; i++) valid[i] = ~raw[i] & keep;
        }
        if (Rn.new_c0 < Rn.new_c1) batch_load(st, Rn.new_c0, min(kBatchCols, Rn.new_c1 - Rn.new_c0));
        const Prog Pn = prog_load(A.prog, Rn.item0 + wv);
        const uint32_t r2 = round_load(r + 2);
        if (mine) tile_item<LV, GW>(A, lds, s_part + wv * 12, P, valid);
        P = Pn;
        R = Rn;
        Rn = round_of(r2);
    }
    stamp();
}

typedef void (*TileFn)(const TileArgs);

}  // namespace

namespace mp {

int tile_words(int gw) { return 64 * gw; }

// ring slots a workgroup of `per_cu` co-resident workgroups can hold
int tile_ring_cols(int gw, int per_cu) {
    const int cs = 16 * 64 * gw;
    return (kLdsBytes / per_cu - kTileWaves * 12 * 4) / cs;
}

// Cuts the chain items (ascending windows, `win` relative to p0) into bands of consecutive items — one workgroup per (band, row
// slice) — and every band into rounds of at most 16 items whose columns fit the ring; writes every item's fetch program (tile_item).
// The caller guarantees chains of at most 8 members: 31 single-base positions, 62 pair entries, 124 other entries and 93 events
// are the most k <= 31 allows, each within its registers.
void plan_tiles(const std::vector<ChainItem> &chains, const std::vector<uint32_t> &events, const std::vector<int32_t> &cand_out, int p0, int k, uint32_t sF, uint32_t sR, int gw,
                int ring_cols, int n_slices, int target_groups, TilePlan &P) {
    P.rounds.clear(); P.bands.clear(); P.prog.clear();
    const int n = (int)chains.size();
    if (!n) return;
    const int cs = 16 * 64 * gw, ps = 4 * 64 * gw;
    P.prog.assign((size_t)(n + kTileWaves) * (kProgRegs * 64), 0u);
    const int n_bands = std::max(1, std::min((n + kTileWaves - 1) / kTileWaves, target_groups / std::max(1, n_slices)));
    for (int b = 0; b < n_bands; b++) {
        const int i0 = (int)((long long)n * b / n_bands), i1 = (int)((long long)n * (b + 1) / n_bands);
        if (i0 == i1) continue;
        TileBand band{(int32_t)P.rounds.size(), 0, p0 + chains[(size_t)i0].win, 0};
        // rounds of equal size rather than full ones and a short one: the band takes ceil(items / 16) rounds either way
        const int want = (i1 - i0 + kTileWaves - 1) / kTileWaves, per = (i1 - i0 + want - 1) / want;
        int hi_prev = -1;
        for (int a = i0; a < i1;) {
            int e = a + 1;
            while (e < i1 && e - a < per && chains[(size_t)e].win + k - chains[(size_t)a].win <= ring_cols) e++;
            const int lo = p0 + chains[(size_t)a].win, hi = p0 + chains[(size_t)e - 1].win + k;
            const int c0 = hi_prev < 0 ? lo : std::max(hi_prev, lo);
            P.rounds.push_back(TileRound{a, e - a, c0, std::max(c0, hi), (c0 - band.cbase) % ring_cols, 0});
            hi_prev = std::max(hi_prev, hi);
            a = e;
        }
        band.n_rounds = (int32_t)P.rounds.size() - band.round0;
        P.bands.push_back(band);
        for (int i = i0; i < i1; i++) {
            const ChainItem &ch = chains[(size_t)i];
            const int slot0 = (p0 + ch.win - band.cbase) % ring_cols;
            auto entry = [&](int j, int base) {
                return (uint32_t)(((slot0 + j) % ring_cols) * cs + base * ps) | (((sF >> j) & 1u) ? kStrictF : 0u) | (((sR >> j) & 1u) ? kStrictR : 0u);
            };
            auto sym = [&](int j) { return (ch.sym[j >> 3] >> (4 * (j & 7))) & 15u; };
            uint32_t *blk = P.prog.data() + (size_t)i * (kProgRegs * 64);
            int n1 = 0, n2 = 0, nq = 0;
            for (int j = 0; j < k; j++)
                if ((ch.pos1 >> j) & 1u) blk[32 + n1++] = entry(j, __builtin_ctz(sym(j)));
            for (int j = 0; j < k; j++)
                if ((ch.pos2 >> j) & 1u) {
                    const uint32_t sy = sym(j);
                    blk[64 + 2 * n2] = entry(j, __builtin_ctz(sy));
                    blk[64 + 2 * n2 + 1] = entry(j, __builtin_ctz(sy & (sy - 1u)));
                    n2++;
                }
            for (int j = 0; j < k; j++)
                if ((ch.pos4 >> j) & 1u) {
                    uint32_t sy = sym(j);
                    while (sy) {
                        const int base = __builtin_ctz(sy);
                        sy &= sy - 1u;
                        blk[128 + nq++] = entry(j, base) | (sy ? 0u : kLast);
                    }
                }
            for (int q = 0; q < ch.n_ev; q++) {
                const uint32_t ev = events[(size_t)ch.ev0 + (size_t)q];          // position | lost base (one-hot) << 8 | step << 16
                blk[256 + q] = entry((int)(ev & 255u), __builtin_ctz((ev >> 8) & 15u)) | ((ev >> 16) << 24);
            }
            for (int t = 0; t < 24; t++) blk[6 * 64 + t] = (uint32_t)cand_out[(size_t)ch.cand0 + (size_t)(t / 3)];
            const uint32_t head[7] = {(uint32_t)ch.win, (uint32_t)ch.cand0, (uint32_t)ch.n_steps, (uint32_t)ch.n_ev, (uint32_t)n1, (uint32_t)n2, (uint32_t)nq};
            for (int q = 0; q < 7; q++) blk[q] = head[q];
        }
    }
}

int launch_eval_tile(mp_ctx *c, int gw, unsigned long long *device_out) {
    static const TileFn fn4[4] = {eval_tile_kernel<1, 4>, eval_tile_kernel<2, 4>, eval_tile_kernel<3, 4>, eval_tile_kernel<4, 4>};
    static const TileFn fn2[4] = {eval_tile_kernel<1, 2>, eval_tile_kernel<2, 2>, eval_tile_kernel<3, 2>, eval_tile_kernel<4, 2>};
    const TileFn fn = (gw == 4 ? fn4 : fn2)[c->v];
    const int nw32 = c->n_pad / 32;
    const size_t lds = (size_t)c->tile_rc * 16 * 64 * gw + kTileWaves * 12 * 4;
    if (!c->tile_attr_set) {          // once per plan: the kernel may take (nearly) all of a CU's LDS
        HIPCK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        c->tile_attr_set = true;
    }
    TileArgs ta{reinterpret_cast<const uint32_t *>(c->cols), reinterpret_cast<const uint32_t *>(c->excl), nw32, c->tile_rc,
                c->tile_prog, device_out, c->tile_rounds, c->tile_bands, c->tile_n_slices, nullptr};
    const unsigned grid = (unsigned)(c->tile_n_bands * c->tile_n_slices);
    const char *prof_path = getenv("MP_TILE_PROF");          // debugging: per-wave phase stamps of ONE launch written to that file
    const size_t n_prof = (size_t)grid * kTileWaves * kProfSlots;
    if (prof_path) {
        HIPCK(c, hipMalloc((void **)&ta.prof, n_prof * 8));
        HIPCK(c, hipMemsetAsync(ta.prof, 0, n_prof * 8, c->stream));
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kTileThreads), lds, c->stream, ta);
    HIPCK(c, hipGetLastError());
    if (prof_path) {
        std::vector<unsigned long long> h(n_prof);
        HIPCK(c, hipStreamSynchronize(c->stream));
        HIPCK(c, hipMemcpy(h.data(), ta.prof, n_prof * 8, hipMemcpyDeviceToHost));
        (void)hipFree(ta.prof);
        if (FILE *f = fopen(prof_path, "wb")) {
            const int hdr[4] = {(int)grid, kTileWaves, kProfSlots, c->tile_n_slices};
            fwrite(hdr, sizeof hdr, 1, f);
            fwrite(h.data(), 8, n_prof, f);
            fclose(f);
        }
    }
    return MP_OK;
}

}  // namespace mp
