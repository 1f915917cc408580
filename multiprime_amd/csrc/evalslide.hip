// evalslide.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of include/mprime.h.
// Candidate x sequence coverage evaluation of nested refinement chains (mis_primer_check + Y_distance, V20:1103-1130, 229-233) by
// SLIDING along the windows: eval_slide_kernel.  The arithmetic and the plan are in slidecore.hpp / slideplan.hpp (shared with the
// CPU emulation tools/slide_emul.cpp); this file is the GPU environment of the band routine and its launch.
//
// Why: the first-pass kernels (eval.hip, evalprog.hip) fetch the k (or more) column planes of a window's most degenerate member for
// EVERY window, so every plane word leaves L2 ~18 times — they are bound by the rate at which a CU's vector memory path returns
// data to registers (64 B/clk per CU).  Here a wave keeps the bit-sliced mismatch count of the per-column
// reference k-mer and moves it from window to window with ONE plane fetch (plus the chain's event planes, fetched once per item and
// used twice: as corrections of the count and as the steps of the walk): ~9 fetches per window and 32-row word instead of ~33.
//
// Work decomposition: workgroup = 4 waves = 4 x 64 lanes x GW row words, one BAND of consecutive windows (k - 1 warm-up columns,
// then one column per window).  grid = bands x row slices, slice = blockIdx % slices: workgroup b runs on XCD b % 8, so an XCD owns
// row slices and walks the bands in order — consecutive bands find their columns in that XCD's L2, HBM sees every plane once.
// LDS: per wave a ring of the last k reference-mismatch words of its rows (the column sliding out shares the slot of the one
// sliding in; strict positions read theirs), per workgroup the 24 counters of every item of the band: waves add their totals there
// (ds_add), one flush of global atomics per workgroup and band.
#include "common.hpp"
#include "bitslice.hpp"
#include "slidecore.hpp"
#include "evalslide.hpp"
#include "chainbody.hpp"

using namespace mp;

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct SlideKernArgs {
    SlideArgs A;
    const uint32_t *cols32;            // [n_cols][4][nw32] one-hot column planes
    const uint32_t *excl32;            // [W][nw32]
    int nw32;
    unsigned long long *out;
    int wc, wc_pad;                    // row slices of 256 x GW words; padded to a multiple of 8 (a slice stays on one XCD)
    int max_items;                     // items of the largest band (LDS table rows)
    int n_slide_blocks;                // the grid's first workgroups slide; the rest run the step's patch units (chainbody.hpp)
    EvalChainArgs chain;
    unsigned long long *clear;         // mp_eval_launch_rotating: the next launch's counter block (chainbody.hpp: clear_counters)
    uint32_t n_clear;
    int patch_words;                   // words per lane of the patch units in the grid's tail: 1 (eval_patch_wave) or 8 (eval_chain_block<LV, 8, 4>)
    unsigned long long *stamps;        // null; MP_EXPERIMENT_STAMPS: [workgroup][8] wall-clock stamps of its phases (tools/slide_stamps.py)
};

template <int GW>
struct DevEnv {
    const SlideKernArgs &K;
    __amdgpu_buffer_rsrc_t rs_cols, rs_excl;
    int voff, lane;
    uint32_t row_bytes;
    bool live;
    uint32_t *ring;                    // this lane's GW words of slot 0; slots are 64 * GW words apart
    uint32_t *tab;                     // the workgroup's counters [item of the band][12]
    uint32_t itv;                      // 64 iteration words, one per lane
    uint32_t live_mask;

    __device__ __forceinline__ DevEnv(const SlideKernArgs &k) : K(k) {}
    // phase stamps of the experiment build-in (one lane per workgroup; a null pointer in every other run: one scalar compare)
    // issue priority by progress through the band (0 = a quarter or more still to do ... ): the SIMD issues its OLDEST wave first, so the
    // first workgroup of a CU runs ahead and its last one finishes alone, with nobody to fill the stalls of its own loads (tools/slide_stamps.py:
    // 84 / 103 / 126 / 148 us).  A wave that is ahead steps down, the ones behind catch up, and a CU's workgroups end together.
    __device__ __forceinline__ void progress(int quarter) const {
#ifndef SLIDE_NO_PRIO
        if (quarter <= 0) __builtin_amdgcn_s_setprio(3);
        else if (quarter == 1) __builtin_amdgcn_s_setprio(2);
        else if (quarter == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
#endif
    }
    __device__ __forceinline__ void stamp(int i) const {
        if (K.stamps && threadIdx.x == 0) K.stamps[(size_t)blockIdx.x * 8 + i] = wall_clock64();
    }
    __device__ __forceinline__ SlideBand uband(int b) const {
        const SlideBand *p = K.A.bands + b;
        SlideBand r;
        r.w0 = __builtin_amdgcn_readfirstlane(p->w0); r.n_win = __builtin_amdgcn_readfirstlane(p->n_win);
        r.item0 = __builtin_amdgcn_readfirstlane(p->item0); r.n_items = __builtin_amdgcn_readfirstlane(p->n_items);
        r.iter0 = __builtin_amdgcn_readfirstlane(p->iter0); r.pad = 0;
        return r;
    }
    // iteration words arrive 64 at a time, one per lane (the array is padded by 64 words)
    __device__ __forceinline__ void load_iters(int idx) { itv = K.A.iters[(size_t)idx + (size_t)lane]; }
    __device__ __forceinline__ uint32_t iter_word(int j) const { return (uint32_t)__builtin_amdgcn_readlane((int)itv, j); }
    // an item's record: the words the band routine reads arrive through SCALAR loads (the address is wave-uniform; the records are
    // read-only: constant address space) — round 4 loaded the 32 words one per lane and took each out with a v_readlane: 13 vector
    // instructions per item on a kernel that is bound by vector-instruction issue
    typedef const uint32_t __attribute__((address_space(4))) * ConstWords;
    struct Rec { uint32_t w[8], sm_lo, sm_hi, flags; ConstWords p; };
    __device__ __forceinline__ Rec load_rec(int item) const {
        Rec r;
        r.p = (ConstWords)(K.A.recs + (size_t)__builtin_amdgcn_readfirstlane(item) * kSlideRec);
#pragma unroll
        for (int q = 0; q < 8; q++) r.w[q] = r.p[q];
        r.sm_lo = r.p[24]; r.sm_hi = r.p[25]; r.flags = r.p[28];
        return r;
    }
    __device__ __forceinline__ uint32_t rec_word(const Rec &r, int q) const {          // q is a constant after inlining
        return q < 8 ? r.w[q] : (q == 24 ? r.sm_lo : (q == 25 ? r.sm_hi : (q == 28 ? r.flags : r.p[q])));
    }
    __device__ __forceinline__ uint32_t rec_word_dyn(const Rec &r, int q) const { return r.p[q]; }
    // AUX: the load's cache policy bits (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
    template <int AUX = 0>
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int soff, uint32_t (&d)[GW]) const {
        if constexpr (GW == 4) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        } else if constexpr (GW == 2) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, AUX);
            d[0] = v.x; d[1] = v.y;
        } else {
            d[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, AUX);
        }
    }
    // SLIDE_EXP_ONE_ROW (experiment builds, tools/build_variant.sh; the results are WRONG, timing only): 1 = every plane fetch reads plane row 0
    // (cache hits only), 2 = the items' event planes only, 3 = the reference columns only
#ifndef SLIDE_EXP_ONE_ROW
#define SLIDE_EXP_ONE_ROW 0
#endif
    // [r6] the reference columns stream through (every word is used once, two iterations after its request): non-temporal, so that they do not push
    // the items' event planes — wanted again by a window up to k - 1 columns on — out of the XCD's L2.  Measured (profiles/r06_slide_limits.txt):
    // columns nt 0.1368 ms against 0.1397; event planes nt 0.1438 (they do live on L2 hits); sc1 on the columns: nothing.
#ifndef SLIDE_COL_AUX
#define SLIDE_COL_AUX 2
#endif
#ifndef SLIDE_EVT_AUX
#define SLIDE_EVT_AUX 0
#endif
    __device__ __forceinline__ void fetch(uint32_t row_off, uint32_t (&d)[GW]) const {
        load<SLIDE_COL_AUX>(rs_cols, SLIDE_EXP_ONE_ROW == 1 || SLIDE_EXP_ONE_ROW == 3 ? 0 : (int)row_off, d);
    }
    __device__ __forceinline__ void fetch_event(uint32_t row_off, uint32_t (&d)[GW]) const {
        load<SLIDE_EVT_AUX>(rs_cols, SLIDE_EXP_ONE_ROW == 1 || SLIDE_EXP_ONE_ROW == 2 ? 0 : (int)row_off, d);
    }
    __device__ __forceinline__ void valid_of(uint32_t row_off, uint32_t (&v)[GW]) const {
        load(rs_excl, (int)row_off, v);
#pragma unroll
        for (int i = 0; i < GW; i++) v[i] = bop<kSlAndNot>(live_mask, v[i], 0u);
    }
    // the lane's GW words of a ring slot move as one 4 / 8 / 16-byte LDS access
    __device__ __forceinline__ void ring_put(uint32_t *p, const uint32_t (&v)[GW]) const {
        if constexpr (GW == 4) *reinterpret_cast<u32x4 *>(p) = u32x4{v[0], v[1], v[2], v[3]};
        else if constexpr (GW == 2) *reinterpret_cast<u32x2 *>(p) = u32x2{v[0], v[1]};
        else p[0] = v[0];
    }
    __device__ __forceinline__ void ring_get(const uint32_t *p, uint32_t (&v)[GW]) const {
        if constexpr (GW == 4) { const u32x4 t = *reinterpret_cast<const u32x4 *>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else if constexpr (GW == 2) { const u32x2 t = *reinterpret_cast<const u32x2 *>(p); v[0] = t.x; v[1] = t.y; }
        else v[0] = p[0];
    }
    __device__ __forceinline__ void ring_zero(int k) {
        uint32_t z[GW];
#pragma unroll
        for (int i = 0; i < GW; i++) z[i] = 0u;
        for (int s = 0; s < k; s++) ring_put(ring + s * (64 * GW), z);
    }
    __device__ __forceinline__ void ring_write(int slot, const uint32_t (&in)[GW]) { ring_put(ring + slot * (64 * GW), in); }
    __device__ __forceinline__ void ring_read(int slot, uint32_t (&o)[GW]) const { ring_get(ring + slot * (64 * GW), o); }
    // The wave's OUT counts of the item's 8 member slots into the workgroup's table: 12 words per item, two 16-bit counts each
    // (registers 0-7: out1 | outF << 16 of slot q; 8-11: outR of slots 2 (q - 8), 2 (q - 8) + 1) — a workgroup covers 4 x 64 x 32 GW
    // <= 32768 rows, so a field never carries.
    // A TRANSPOSING reduction: a step that adds partner lanes also halves the registers — of two registers, half of the lanes go on
    // with the first and the other half with the second.  DPP masks select whole quads (bank_mask) and rows, so the steps ACROSS the
    // quads of a row come first: row_ror:4 under bank masks 0x5 / 0xa takes 12 registers to 6 (even quads: the first register's quads
    // summed in pairs, odd quads: the second's), row_ror:8 under 0x3 / 0xc takes 6 to 3 — quad q of register j then holds, lane by
    // lane, the row's partial sums of register 4 j + q.  The two steps INSIDE a quad follow on 3 registers instead of 12, and
    // v_permlane16_swap / v_permlane32_swap (gfx950) fold the four rows the same transposing way: 3 registers -> 1, whose row j (j < 3)
    // holds the WAVE's total of register 4 j + q in quad q.  12 lanes hand them to the table (ds_add, one address per lane: lanes of
    // one wave never collide — eight lanes adding to ONE address took twice the whole kernel's time).  31 instructions, no LDS
    // parking, no wave barrier: round 4 reduced all 12 registers over the row (48 DPP adds), parked the rows' sums in LDS and had 12
    // lanes add them up (66 instructions and two LDS round trips per item).
    // The 24 DPP adds are written out: `v_add_u32_dpp dst, a, a ... bank_mask` leaves the quads outside the bank mask as they are, which
    // is what lets one register take half of its quads' sums from itself and the other half from its partner register in two
    // instructions (through the builtin — a v_mov_dpp whose result then has to be added — a pair costs three to four).  The order keeps
    // every DPP read at least two instructions behind the write of its register (the hazard the compiler would otherwise pad with
    // s_nop; it does not look inside an asm block, hence the s_nop in front: the inputs may come straight out of a VALU instruction).
    __device__ __forceinline__ void commit(int idx, const uint32_t (&accPF)[8], const uint32_t (&accR)[4]) {
        uint32_t x0 = accPF[0], x1 = accPF[1], x2 = accPF[2], x3 = accPF[3], x4 = accPF[4], x5 = accPF[5], x6 = accPF[6], x7 = accPF[7];
        uint32_t x8 = accR[0], x9 = accR[1], x10 = accR[2], x11 = accR[3];
        asm volatile(
            "s_nop 1\n\t"
            // across quads, 12 -> 6 registers: the even quads of x[2q] go on with x[2q], the odd quads with x[2q + 1]
            "v_add_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_u32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_u32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_u32_dpp %6, %6, %6 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_u32_dpp %8, %8, %8 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_u32_dpp %10, %10, %10 row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_u32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_u32_dpp %2, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_u32_dpp %4, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_u32_dpp %6, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_u32_dpp %8, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_u32_dpp %10, %11, %11 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
            // 6 -> 3 registers: quads 0, 1 of x[4j] go on with x[4j] (quad 0: all of x[4j], quad 1: all of x[4j + 1]), quads 2, 3 with x[4j + 2]
            "v_add_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_u32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_u32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_u32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_u32_dpp %4, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_u32_dpp %8, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            // inside the quads: every lane of quad q = the row's sum of register 4j + q
            "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_u32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_u32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_u32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_u32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1"
            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8), "+v"(x9), "+v"(x10), "+v"(x11));
        // rows: (z0, z1) -> rows [z0: 0+1, z1: 0+1, z0: 2+3, z1: 2+3]; z2 with itself -> [0+1, 0+1, 2+3, 2+3]; halves -> [z0, z1, z2, z2]
        typedef unsigned int u32pair __attribute__((ext_vector_type(2)));
        const u32pair s01 = __builtin_amdgcn_permlane16_swap(x0, x4, false, false);
        const uint32_t a = s01.x + s01.y;
        const u32pair s22 = __builtin_amdgcn_permlane16_swap(x8, x8, false, false);
        const uint32_t b = s22.x + s22.y;
        const u32pair h = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        const uint32_t tot = h.x + h.y;
        if (lane < 48 && (lane & 3) == 0) atomicAdd(&tab[idx * 12 + (lane >> 4) * 4 + ((lane >> 2) & 3)], tot);
    }
};

#ifndef SLIDE_MIN_WAVES
#define SLIDE_MIN_WAVES 1
#endif
template <int LV, int GW, bool FAST>
__global__ __launch_bounds__(kBlock, GW == 4 ? 1 : SLIDE_MIN_WAVES) void eval_slide_kernel(const SlideKernArgs K) {
    extern __shared__ __align__(16) uint32_t lds[];
    clear_counters(K.clear, K.n_clear, blockIdx.x, gridDim.x);
    if (K.stamps && threadIdx.x == 0) K.stamps[(size_t)blockIdx.x * 8] = wall_clock64();
    if ((int)blockIdx.x >= K.n_slide_blocks) {
        // the tail of the grid: the patch-list rows of the same step (their real k-mers added, their plain slices taken back) — no
        // launch of their own, they fill the slots the sliding workgroups leave as they finish
        // (a wave per unit, one word per lane: chainbody.hpp eval_patch_wave; the first patch.n_blocks workgroups add the patch rows' real
        // k-mers, the following patch.neg_blocks take their plain slices back)
        // ... when a window's patch rows fit one such unit (up to 2048 of them: the shallow alignments, where these units are the launch's
        // tail); deeper ones keep the units of 8 words per lane — a third of the waves, and the tail hides behind the sliding workgroups
        // that finish last (10^6 rows: 0.165 ms per step with them, 0.176 with three one-word units per window)
        const unsigned pb = blockIdx.x - (unsigned)K.n_slide_blocks, wv = threadIdx.x >> 6;
#ifndef SLIDE_PATCH_PRIO
#define SLIDE_PATCH_PRIO 3
#endif
        __builtin_amdgcn_s_setprio(SLIDE_PATCH_PRIO);                  // short and late: as the youngest waves of a busy CU they would wait for everybody
        if (K.patch_words == 1) {
            const bool negative = (int)pb >= K.chain.patch.n_blocks;
            const unsigned unit = (pb - (negative ? (unsigned)K.chain.patch.n_blocks : 0u)) * (kBlock / 64) + wv;
            eval_patch_wave<LV>(K.chain, lds + wv * 12, lds + 64 + wv * (64 * kPatchEvents), unit, negative);
        } else {
            uint32_t(&s_part)[kBlock / 64][12] = *reinterpret_cast<uint32_t(*)[kBlock / 64][12]>(lds);
            eval_chain_block<LV, 8, 4>(K.chain, s_part, pb);
        }
        if (K.stamps && threadIdx.x == 0) K.stamps[(size_t)blockIdx.x * 8 + 7] = wall_clock64();
        return;
    }
    const int slice = (int)(blockIdx.x % (unsigned)K.wc_pad), band = (int)(blockIdx.x / (unsigned)K.wc_pad);
    if (slice >= K.wc) return;
    const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int ring_words = K.A.k * 64 * GW;
    uint32_t *tab = lds + (kBlock / 64) * ring_words;
    DevEnv<GW> env(K);
    env.lane = lane;
    env.ring = lds + wv * ring_words + lane * GW;
    env.tab = tab;
    env.itv = 0u;
    const int word0 = (slice * kBlock + (int)threadIdx.x) * GW;
    env.live = word0 < K.nw32;                         // nw32 is a multiple of 8 >= GW: a lane's words are inside the row or all past it
    env.voff = env.live ? word0 * 4 : 0x7FFFFF00;           // past num_records: a lane without rows reads zeros (all gaps: never counted)
    env.live_mask = env.live ? 0xFFFFFFFFu : 0u;
    env.row_bytes = (uint32_t)K.nw32 * 4u;
    env.rs_cols = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(K.cols32), 0, 0x7FFFFF00, 0x00020000);
    env.rs_excl = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(K.excl32), 0, 0x7FFFFF00, 0x00020000);
    const SlideBand bd = env.uband(band);
    for (int i = (int)threadIdx.x; i < bd.n_items * 12; i += kBlock) tab[i] = 0u;
    __syncthreads();
    env.stamp(1);
    const bool wave_live = (slice * kBlock + wv * 64) * GW < K.nw32;          // some lane of the wave holds rows
    if (wave_live) slide_band<LV, GW, true, false, FAST>(env, K.A, band);
    env.stamp(4);
    __syncthreads();
    env.stamp(5);
    // One flush per workgroup and band.  The table holds OUT rows; member t of an item reports its slot's counts turned round:
    // perfect = rows - out1, forward (1..v mismatches, none at a strict position) = out1 - outF, reverse = out1 - outR.
    int live_waves = 0;
    for (int w = 0; w < kBlock / 64; w++) live_waves += (slice * kBlock + w * 64) * GW < K.nw32 ? 1 : 0;
    const uint32_t rows = (uint32_t)live_waves * 64u * 32u * GW;
    // A thread's entries are kBlock apart; what it needs of their records (the candidate a member slot reports to, the slot map) is
    // requested for four entries at once — one memory round trip per four instead of one per entry: this flush is the last thing the
    // kernel's last workgroups do, nothing hides it (6.5 us per workgroup at 60 items a band before, tools/slide_stamps.py).
    const int n_entries = bd.n_items * 24;
    for (int i0 = (int)threadIdx.x; i0 < n_entries; i0 += 4 * kBlock) {
        int oc[4];
        uint32_t map[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * kBlock, ic = i < n_entries ? i : (int)threadIdx.x;      // (past the end: the thread's first entry again, unused)
            const uint32_t *rec = K.A.recs + (size_t)(bd.item0 + ic / 24) * kSlideRec;
            oc[u] = (int)rec[8 + (ic % 24) / 3];
            map[u] = rec[26];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * kBlock;
            if (i >= n_entries || oc[u] < 0) continue;
            const int item = i / 24, r = i % 24, t = r / 3, kind = r % 3;              // 0 perfect, 1 forward, 2 reverse
            const int slot = (int)((map[u] >> (4 * t)) & 15u);
            const uint32_t *row = tab + item * 12;
            const uint32_t out1 = row[slot] & 0xFFFFu, outF = row[slot] >> 16, outR = (row[8 + (slot >> 1)] >> (16 * (slot & 1))) & 0xFFFFu;
            const uint32_t val = kind == 0 ? rows - out1 : out1 - (kind == 1 ? outF : outR);
            if (val) atomicAdd(&K.out[(size_t)oc[u] * 3 + kind], (unsigned long long)val);
        }
    }
    env.stamp(6);
}

typedef void (*SlideFn)(const SlideKernArgs);

}  // namespace

namespace mp {

void free_slide(mp_ctx *c) {
    dev_free(c, &c->slide_bands, (size_t)c->slide_n_bands);
    dev_free(c, &c->slide_iters, c->slide_n_iters);
    dev_free(c, &c->slide_recs, (size_t)c->slide_items * kSlideRec);
    dev_free(c, &c->chain_rest, (size_t)c->n_rest);
    dev_free(c, &c->chain_slid, (size_t)c->slide_items);
    c->slide_items = c->slide_n_bands = c->slide_max_items = 0;
    c->slide_n_iters = 0;
    c->n_rest = c->rest_max_steps = 0;
}

// MP_EVAL_SLIDE=0 keeps the first-pass kernels; =1 forces the sliding kernel at any size; default: above 262144 rows (at and below, the
// bands that fill the chip are so short that their k - 1 warm-up columns outweigh what sliding saves, and the column planes still
// come out of L2 / the Infinity Cache for the first-pass kernel: equal at 262144, 18-24 % faster from 327680 up —
// profiles/r04_slide_sizes.txt).
// MP_SLIDE_GW (1, 2, 4): row words per lane; MP_SLIDE_BAND: windows per band.
int upload_eval_slide(mp_ctx *c, const std::vector<ChainItem> &chains, const std::vector<uint32_t> &events, const std::vector<int32_t> &cand_out) {
    free_slide(c);
    const char *se = getenv("MP_EVAL_SLIDE");
    // (k > v: the column pass runs without exclusion words and relies on an all-gap slice — a padding row — having more than v mismatches)
    if (chains.empty() || c->v > 3 || c->k <= c->v || (se ? atoi(se) != 1 : c->n_pad <= 262144)) return MP_OK;
    const int nw32 = c->n_pad / 32, n_cols = c->n_chunks * 32;
    if (((unsigned long long)n_cols * 4ull + 1ull) * (unsigned long long)nw32 * 4ull >= 0x7FFFFFFFull) return MP_OK;     // plane rows are addressed by a 32-bit scalar offset
    if ((unsigned long long)c->n_win * (unsigned long long)nw32 * 4ull >= 0x7FFFFFFFull) return MP_OK;
    int gw = 2;
    if (const char *e = getenv("MP_SLIDE_GW")) { const int g = atoi(e); if (g == 1 || g == 2 || g == 4) gw = g; }
    const int wc = (nw32 + kBlock * gw - 1) / (kBlock * gw), wc_pad = wc >= 8 ? (wc + 7) / 8 * 8 : wc;
    // Bands: ONE round of workgroups where that leaves bands of at most 96 windows — 256 CUs x 4 resident workgroups (VGPRs and the
    // rings' LDS allow four) = 1024 at a time, so 1024 / slices bands; every workgroup does the same work, a second round would
    // only add warm-up columns (0.184 ms with 16 bands of 60 windows against 0.190 with 32 of 30 at 1 048 576 rows).  More rounds
    // when a band would be longer, never bands shorter than 8 windows.
    // [r6] ... and "resident" is what the rings' LDS allows: k x 64 x GW words per wave.  k = 18 at two words per lane is the last k with FOUR
    // workgroups per CU (39.7 KB each); from k = 19 on three fit (two from k = 26), and bands sized for 1024 workgroups at a time then ran in a
    // round and a third — k = 20 took 1.42 x the time of k = 18 (bench_detail k_sweep, 0.185 ms).  The band length follows the residency the
    // plan's own LDS need allows; a plan is rebuilt when its longest band's table does not fit the residency it was sized for.
    const int span = chains.back().win - chains.front().win + 1;
    std::vector<SlideChainIn> in(chains.size());
    for (size_t i = 0; i < chains.size(); i++) {
        const ChainItem &ch = chains[i];
        in[i] = SlideChainIn{ch.win, ch.cand0, ch.n_steps, ch.ev0, ch.n_ev, {ch.sym[0], ch.sym[1], ch.sym[2], ch.sym[3]}};
    }
    const size_t ring_bytes = (size_t)(kBlock / 64) * ((size_t)c->k * 64 * gw) * sizeof(uint32_t), lds_cu = 160 * 1024;
    int resident = (int)std::min<size_t>(4, std::max<size_t>(1, lds_cu / (ring_bytes + 1024)));
    if (gw == 4) resident = 1;                                           // (launch bounds of the four-word form)
    SlidePlan P;
    for (;; resident--) {
        const int per_round = std::max(1, 256 * resident / std::max(1, wc_pad));
        int band = (span + per_round - 1) / per_round;
        for (int r = 2; band > 96; r++) band = (span + per_round * r - 1) / (per_round * r);
        band = std::max(8, band);
        if (const char *e = getenv("MP_SLIDE_BAND")) band = std::max(1, atoi(e));
        if (!build_slide_plan(in, events, cand_out, c->k, (uint32_t)c->sF, (uint32_t)c->sR, c->p0, n_cols, band, (uint32_t)nw32 * 4u, true, P)) return MP_OK;
        const size_t need = ring_bytes + (size_t)P.max_items_band * 12 * sizeof(uint32_t);
        if (resident <= 1 || need * (size_t)resident <= lds_cu) break;
    }
    P.iters.resize(P.iters.size() + 64, 0u);                      // uiter reads 64 words at a time
    int rc;
    if ((rc = dev_alloc(c, &c->slide_bands, P.bands.size()))) return rc;
    c->slide_n_bands = (int)P.bands.size();
    if ((rc = dev_alloc(c, &c->slide_iters, P.iters.size()))) return rc;
    c->slide_n_iters = P.iters.size();
    c->slide_items = (int)P.item_of.size();
    if ((rc = dev_alloc(c, &c->slide_recs, P.recs.size()))) return rc;
    HIPCK(c, hipMemcpy(c->slide_bands, P.bands.data(), sizeof(SlideBand) * P.bands.size(), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(c->slide_iters, P.iters.data(), sizeof(uint32_t) * P.iters.size(), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(c->slide_recs, P.recs.data(), sizeof(uint32_t) * P.recs.size(), hipMemcpyHostToDevice));
    c->slide_max_items = P.max_items_band;
    c->slide_ns = P.ns; c->slide_spos = P.spos; c->slide_fmask = P.fmask; c->slide_rmask = P.rmask;
    // [r6] at most three strict positions per side (the reference's default `-c 1,2,-1`, V20:85): the strict positions as a two-bit count per
    // side (slidecore.hpp FAST); MP_SLIDE_STRICT=0 keeps the per-position form
    c->slide_fast = slide_strict_lists(c->k, (uint32_t)c->sF, (uint32_t)c->sR, c->slide_fpos, c->slide_rpos);
    if (const char *e = getenv("MP_SLIDE_STRICT")) { if (atoi(e) == 0) c->slide_fast = false; }
    c->slide_gw = gw;
    {
        std::vector<ChainItem> slid;
        for (int32_t i : P.item_of) slid.push_back(chains[(size_t)i]);
        if ((rc = dev_alloc(c, &c->chain_slid, slid.size()))) return rc;
        HIPCK(c, hipMemcpy(c->chain_slid, slid.data(), sizeof(ChainItem) * slid.size(), hipMemcpyHostToDevice));
    }
    if (!P.rest.empty()) {
        std::vector<ChainItem> rest;
        for (int32_t i : P.rest) { rest.push_back(chains[(size_t)i]); c->rest_max_steps = std::max(c->rest_max_steps, (int)chains[(size_t)i].n_steps); }
        if ((rc = dev_alloc(c, &c->chain_rest, rest.size()))) return rc;
        c->n_rest = (int)rest.size();
        HIPCK(c, hipMemcpy(c->chain_rest, rest.data(), sizeof(ChainItem) * rest.size(), hipMemcpyHostToDevice));
    }
    return MP_OK;
}

int launch_eval_slide(mp_ctx *c, unsigned long long *device_out, const EvalChainArgs *patch, int patch_blocks, int patch_words,
                      unsigned long long *clear, uint32_t n_clear) {
#define SLIDE_ROW(LV, F) {eval_slide_kernel<LV, 1, F>, eval_slide_kernel<LV, 2, F>, eval_slide_kernel<LV, 4, F>}
    static const SlideFn fn[2][4][3] = {{SLIDE_ROW(1, false), SLIDE_ROW(2, false), SLIDE_ROW(3, false), SLIDE_ROW(4, false)},
                                        {SLIDE_ROW(1, true), SLIDE_ROW(2, true), SLIDE_ROW(3, true), SLIDE_ROW(4, true)}};
#undef SLIDE_ROW
    const int gw = c->slide_gw, gi = gw == 4 ? 2 : gw - 1;
    const int nw32 = c->n_pad / 32;
    SlideKernArgs K;
    K.A = SlideArgs{c->slide_bands, c->slide_iters, c->slide_recs, c->k, c->p0, c->slide_ns, c->slide_spos, c->slide_fmask, c->slide_rmask,
                    (uint32_t)nw32 * 4u, c->slide_fpos, c->slide_rpos};
    K.cols32 = reinterpret_cast<const uint32_t *>(c->cols);
    K.excl32 = reinterpret_cast<const uint32_t *>(c->excl);
    K.nw32 = nw32;
    K.out = device_out;
    K.wc = (nw32 + kBlock * gw - 1) / (kBlock * gw);
    K.wc_pad = K.wc >= 8 ? (K.wc + 7) / 8 * 8 : K.wc;
    K.max_items = c->slide_max_items;
    K.n_slide_blocks = c->slide_n_bands * K.wc_pad;
    if (patch) K.chain = *patch;
    else { memset(&K.chain, 0, sizeof K.chain); patch_blocks = 0; }
    K.chain.clear = nullptr; K.chain.n_clear = 0;
    K.clear = clear; K.n_clear = n_clear;
    K.patch_words = patch_words;
    K.stamps = nullptr;
    const char *stamp_file = getenv("MP_EXPERIMENT_STAMPS");           // tools/slide_stamps.py: the launch is synchronous then
    const size_t n_stamp = ((size_t)K.n_slide_blocks + (size_t)patch_blocks) * 8;
    if (stamp_file) {
        HIPCK(c, hipMalloc(&K.stamps, n_stamp * sizeof(unsigned long long)));
        HIPCK(c, hipMemsetAsync(K.stamps, 0, n_stamp * sizeof(unsigned long long), c->stream));
    }
    size_t lds = ((size_t)(kBlock / 64) * ((size_t)c->k * 64 * gw) + (size_t)c->slide_max_items * 12) * sizeof(uint32_t);
    if (patch_blocks) lds = std::max(lds, (size_t)(64 + (kBlock / 64) * 64 * kPatchEvents) * sizeof(uint32_t));      // the patch units' rows and stashes
    if (lds > 160 * 1024) return fail(c, MP_ERR_ARG, "sliding evaluation: a band needs %zu bytes of LDS", lds);
    SlideFn f = fn[c->slide_fast ? 1 : 0][c->v][gi];
    if (lds > 48 * 1024) HIPCK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(f, dim3((unsigned)K.n_slide_blocks + (unsigned)patch_blocks), dim3(kBlock), lds, c->stream, K);
    HIPCK(c, hipGetLastError());
    if (stamp_file) {
        std::vector<unsigned long long> h(n_stamp + 2);
        HIPCK(c, hipStreamSynchronize(c->stream));
        HIPCK(c, hipMemcpy(h.data() + 2, K.stamps, n_stamp * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIPCK(c, hipFree(K.stamps));
        h[0] = (unsigned long long)K.n_slide_blocks; h[1] = (unsigned long long)patch_blocks;
        if (FILE *fo = fopen(stamp_file, "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), fo); fclose(fo); }
    }
    return MP_OK;
}

}  // namespace mp
