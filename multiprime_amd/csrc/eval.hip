// eval.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of
// include/mprime.h.  Candidate x sequence coverage evaluation (mp_eval_*) and per-window statistics (mp_window_stats):
//   (4)  row-per-lane evaluation on the window words — eval_kernel (v > 3, MP_EVAL_MODE=rows); patch planes for the rest
//   (4b) bit-sliced evaluation on the one-hot column planes, any 8 candidates — eval_bits_kernel
//   (4c) bit-sliced evaluation of nested refinement chains — eval_chain_kernel (the benchmarked kernel)
//   (4d) state_matrix / trans_matrix counts — window_stats_kernel
//   per-sequence coverage masks — mask_rows_kernel
#include "common.hpp"
#include "winwords.hpp"
#include "bitslice.hpp"
#include "evalprog.hpp"
#include "evalslide.hpp"
#include "chainbody.hpp"
#include "evalx.hpp"

using namespace mp;

namespace {

// ----------------------------------------------------------------------------------------------
// (4) candidate x sequence evaluation (V20:1103-1130, Y_distance V20:229-233)
// ----------------------------------------------------------------------------------------------
// A candidate is held as four k-bit words nX = positions whose symbol does NOT contain base X.
// For a sequence k-mer (b0,b1,g) the mismatch word is  g | select(nA,nC,nG,nT by (b1,b0))  — three
// v_bfi_b32 and one v_or_b32.  With D = set bits of mm:
//   perfect = (mm == 0)
//   F_raw   = |D| <= v  and  mm & strictF == 0      (R_raw likewise)
// F_raw includes the perfect rows, so the kernel counts F_raw and subtracts `perfect` once per
// block (F_mis = F_raw - perfect), which saves the |D| != 0 test per evaluation.
// VMODE 1 (v == 1, the pipeline default): |D| <= 1  <=>  mm & (mm-1) == 0, so
//   F_raw <=> mm & ((mm-1) | strictF) == 0  — one v_add, one v_bitop3, one compare, no popcount.
template <typename W> __device__ inline W bfi(W s, W a, W b) { return (s & a) | (~s & b); }

// W = the window word type (winwords.hpp): uint32_t for k <= 31, uint64_t for primers of 32..63 bases, which only this row-per-lane
// path evaluates (the bit-sliced kernels keep 32 positions per item)
template <typename W> struct alignas(4 * sizeof(W)) CandN { W x, y, z, w; };      // nA,nC,nG,nT of one candidate (uint4 for W = uint32_t)
template <typename W>
struct EvalArgsT {
    MsaArgs M;                  // window words are derived from the planes on the fly (winwords.hpp)
    int p0;
    int n_pad, k;
    const EvalItem *items;
    const CandN<W> *cand_n;     // [padded cand] nA,nC,nG,nT
    const int32_t *cand_out;    // [padded cand] index into out or -1
    const int32_t *extra_off;   // [W+1] or nullptr
    const W *extra_words;
    W sF, sR;
    int v;
    W kmask;
    int rows_per_split;
    unsigned long long *out;
};
typedef EvalArgsT<uint32_t> EvalArgs;

// COUNT 0: per-lane VGPR accumulators (v_cmp + v_addc); COUNT 1: wave ballots counted on the
// scalar unit (v_cmp -> s_bcnt1_i32_b64 -> s_add), accumulators live in SGPRs.
template <int CC, int COUNT>
struct EvalAcc {
    uint32_t p[CC], f[CC], r[CC];
    __device__ inline void clear() {
#pragma unroll
        for (int c = 0; c < CC; c++) p[c] = f[c] = r[c] = 0;
    }
    __device__ inline void add(int c, bool pp, bool ff, bool rr) {
        if (COUNT == 0) {
            p[c] += pp; f[c] += ff; r[c] += rr;
        } else {
            p[c] += (uint32_t)__popcll(__ballot(pp));
            f[c] += (uint32_t)__popcll(__ballot(ff));
            r[c] += (uint32_t)__popcll(__ballot(rr));
        }
    }
};

// FORM 0: bfi select, operand placement left to the compiler (candidate words end up in SGPRs and
//         the one-SGPR-per-VALU constant-bus rule splits every v_bfi_b32 in two);
// FORM 1: candidate words pinned in VGPRs: three v_bfi_b32 + one v_or_b32 per evaluation;
// FORM 2: per-row one-hot words eqX (4 ops per row, shared by the candidates) and a chain of four
//         v_and_or_b32 with the candidate words as the single SGPR operand.
template <int CC, int VMODE, int COUNT, int FORM, typename W>
__device__ inline void eval_row(W b0, W b1, W g, const EvalArgsT<W> &A,
                                const W (&nA)[CC], const W (&nC)[CC], const W (&nG)[CC],
                                const W (&nT)[CC], EvalAcc<CC, COUNT> &acc) {
    W gk = g & A.kmask;
    // rows outside the universe (SKIP slots and k-mers with more than v gaps, V20:689) get an
    // all-ones mismatch word: 32 (64) mismatches, counted nowhere
    if (popcw(gk) > A.v) gk = ~(W)0;
    W eA = 0, eC = 0, eG = 0, eT = 0;
    if (FORM == 2) { eA = ~(b0 | b1 | gk); eC = b0 & ~b1; eG = b1 & ~b0; eT = b0 & b1; }
#pragma unroll
    for (int c = 0; c < CC; c++) {
        W mm;
        if (FORM == 2) mm = (eT & nT[c]) | ((eG & nG[c]) | ((eC & nC[c]) | ((eA & nA[c]) | gk)));
        else mm = bfi<W>(b1, bfi<W>(b0, nT[c], nG[c]), bfi<W>(b0, nC[c], nA[c])) | gk;
        bool pp = mm == 0, ff, rr;
        if (VMODE == 0) {
            ff = rr = pp;
        } else if (VMODE == 1) {
            W t;
            if (FORM == 0 || sizeof(W) == 8) t = mm - 1u;
            else {
                uint32_t t32;
                asm("v_add_u32_e32 %0, -1, %1" : "=v"(t32) : "v"((uint32_t)mm));   // no carry-out: keeps `mm == 0` a plain v_cmp
                t = t32;
            }
            ff = (mm & (t | A.sF)) == 0;
            rr = (mm & (t | A.sR)) == 0;
        } else {
            bool le = popcw(mm) <= A.v;
            ff = le && (mm & A.sF) == 0;
            rr = le && (mm & A.sR) == 0;
        }
        acc.add(c, pp, ff, rr);
    }
}

template <typename W> struct Raw4 { W b0[4], b1[4], g[4]; };
template <typename W>
__device__ inline Raw4<W> load4(const FlyViewT<W> &V, int r) {
    Raw4<W> q;
#pragma unroll
    for (int i = 0; i < 4; i++) V.load(r + i, q.b0[i], q.b1[i], q.g[i]);
    return q;
}

template <int CC, int VMODE, int COUNT, bool PREFETCH, int FORM, typename W = uint32_t>
__global__ __launch_bounds__(kBlock) void eval_kernel(const EvalArgsT<W> A) {
    __shared__ uint32_t s_acc[3 * CC];
    const EvalItem it = A.items[blockIdx.x];
    W nA[CC], nC[CC], nG[CC], nT[CC];
    EvalAcc<CC, COUNT> acc;
    acc.clear();
#pragma unroll
    for (int c = 0; c < CC; c++) {
        const CandN<W> q = A.cand_n[it.cand0 + c];
        nA[c] = q.x; nC[c] = q.y; nG[c] = q.z; nT[c] = q.w;
        if (FORM == 1) {
            asm volatile("" : "+v"(nA[c]));
            asm volatile("" : "+v"(nC[c]));
            asm volatile("" : "+v"(nG[c]));
            asm volatile("" : "+v"(nT[c]));
        }
    }
    if (threadIdx.x < 3 * CC) s_acc[threadIdx.x] = 0;
    const FlyViewT<W> V(A.M, A.p0 + it.win, A.k, A.kmask);
    const int r0 = blockIdx.y * A.rows_per_split;
    const int r1 = r0 + A.rows_per_split < A.n_pad ? r0 + A.rows_per_split : A.n_pad;
    // 4 consecutive sequences per lane and iteration, 16-byte loads (n_pad % 4 == 0)
    int r = r0 + threadIdx.x * 4;
    Raw4<W> cur;
    if (PREFETCH && r < r1) cur = load4<W>(V, r);
#pragma unroll 1
    while (r < r1) {
        const int rn = r + kBlock * 4;
        Raw4<W> now;
        if (PREFETCH) {
            now = cur;
            if (rn < r1) cur = load4<W>(V, rn);       // next group in flight while this one computes
        } else {
            now = load4<W>(V, r);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const W b0 = now.b0[i], b1 = now.b1[i], g = now.g[i];
            eval_row<CC, VMODE, COUNT, FORM, W>(b0, b1, g, A, nA, nC, nG, nT, acc);
        }
        r = rn;
    }
    if (blockIdx.y == 0 && A.extra_off) {      // host-expanded IUPAC rows of this window
        const int e0 = A.extra_off[it.win], e1 = A.extra_off[it.win + 1];
        for (int eb = e0; eb < e1; eb += kBlock) {   // uniform trip count: COUNT 1 ballots need every lane
            int e = eb + threadIdx.x;
            W b0 = 0, b1 = 0, g = ~(W)0;
            if (e < e1) { b0 = A.extra_words[3 * e]; b1 = A.extra_words[3 * e + 1]; g = A.extra_words[3 * e + 2]; }
            eval_row<CC, VMODE, COUNT, FORM, W>(b0, b1, g, A, nA, nC, nG, nT, acc);
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CC; c++) {
        uint32_t x = acc.p[c], y = acc.f[c], z = acc.r[c];
        if (COUNT == 0) {
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) {
                x += __shfl_xor(x, s);
                y += __shfl_xor(y, s);
                z += __shfl_xor(z, s);
            }
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&s_acc[3 * c], x);
            atomicAdd(&s_acc[3 * c + 1], y - x);      // F_mis = F_raw - perfect
            atomicAdd(&s_acc[3 * c + 2], z - x);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * CC) {
        int oc = A.cand_out[it.cand0 + threadIdx.x / 3];
        uint32_t val = s_acc[threadIdx.x];
        if (oc >= 0 && val) atomicAdd(&A.out[(size_t)oc * 3 + threadIdx.x % 3], (unsigned long long)val);
    }
}


// ----------------------------------------------------------------------------------------------
// Patch planes: the rows `excl` takes out of the column-plane pass — k-mers with edge-gap repair or a ragged end
// (device-built patch list) and the host-expanded IUPAC rows — get one-hot planes of their own, per window and
// position: pplanes[poff + (j * 4 + base) * npw + word], 32 rows per word, npw a multiple of 8 words, zero padded.
// A bit-sliced workgroup then either covers a slice of the alignment's columns or the patch planes of its item's
// window (`WordTile`); the kernels do not care which.
// ----------------------------------------------------------------------------------------------
// one block per (window, run of kPatchRun patch rows) — the host lists the runs, so a window with a hundred thousand patch rows is
// spread over as many CUs as it has runs; 64 rows per wave pass, one ballot per (position, base); the padding words up to npw are
// written here as well (no fill beforehand)
constexpr int kPatchRun = 1024;
struct PatchRun { int32_t win, row0; };
template <typename W>
__global__ __launch_bounds__(kBlock) void patch_planes_kernel(const int32_t *__restrict__ off_a, const W *__restrict__ words_a,
                                                              const int32_t *__restrict__ off_b, const W *__restrict__ words_b,
                                                              const PatchWin *__restrict__ pwin, const PatchRun *__restrict__ runs, int k, int v,
                                                              uint32_t *__restrict__ pplanes, uint32_t *__restrict__ pvalid) {
    const PatchRun run = runs[blockIdx.x];
    const int win = run.win;
    const PatchWin pw = pwin[win];
    const int na = off_a ? off_a[win + 1] - off_a[win] : 0, nb = off_b ? off_b[win + 1] - off_b[win] : 0;
    const W kmask = kmask_of<W>(k);
    constexpr int NM = sizeof(W) == 4 ? 2 : 4;                                     // 64 (position, base) slots per register of ballots: k * 4 <= 64 * NM
    const int lane = threadIdx.x & 63;
    const int end = min(run.row0 + kPatchRun, pw.npw * 32);                        // npw is a multiple of 8 words: whole 64-row passes
    for (int r0 = run.row0 + (threadIdx.x >> 6) * 64; r0 < end; r0 += kBlock) {    // uniform per wave
        const int e = r0 + lane;
        W b0 = 0, b1 = 0, g = ~(W)0;
        if (e < na) {
            const size_t q = 3 * ((size_t)off_a[win] + e);
            b0 = words_a[q]; b1 = words_a[q + 1]; g = words_a[q + 2];
        } else if (e < na + nb) {
            const size_t q = 3 * ((size_t)off_b[win] + (e - na));
            b0 = words_b[q]; b1 = words_b[q + 1]; g = words_b[q + 2];
        }
        const bool ok = e < na + nb && !(g & WordTraits<W>::kSkip) && popcw((W)(g & kmask)) <= v;
        const unsigned long long okb = __ballot(ok);
        const int w0 = r0 >> 5;                                                    // r0 % 64 == 0
        if (lane == 0) *reinterpret_cast<uint2 *>(pvalid + pw.voff + w0) = uint2{(uint32_t)okb, (uint32_t)(okb >> 32)};
        // lane (j * 4 + b) keeps the ballot of its (position, base) and stores it: one 8-byte store per lane instead of 4k from lane 0
        unsigned long long mine[NM];
#pragma unroll
        for (int q = 0; q < NM; q++) mine[q] = 0;
        for (int j = 0; j < k; j++) {
            const bool base_here = ok && !((g >> j) & 1u);
            const uint32_t code = (uint32_t)((b0 >> j) & 1u) | ((uint32_t)((b1 >> j) & 1u) << 1);
#pragma unroll
            for (uint32_t b = 0; b < 4; b++) {
                const unsigned long long bal = __ballot(base_here && code == b);
                const int slot = j * 4 + (int)b;
                if (lane == (slot & 63)) {
#pragma unroll
                    for (int q = 0; q < NM; q++) if ((slot >> 6) == q) mine[q] = bal;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NM; q++)
            if (lane + 64 * q < k * 4)
                *reinterpret_cast<uint2 *>(pplanes + pw.poff + (size_t)(lane + 64 * q) * pw.npw + w0) = uint2{(uint32_t)mine[q], (uint32_t)(mine[q] >> 32)};
    }
}

// The same layout for the PLAIN column slices of the window's patch-list rows (the extra rows have none): bit (position j, base b) =
// the row carries base b at column p0 + win + j — straight out of the row-major bit planes; a gap and an IUPAC cell set none,
// exactly as the column planes hold them.  Validity = every patch-list row.  (Sliding evaluation: PatchArgs::qplanes.)
__global__ __launch_bounds__(kBlock) void plain_planes_kernel(const int32_t *__restrict__ off_a, const int32_t *__restrict__ rows_a,
                                                              const PatchWin *__restrict__ pwin, const PatchRun *__restrict__ runs,
                                                              const uint32_t *__restrict__ planes, int n_pad, int n_chunks, int p0, int k,
                                                              uint32_t *__restrict__ qplanes, uint32_t *__restrict__ qvalid) {
    const PatchRun run = runs[blockIdx.x];
    const int win = run.win;
    const PatchWin pw = pwin[win];
    const int na = off_a ? off_a[win + 1] - off_a[win] : 0;
    const uint32_t kmask = (1u << k) - 1u;
    const int lane = threadIdx.x & 63;
    const int p = p0 + win, c0 = p >> 5, sh = p & 31;
    const size_t np = (size_t)n_pad;
    const int end = min(run.row0 + kPatchRun, pw.npw * 32);
    for (int r0 = run.row0 + (threadIdx.x >> 6) * 64; r0 < end; r0 += kBlock) {    // uniform per wave
        const int e = r0 + lane;
        const bool ok = e < na;
        uint32_t m[4] = {0u, 0u, 0u, 0u};
        if (ok) {
            const int r = rows_a[(size_t)off_a[win] + e];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t lo = planes[((size_t)c0 * 4 + b) * np + r];
                const uint32_t hi = c0 + 1 < n_chunks ? planes[((size_t)(c0 + 1) * 4 + b) * np + r] : 0u;
                m[b] = (sh ? (lo >> sh) | (hi << (32 - sh)) : lo) & kmask;
            }
            const uint32_t multi = (m[0] & (m[1] | m[2] | m[3])) | (m[1] & (m[2] | m[3])) | (m[2] & m[3]);     // IUPAC cells: none set, as colplane_kernel leaves them
#pragma unroll
            for (int b = 0; b < 4; b++) m[b] &= ~multi;
        }
        const unsigned long long okb = __ballot(ok);
        const int w0 = r0 >> 5;
        if (lane == 0) *reinterpret_cast<uint2 *>(qvalid + pw.voff + w0) = uint2{(uint32_t)okb, (uint32_t)(okb >> 32)};
        unsigned long long mine = 0, mine2 = 0;
        for (int j = 0; j < k; j++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const unsigned long long bal = __ballot((m[b] >> j) & 1u);
                const int slot = j * 4 + b;
                if (lane == (slot & 63)) { if (slot < 64) mine = bal; else mine2 = bal; }
            }
        }
        if (lane < k * 4) *reinterpret_cast<uint2 *>(qplanes + pw.poff + (size_t)lane * pw.npw + w0) = uint2{(uint32_t)mine, (uint32_t)(mine >> 32)};
        if (lane + 64 < k * 4)
            *reinterpret_cast<uint2 *>(qplanes + pw.poff + (size_t)(lane + 64) * pw.npw + w0) = uint2{(uint32_t)mine2, (uint32_t)(mine2 >> 32)};
    }
}

// ----------------------------------------------------------------------------------------------
// (4b) bit-sliced evaluation: 32 sequences per register word
// ----------------------------------------------------------------------------------------------
// For the sequences whose k-mer at window w is the plain column slice (everything `excl` does not
// flag) the symbol at window position j is column p0+w+j of the alignment, so the evaluation runs
// on the one-hot COLUMN planes (A, C, G, T; a gap sets none): "sequence matches symbol s at position
// j" is the OR of the planes of s's bases — for a concrete symbol just one loaded word, no ALU work.
// Mismatch counts are bit-sliced saturating counters (t1 = ">= 1", t2 = ">= 2", t3, t4 likewise: LV =
// v+1 levels), the strict-position sets two more words, the three coverage counters popcounts at
// the end.  Rows with more than v gaps never count (V20:689): build_windows folded them into `excl`.
// Two kernels share this scheme:
//   eval_chain_kernel — the candidates of an item form a NESTED chain (a refinement chain): one pass
//       over the k positions for the most degenerate candidate, then one bit plane per refinement step;
//   eval_bits_kernel  — any 8 candidates: per position the match word of every possible symbol (11
//       VALU per word, shared by the candidates), each candidate picks its word by register index.
// The inputs (N*L/2 bytes of planes) stay in L2 / Infinity Cache across the windows of a launch.


struct EvalBitsArgs {
    const unsigned long long *cols;    // [n_cols][4][nw]
    const unsigned long long *excl;    // [W][nw]
    int nw, p0, k, v;
    const EvalItem *items;
    const uint32_t *cand_symT;         // [item][32] u32: nibble c of word j = symbol of candidate c at position j
    const int32_t *cand_out;
    uint32_t sF, sR;
    unsigned long long *out;
    BlockMap map;                      // row slices per item (ny; ny_pad = rounded up to 1, 2, 4 or a multiple of 8), items
    const uint32_t *diff_mask;         // [item] positions where the item's candidates do not all carry the same symbol
    const int32_t *item_ids;           // items this launch covers (null: all of them)
    PatchArgs patch;                   // patch / IUPAC rows: the first patch.n_blocks workgroups run on their planes
    uint32_t *mask_f, *mask_r;         // MASKS form: [candidate][2 nw] "not covered" bit sets instead of counts (mp_eval_masks)
    int n_rows;
    unsigned long long *clear;         // mp_eval_launch_rotating: the next launch's counter block (chainbody.hpp: clear_counters)
    uint32_t n_clear;
};


// GW 32-bit words (32 sequences each) per thread, LV = v + 1 saturating counter levels.
// CHAIN: positions where all 8 candidates carry the same symbol (not in diff_mask, from the host) update ONE
// shared set of counters (c1..) with one match word; only the positions where they differ run the symbol
// table, the register-indexed select and the per-candidate updates.  The two counter sets add up exactly at the
// end (saturating sums: >=1: a1|b1, >=2: a2|b2|a1&b1, ...).
// MASKS: the same pass, but what leaves the thread is the bit set itself — per candidate the words "not covered as a forward /
// reverse primer" (V20:689-698, 1107-1127: more than v gaps, too many mismatches, or a mismatch at a strict position) of the plain
// rows, 32 per word exactly as the registers hold them: ~(valid & ~far & ~strict) inside the alignment's rows.  `excl` rows come
// out as 1; the ones among them that are not plain column slices (the window's patch list) are then ASSIGNED by mask_patch_kernel.
template <int LV, int GW, bool CHAIN, int D, bool MASKS = false>
__global__ __launch_bounds__(kBlock) void eval_bits_kernel(const EvalBitsArgs A) {
    constexpr int CC = 8;
    __shared__ uint32_t s_part[kBlock / 64][3 * CC / 2];
    if (!MASKS) clear_counters(A.clear, A.n_clear, blockIdx.x, gridDim.x);
    const bool on_patch = (int)blockIdx.x < A.patch.n_blocks;
    int slice, idx, word0;
    if (on_patch) {                                    // a wave per patch unit: everything below is wave-uniform
        const int unit = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)));
        idx = unit / A.patch.per_item;
        slice = unit % A.patch.per_item;
        if (idx >= A.map.n_items) return;
        word0 = (slice * 64 + (int)(threadIdx.x & 63)) * GW;
    } else {
        if (!map_block(A.map, blockIdx.x - A.patch.n_blocks, slice, idx)) return;
        word0 = (slice * kBlock + threadIdx.x) * GW;
    }
    const int item = A.item_ids ? A.item_ids[idx] : idx;
    const EvalItem it = A.items[item];
    const WordTile T = on_patch ? patch_tile(A.patch, it.win, word0) : column_tile(A.cols, A.excl, A.nw, A.p0, it.win, word0);
    if (on_patch && slice * 64 * GW >= (int)T.stride) return;               // nothing of this window's patch planes left for the wave
    const size_t nw32 = T.stride;                     // 32-bit words per plane row of the tile
    const bool live = T.live;
    const uint32_t kbits = (1u << A.k) - 1u;           // k <= MP_MAX_K = 31
    const uint32_t diff = CHAIN ? (__builtin_amdgcn_readfirstlane(A.diff_mask[item]) & kbits) : kbits;
    const uint32_t same = kbits & ~diff;
    uint32_t accP[CC], accF[CC], accR[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) accP[c] = accF[c] = accR[c] = 0;
    if (live) {
        uint32_t t1[CC][GW], t2[CC][GW], t3[CC][GW], t4[CC][GW], sf[CC][GW], sr[CC][GW];
        uint32_t c1[GW], c2[GW], c3[GW], c4[GW], csf[GW], csr[GW];      // counters of the positions all candidates share
#pragma unroll
        for (int i = 0; i < GW; i++) {
            c1[i] = c2[i] = c3[i] = c4[i] = csf[i] = csr[i] = 0;
#pragma unroll
            for (int c = 0; c < CC; c++) t1[c][i] = t2[c][i] = t3[c][i] = t4[c][i] = sf[c][i] = sr[c][i] = 0;
        }
        const uint32_t *Pw = T.planes;
        const uint32_t *symrow = A.cand_symT + (size_t)item * 32;
        // (1) positions where all candidates of the item carry the same symbol: one match word for all of them,
        // branch-free (the symbol's base masks are wave-uniform)
        if (CHAIN) {
            uint32_t rem = same;
#pragma unroll 1
            while (rem) {
                int js[D]; bool has[D];
                uint32_t pl[D][4][GW], sws[D];
#pragma unroll
                for (int u = 0; u < D; u++) {           // D positions of loads in flight before the first is consumed
                    has[u] = rem != 0u;
                    js[u] = has[u] ? __builtin_ctz(rem) : js[0];
                    rem &= rem - 1u;
                    const uint32_t *P = Pw + (size_t)js[u] * 4 * nw32;
#pragma unroll
                    for (int b = 0; b < 4; b++)
#pragma unroll
                        for (int i = 0; i < GW; i++) pl[u][b][i] = P[b * nw32 + i];
                    sws[u] = symrow[js[u]];
                }
#pragma unroll
                for (int u = 0; u < D; u++) {
                    if (!has[u]) break;
                    const uint32_t sy = __builtin_amdgcn_readfirstlane(sws[u]) & 15u;
                    const uint32_t kA = (sy & 1u) ? 0xFFFFFFFFu : 0u, kC = (sy & 2u) ? 0xFFFFFFFFu : 0u;
                    const uint32_t kG = (sy & 4u) ? 0xFFFFFFFFu : 0u, kT = (sy & 8u) ? 0xFFFFFFFFu : 0u;
                    const uint32_t fF = ((A.sF >> js[u]) & 1u) ? 0xFFFFFFFFu : 0u, fR = ((A.sR >> js[u]) & 1u) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                    for (int i = 0; i < GW; i++) {
                        uint32_t m = pl[u][0][i] & kA;
                        m = __builtin_amdgcn_bitop3_b32(pl[u][1][i], kC, m, kLutAndOr);
                        m = __builtin_amdgcn_bitop3_b32(pl[u][2][i], kG, m, kLutAndOr);
                        m = __builtin_amdgcn_bitop3_b32(pl[u][3][i], kT, m, kLutAndOr);
                        count_unmatched<LV>(c1[i], c2[i], c3[i], c4[i], m);
                        csf[i] = __builtin_amdgcn_bitop3_b32(csf[i], m, fF, kLutOrNotAnd);
                        csr[i] = __builtin_amdgcn_bitop3_b32(csr[i], m, fR, kLutOrNotAnd);
                    }
                }
            }
        }
        // (2) the other positions: the match word of every possible symbol (11 x GW VALU, shared by all
        // candidates); a candidate then picks its word by a wave-uniform register index
        {
            uint32_t rem = diff;
#pragma unroll 1
            while (rem) {
                int js[D]; bool has[D];
                uint32_t pl[D][4][GW], sws[D];
#pragma unroll
                for (int u = 0; u < D; u++) {
                    has[u] = rem != 0u;
                    js[u] = has[u] ? __builtin_ctz(rem) : js[0];
                    rem &= rem - 1u;
                    const uint32_t *P = Pw + (size_t)js[u] * 4 * nw32;
#pragma unroll
                    for (int b = 0; b < 4; b++)
#pragma unroll
                        for (int i = 0; i < GW; i++) pl[u][b][i] = P[b * nw32 + i];
                    sws[u] = symrow[js[u]];
                }
#pragma unroll
                for (int u = 0; u < D; u++) {
                    if (!has[u]) break;
                    const uint32_t sw = __builtin_amdgcn_readfirstlane(sws[u]);
                    const bool jF = (A.sF >> js[u]) & 1u, jR = (A.sR >> js[u]) & 1u;
                    uint32_t tab[16][GW];
#pragma unroll
                    for (int i = 0; i < GW; i++) {
                        const uint32_t a = pl[u][0][i], cc = pl[u][1][i], g = pl[u][2][i], t = pl[u][3][i];
                        tab[0][i] = 0u;
                        tab[1][i] = a; tab[2][i] = cc; tab[4][i] = g; tab[8][i] = t;
                        tab[3][i] = a | cc; tab[5][i] = a | g; tab[9][i] = a | t;
                        tab[6][i] = cc | g; tab[10][i] = cc | t; tab[12][i] = g | t;
                        tab[7][i] = __builtin_amdgcn_bitop3_b32(a, cc, g, kLutOr3);
                        tab[11][i] = __builtin_amdgcn_bitop3_b32(a, cc, t, kLutOr3);
                        tab[13][i] = __builtin_amdgcn_bitop3_b32(a, g, t, kLutOr3);
                        tab[14][i] = __builtin_amdgcn_bitop3_b32(cc, g, t, kLutOr3);
                        tab[15][i] = tab[3][i] | tab[12][i];
                    }
                    uint32_t m[CC][GW];
#pragma unroll
                    for (int c = 0; c < CC; c++) {
                        const uint32_t sy = (sw >> (4 * c)) & 15u;
#pragma unroll
                        for (int i = 0; i < GW; i++) m[c][i] = tab[sy][i];
#pragma unroll
                        for (int i = 0; i < GW; i++) count_unmatched<LV>(t1[c][i], t2[c][i], t3[c][i], t4[c][i], m[c][i]);
                    }
                    if (jF) {
#pragma unroll
                        for (int c = 0; c < CC; c++)
#pragma unroll
                            for (int i = 0; i < GW; i++) sf[c][i] = __builtin_amdgcn_bitop3_b32(sf[c][i], m[c][i], m[c][i], kLutOrNot);
                    }
                    if (jR) {
#pragma unroll
                        for (int c = 0; c < CC; c++)
#pragma unroll
                            for (int i = 0; i < GW; i++) sr[c][i] = __builtin_amdgcn_bitop3_b32(sr[c][i], m[c][i], m[c][i], kLutOrNot);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < GW; i++) {
            const uint32_t valid = T.mask[i] ^ T.mask_flip;
            const int row0 = (word0 + i) * 32;                 // MASKS: bits of rows past the alignment's last stay 0
            const uint32_t inside = row0 + 32 <= A.n_rows ? 0xFFFFFFFFu : (row0 >= A.n_rows ? 0u : ((1u << (A.n_rows - row0)) - 1u));
#pragma unroll
            for (int c = 0; c < CC; c++) {
                // exact saturating sum of the shared and the per-candidate counters
                const uint32_t a1 = t1[c][i] | c1[i];
                uint32_t far = a1;
                if (LV >= 2) far = t2[c][i] | c2[i] | (t1[c][i] & c1[i]);
                if (LV >= 3) far = t3[c][i] | c3[i] | (t2[c][i] & c1[i]) | (t1[c][i] & c2[i]);
                if (LV >= 4) far = t4[c][i] | c4[i] | (t3[c][i] & c1[i]) | (t2[c][i] & c2[i]) | (t1[c][i] & c3[i]);
                const uint32_t bf = sf[c][i] | csf[i], br = sr[c][i] | csr[i];
                if (MASKS) {
                    const int oc = A.cand_out[it.cand0 + c];
                    if (oc >= 0) {
                        const size_t at = (size_t)oc * nw32 + (size_t)(word0 + i);
                        A.mask_f[at] = ~__builtin_amdgcn_bitop3_b32(valid, far, bf, kLutAndNotNot) & inside;
                        A.mask_r[at] = ~__builtin_amdgcn_bitop3_b32(valid, far, br, kLutAndNotNot) & inside;
                    }
                    continue;
                }
                accP[c] += __popc(valid & ~a1);
                accF[c] += __popc(__builtin_amdgcn_bitop3_b32(valid, far, bf, kLutAndNotNot));
                accR[c] += __popc(__builtin_amdgcn_bitop3_b32(valid, far, br, kLutAndNotNot));
            }
        }
    }
    if (MASKS) return;
    if (on_patch) wave_commit<GW>(accP, accF, accR, s_part[threadIdx.x >> 6], A.cand_out + it.cand0, A.out);
    else block_commit<GW>(accP, accF, accR, s_part, A.cand_out + it.cand0, A.out);
}

// (4c) bit-sliced evaluation of a NESTED chain: chainbody.hpp (the workgroup's work as a device function: the sliding launch runs its
// patch units through the same code)
template <int LV, int GW, int D>
__global__ __launch_bounds__(kBlock) void eval_chain_kernel(const EvalChainArgs A) {
    __shared__ uint32_t s_part[kBlock / 64][12];
    clear_counters(A.clear, A.n_clear, blockIdx.x, gridDim.x);
    eval_chain_block<LV, GW, D>(A, s_part, blockIdx.x);
}

// The same for chains of more than 8 members (up to 64): the members are counted 8 at a time — one group of output
// slots, one commit — and the counters carry over from group to group, so the first pass runs once per chain instead
// of once per 8 members.  A separate kernel because the extra loop costs registers (120 instead of 85 VGPRs at 8 words
// per thread): chains of up to 8 members, the common case, keep the leaner one.
template <int LV, int GW, int D>
__global__ __launch_bounds__(kBlock) void eval_chain_long_kernel(const EvalChainArgs A) {
    static_assert(GW <= 8, "plane rows are padded to multiples of 8 words");
    constexpr int CC = 8;
    __shared__ uint32_t s_part[kBlock / 64][3 * CC / 2];
    clear_counters(A.clear, A.n_clear, blockIdx.x, gridDim.x);
    // the patch units come first; a second run of them (sliding evaluation) subtracts the plain slices of the items that slide
    const int patch_blocks = A.patch.n_blocks + A.patch.neg_blocks;
    const bool on_patch = (int)blockIdx.x < patch_blocks;
    const bool negative = on_patch && (int)blockIdx.x >= A.patch.n_blocks;
    int slice, item, word0;
    if (on_patch) {                                    // a wave per patch unit: everything below is wave-uniform
        const int pb = (int)blockIdx.x - (negative ? A.patch.n_blocks : 0);
        const int unit = __builtin_amdgcn_readfirstlane(pb * (kBlock / 64) + (int)(threadIdx.x >> 6));
        item = unit / A.patch.per_item;
        slice = unit % A.patch.per_item;
        if (item >= (negative ? A.n_neg : A.map.n_items)) return;
        word0 = (slice * 64 + (int)(threadIdx.x & 63)) * GW;
    } else {
        if (!map_block(A.map, blockIdx.x - patch_blocks, slice, item)) return;
        word0 = (slice * kBlock + threadIdx.x) * GW;
    }
    const ChainItem it = negative ? A.neg_items[item] : A.items[item];
    const WordTile T = !on_patch ? column_tile(A.cols, A.excl, A.nw, A.p0, it.win, word0)
                                 : (negative ? plain_tile(A.patch, it.win, word0) : patch_tile(A.patch, it.win, word0));
    if (on_patch && slice * 64 * GW >= (int)T.stride) return;               // nothing of this window's patch planes left for the wave
    const size_t nw32 = T.stride;
    const bool live = T.live;
    uint32_t t1[GW], t2[GW], t3[GW], t4[GW], sf[GW], sr[GW], valid[GW], cur[GW];
#pragma unroll
    for (int i = 0; i < GW; i++) t1[i] = t2[i] = t3[i] = t4[i] = sf[i] = sr[i] = valid[i] = cur[i] = 0;
    const uint32_t *Pw = T.planes;
    const uint32_t *ev = A.events + it.ev0;
    int e = 0;
    uint32_t evw = it.n_ev ? ev[0] : (1u << 8);
    if (live) {
        const unsigned long long sy_lo = it.sym[0] | ((unsigned long long)it.sym[1] << 32);
        const unsigned long long sy_hi = it.sym[2] | ((unsigned long long)it.sym[3] << 32);
        // (1) the first candidate over all k positions, grouped by the number of bases of its symbol there
        chain_first_pass<LV, GW, D, 1>(it.pos1, Pw, nw32, sy_lo, sy_hi, A.sF, A.sR, t1, t2, t3, t4, sf, sr);
        chain_first_pass<LV, GW, (D + 1) / 2, 2>(it.pos2, Pw, nw32, sy_lo, sy_hi, A.sF, A.sR, t1, t2, t3, t4, sf, sr);
        chain_first_pass<LV, GW, (D + 3) / 4, 4>(it.pos4, Pw, nw32, sy_lo, sy_hi, A.sF, A.sR, t1, t2, t3, t4, sf, sr);
#pragma unroll
        for (int i = 0; i < GW; i++) valid[i] = T.mask[i] ^ T.mask_flip;
        const uint32_t *P = Pw + ((size_t)(evw & 255u) * 4 + (size_t)__builtin_ctz(((evw >> 8) & 15u) | 16u)) * nw32;
#pragma unroll
        for (int i = 0; i < GW; i++) cur[i] = P[i];
    }
    // (2) walk down the chain, 8 members (one group of output slots) at a time; the counters carry over between groups
#pragma unroll 1
    for (int g0 = 0; g0 < it.n_steps; g0 += CC) {
        uint32_t accP[CC], accF[CC], accR[CC];
#pragma unroll
        for (int c = 0; c < CC; c++) accP[c] = accF[c] = accR[c] = 0;
        if (live) {
#pragma unroll
            for (int s = 0; s < CC; s++) {
                const int step = g0 + s;
                if (step >= it.n_steps) break;
                if (step > 0) {                            // apply the events of this step, then count the member
#pragma unroll 1
                    while (e < it.n_ev && (int)(evw >> 16) == step) {
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
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    const uint32_t far = LV == 1 ? t1[i] : (LV == 2 ? t2[i] : (LV == 3 ? t3[i] : t4[i]));
                    accP[s] += __popc(valid[i] & ~t1[i]);
                    accF[s] += __popc(__builtin_amdgcn_bitop3_b32(valid[i], far, sf[i], kLutAndNotNot));
                    accR[s] += __popc(__builtin_amdgcn_bitop3_b32(valid[i], far, sr[i], kLutAndNotNot));
                }
            }
        }
        if (on_patch) {
            wave_commit<GW>(accP, accF, accR, s_part[threadIdx.x >> 6], A.cand_out + it.cand0 + g0, A.out, negative);
        } else {
            if (g0) __syncthreads();                       // the previous group's totals have been read out of s_part
            block_commit<GW>(accP, accF, accR, s_part, A.cand_out + it.cand0 + g0, A.out);
        }
    }
}

// ----------------------------------------------------------------------------------------------
// (4d) per-window base and nearest-neighbour counts (mp_window_stats; state_matrix / trans_matrix,
// V20:541-577).  Same universe, same planes: freq[w][b][j] = popcount(valid_w & plane[col w+j][b]),
// nn[w][j][a][b] = popcount(valid_w & plane[col w+j][a] & plane[col w+j+1][b]) over the plain rows, plus
// the same pass over the windows' patch planes.  20 counters per position: a thread sums its GW words,
// packs two 16-bit counts per register, six DPP adds give the wave total, lane 63 adds it to the block's
// LDS table, the block adds its k x 20 totals to the global counters.
// ----------------------------------------------------------------------------------------------
struct StatsArgs {
    const unsigned long long *cols;    // [n_cols][4][nw]
    const unsigned long long *excl;    // [W][nw]
    int nw, p0, k, v;
    BlockMap map;                      // items = windows (window_stats_kernel) or groups of G windows (window_stats_group_kernel)
    int n_win;
    unsigned long long *freq;          // [W][4][k]
    unsigned long long *nn;            // [W][k-1][16]
    PatchArgs patch;
};

__device__ __forceinline__ void stats_flush(const StatsArgs &A, int win, const uint32_t (*s_cnt)[20]) {
    for (int t = threadIdx.x; t < A.k * 20; t += kBlock) {
        const int j = t / 20, q = t % 20;
        const uint32_t val = s_cnt[j][q];
        if (!val) continue;
        if (q < 4) atomicAdd(&A.freq[((size_t)win * 4 + q) * A.k + j], (unsigned long long)val);
        else if (j + 1 < A.k) atomicAdd(&A.nn[((size_t)win * (A.k - 1) + j) * 16 + (q - 4)], (unsigned long long)val);
    }
}

template <int GW>
__device__ __forceinline__ void stats_window_body(const StatsArgs &A, uint32_t (*s_cnt)[20]) {
    const bool on_patch = (int)blockIdx.x < A.patch.n_blocks;
    int slice, win;
    if (on_patch) {
        win = blockIdx.x / A.patch.per_item;
        slice = blockIdx.x % A.patch.per_item;
        if (win >= A.n_win) return;                   // (A.map counts window GROUPS in window_stats_group_kernel)
    } else if (!map_block(A.map, blockIdx.x - A.patch.n_blocks, slice, win)) {
        return;
    }
    const int word0 = (slice * kBlock + threadIdx.x) * GW;
    const WordTile T = on_patch ? patch_tile(A.patch, win, word0) : column_tile(A.cols, A.excl, A.nw, A.p0, win, word0);
    if (on_patch && (int)(slice * kBlock * GW) >= (int)T.stride) return;
    for (int t = threadIdx.x; t < MP_MAX_K * 20; t += kBlock) (&s_cnt[0][0])[t] = 0;
    __syncthreads();
    const size_t nw32 = T.stride;
    const bool live = T.live;
    const uint32_t *Pw = T.planes;
    uint32_t valid[GW], cur[4][GW], nxt[4][GW];
#pragma unroll
    for (int i = 0; i < GW; i++) {
        valid[i] = live ? (T.mask[i] ^ T.mask_flip) : 0u;
#pragma unroll
        for (int b = 0; b < 4; b++) cur[b][i] = live ? (valid[i] & Pw[b * nw32 + i]) : 0u;
    }
#pragma unroll 1
    for (int j = 0; j < A.k; j++) {
        const bool more = j + 1 < A.k;
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int i = 0; i < GW; i++) nxt[b][i] = (live && more) ? (valid[i] & Pw[((size_t)(j + 1) * 4 + b) * nw32 + i]) : 0u;
        uint32_t cnt[20];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            cnt[b] = 0;
#pragma unroll
            for (int i = 0; i < GW; i++) cnt[b] += __popc(cur[b][i]);
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                cnt[4 + a * 4 + b] = 0;
#pragma unroll
                for (int i = 0; i < GW; i++) cnt[4 + a * 4 + b] += __popc(cur[a][i] & nxt[b][i]);
            }
        static_assert(64 * 32 * GW < 65536, "packed wave sums must fit 16 bits");
        uint32_t tot[10];
#pragma unroll
        for (int q = 0; q < 10; q++) tot[q] = wave_sum_lane63(cnt[2 * q] | (cnt[2 * q + 1] << 16));
        if ((threadIdx.x & 63) == 63) {
#pragma unroll
            for (int q = 0; q < 10; q++) {
                if (tot[q] & 0xFFFFu) atomicAdd(&s_cnt[j][2 * q], tot[q] & 0xFFFFu);
                if (tot[q] >> 16) atomicAdd(&s_cnt[j][2 * q + 1], tot[q] >> 16);
            }
        }
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int i = 0; i < GW; i++) cur[b][i] = nxt[b][i];
    }
    __syncthreads();
    stats_flush(A, win, s_cnt);
}

template <int GW>
__global__ __launch_bounds__(kBlock) void window_stats_kernel(const StatsArgs A) {
    __shared__ uint32_t s_cnt[MP_MAX_K][20];           // [position][4 base counts, then 16 pair counts (j, j+1)]
    stats_window_body<GW>(A, s_cnt);
}

// Wave totals of 12 registers, TRANSPOSED (the reduction of evalslide.hip's commit: quad-masked DPP adds across the quads of a row, then
// inside the quads, then v_permlane16/32_swap across the rows — a step that adds partner lanes also halves the registers): 31 instructions
// instead of 12 x 6 DPP adds.  Afterwards lane L (L < 48, L % 4 == 0) holds the wave's total of register (L >> 4) * 4 + ((L >> 2) & 3).
__device__ __forceinline__ uint32_t wave_sum12_transposed(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, uint32_t x5, uint32_t x6,
                                                          uint32_t x7, uint32_t x8, uint32_t x9, uint32_t x10, uint32_t x11) {
    asm volatile(
        "s_nop 1\n\t"
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
        "v_add_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_u32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_u32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_u32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_u32_dpp %4, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_u32_dpp %8, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8), "+v"(x9), "+v"(x10), "+v"(x11));
    typedef unsigned int u32pair __attribute__((ext_vector_type(2)));
    const u32pair s01 = __builtin_amdgcn_permlane16_swap(x0, x4, false, false);
    const uint32_t a = s01.x + s01.y;
    const u32pair s22 = __builtin_amdgcn_permlane16_swap(x8, x8, false, false);
    const uint32_t b = s22.x + s22.y;
    const u32pair h = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return h.x + h.y;
}

// [r6] The same counts for G CONSECUTIVE windows per workgroup: window w uses column w + j at position j, so G neighbouring windows share all but
// G - 1 of their k + G - 1 columns — the workgroup walks the columns once, every plane word is loaded once and counted under the validity words
// of the (up to G) windows it belongs to.  The per-window kernel above re-reads every plane k times through L2 (9.4 GB at 10^6 x 1000, k = 18:
// 2.27 ms with the vector ALUs 46 % busy); here (k + G - 1) / (G k) of that.  The arithmetic per (window, position) is the same 20 AND + popcount
// pairs (the pair counts as one v_bitop3 a & b & valid each).  Plain rows only: the patch planes keep the kernel above (its patch blocks).
template <int GW, int G>
__global__ __launch_bounds__(kBlock) void window_stats_group_kernel(const StatsArgs A) {
    // per window of the group and position: 10 words of two 16-bit counts (base counts 0-3, pair counts 4-19; a workgroup covers
    // kBlock x 32 GW <= 32768 rows, so a field never carries) + 2 words of padding (the reduction works on 12 registers)
    static_assert(kBlock * 32 * GW <= 32768, "two 16-bit counts per word");
    __shared__ uint32_t s_grp[G][MP_MAX_K][12];
    __shared__ uint32_t s_one[MP_MAX_K][20];
    if ((int)blockIdx.x < A.patch.n_blocks) {                       // the patch planes: per window, as before
        stats_window_body<GW>(A, s_one);
        return;
    }
    int slice, grp;
    if (!map_block(A.map, blockIdx.x - (unsigned)A.patch.n_blocks, slice, grp)) return;
    const int w0 = grp * G, n_here = min(G, A.n_win - w0), k = A.k;
    const int word0 = (slice * kBlock + (int)threadIdx.x) * GW, lane = (int)(threadIdx.x & 63);
    const size_t nw32 = (size_t)A.nw * 2;
    const bool live = word0 < (int)nw32;
    for (int t = threadIdx.x; t < G * MP_MAX_K * 12; t += kBlock) (&s_grp[0][0][0])[t] = 0;
    __syncthreads();
    const uint32_t *Pw = reinterpret_cast<const uint32_t *>(A.cols) + ((size_t)(A.p0 + w0) * 4) * nw32 + word0;
    const uint32_t *Ex = reinterpret_cast<const uint32_t *>(A.excl) + (size_t)w0 * nw32 + word0;
    uint32_t valid[G][GW], cur[4][GW], nxt[4][GW];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int i = 0; i < GW; i++) valid[g][i] = (live && g < n_here) ? ~Ex[(size_t)g * nw32 + i] : 0u;
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int i = 0; i < GW; i++) cur[b][i] = live ? Pw[b * nw32 + i] : 0u;
    const int n_col = n_here + k - 1;
    const bool adder = lane < 48 && (lane & 3) == 0;
    const int my_word = (lane >> 4) * 4 + ((lane >> 2) & 3);
#pragma unroll 1
    for (int c = 0; c < n_col; c++) {
        const bool more = c + 1 < n_col;
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int i = 0; i < GW; i++) nxt[b][i] = (live && more) ? Pw[((size_t)(c + 1) * 4 + b) * nw32 + i] : 0u;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int j = c - g;                                   // the column's position in window w0 + g (wave-uniform)
            if (g >= n_here || j < 0 || j >= k) continue;
            uint32_t cnt[20];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                cnt[b] = 0;
#pragma unroll
                for (int i = 0; i < GW; i++) cnt[b] += __popc(cur[b][i] & valid[g][i]);
            }
            // (at the window's last position nxt is the next window's column or zero: those pair counts are never read — stats_flush_packed)
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    cnt[4 + a * 4 + b] = 0;
#pragma unroll
                    for (int i = 0; i < GW; i++) cnt[4 + a * 4 + b] += __popc(__builtin_amdgcn_bitop3_b32(cur[a][i], nxt[b][i], valid[g][i], 0x80));
                }
            const uint32_t tot = wave_sum12_transposed(cnt[0] | (cnt[1] << 16), cnt[2] | (cnt[3] << 16), cnt[4] | (cnt[5] << 16), cnt[6] | (cnt[7] << 16),
                                                       cnt[8] | (cnt[9] << 16), cnt[10] | (cnt[11] << 16), cnt[12] | (cnt[13] << 16), cnt[14] | (cnt[15] << 16),
                                                       cnt[16] | (cnt[17] << 16), cnt[18] | (cnt[19] << 16), 0u, 0u);
            if (adder) atomicAdd(&s_grp[g][j][my_word], tot);
        }
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int i = 0; i < GW; i++) cur[b][i] = nxt[b][i];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n_here * k * 20; t += kBlock) {
        const int g = t / (k * 20), r = t % (k * 20), j = r / 20, q = r % 20, win = w0 + g;
        const uint32_t val = (s_grp[g][j][q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
        if (!val) continue;
        if (q < 4) atomicAdd(&A.freq[((size_t)win * 4 + q) * k + j], (unsigned long long)val);
        else if (j + 1 < k) atomicAdd(&A.nn[((size_t)win * (k - 1) + j) * 16 + (q - 4)], (unsigned long long)val);
    }
}

// Per-sequence coverage masks (mp_eval_masks): thread = sequence, the wave's 64 "not covered" bits
// go out as one 64-bit word per candidate straight from the ballot (blocks are 64-row aligned), so
// there are no atomics; works on the window words, i.e. after edge-gap repair, for any v.
// decision of one (candidate, k-mer) pair for the masks: would the sequence appear in gap_seq_id / the F (R) dict of
// non_coverage_seq_id (V20:689-698, 1107-1127)?
template <typename W>
__device__ inline void mask_decide(W b0, W b1, W gk, const CandN<W> q, const EvalArgsT<W> &A, bool &bad_f, bool &bad_r) {
    const bool gap_row = popcw(gk) > A.v;
    const W mm = bfi<W>(b1, bfi<W>(b0, q.w, q.z), bfi<W>(b0, q.y, q.x)) | gk;
    const int d = popcw(mm);
    const bool near = d <= A.v;
    bad_f = gap_row || !(near && (d == 0 || !(mm & A.sF)));
    bad_r = gap_row || !(near && (d == 0 || !(mm & A.sR)));
}

// thread = row of one item's window, 8 candidates: plain column slices straight from the planes (fast_words, no divergence);
// the repaired / ragged rows of the window are added by mask_patch_kernel from the patch list, IUPAC rows stay 0 (host).
template <int CC, typename W = uint32_t>
__global__ __launch_bounds__(kBlock) void mask_rows_kernel(const EvalArgsT<W> A, int n_rows, unsigned long long *__restrict__ not_f,
                                                           unsigned long long *__restrict__ not_r) {
    const EvalItem it = A.items[blockIdx.x];
    const int r = blockIdx.y * kBlock + threadIdx.x;             // n_pad is a multiple of kBlock
    const size_t nw = (size_t)A.n_pad / 64, np = (size_t)A.n_pad;
    const int p = A.p0 + it.win;
    W b0 = 0, b1 = 0, g = 0;
    bool plain = false;
    if (r < n_rows)
        plain = fast_words<W>(p, A.k, A.kmask, A.M.rlen[r], load_plane_words<W>(A.M.planes + ((size_t)(p >> 5) * 4) * np + r, np), b0, b1, g);
#pragma unroll
    for (int c = 0; c < CC; c++) {
        bool bad_f, bad_r;
        mask_decide<W>(b0, b1, g, A.cand_n[it.cand0 + c], A, bad_f, bad_r);
        const unsigned long long wf = __ballot(plain && bad_f), wr = __ballot(plain && bad_r);
        const int oc = A.cand_out[it.cand0 + c];
        if ((threadIdx.x & 63) == 0 && oc >= 0) {
            not_f[(size_t)oc * nw + (size_t)(r >> 6)] = wf;
            not_r[(size_t)oc * nw + (size_t)(r >> 6)] = wr;
        }
    }
}

// one workgroup per item: the window's slow pairs (patch list: rows and window words from repair_kernel)
template <int CC, typename W = uint32_t>
__global__ __launch_bounds__(kBlock) void mask_patch_kernel(const EvalArgsT<W> A, const int32_t *__restrict__ patch_off,
                                                            const int32_t *__restrict__ patch_rows, const W *__restrict__ patch_words,
                                                            unsigned long long *__restrict__ not_f, unsigned long long *__restrict__ not_r) {
    const EvalItem it = A.items[blockIdx.x];
    const size_t nw = (size_t)A.n_pad / 64;
    for (int e = patch_off[it.win] + threadIdx.x; e < patch_off[it.win + 1]; e += kBlock) {
        const W b0 = patch_words[3 * (size_t)e], b1 = patch_words[3 * (size_t)e + 1], g = patch_words[3 * (size_t)e + 2];
        const bool skip = (g & WordTraits<W>::kSkip) != 0;        // IUPAC k-mer (the host owns it) or a too-short row: both bits 0
        const int r = patch_rows[e];
        const unsigned long long bit = 1ull << (r & 63);
        for (int c = 0; c < CC; c++) {
            const int oc = A.cand_out[it.cand0 + c];
            if (oc < 0) continue;
            bool bad_f = false, bad_r = false;
            if (!skip) mask_decide<W>(b0, b1, g & A.kmask, A.cand_n[it.cand0 + c], A, bad_f, bad_r);
            // the bit-sliced pass left a 1 for every row `excl` flags: the patch rows get their own verdict either way
            if (bad_f) atomicOr(&not_f[(size_t)oc * nw + (size_t)(r >> 6)], bit);
            else atomicAnd(&not_f[(size_t)oc * nw + (size_t)(r >> 6)], ~bit);
            if (bad_r) atomicOr(&not_r[(size_t)oc * nw + (size_t)(r >> 6)], bit);
            else atomicAnd(&not_r[(size_t)oc * nw + (size_t)(r >> 6)], ~bit);
        }
    }
}

// fix-ups of single bits (the host's verdict on IUPAC rows): thread = one (mask, row) assignment
__global__ __launch_bounds__(kBlock) void mask_set_kernel(long long n, const int32_t *__restrict__ cand, const int32_t *__restrict__ row,
                                                          const uint8_t *__restrict__ which, const uint8_t *__restrict__ value, size_t nw,
                                                          unsigned long long *__restrict__ not_f, unsigned long long *__restrict__ not_r) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    unsigned long long *m = (which[i] ? not_r : not_f) + (size_t)cand[i] * nw + (size_t)(row[i] >> 6);
    const unsigned long long bit = 1ull << (row[i] & 63);
    if (value[i]) atomicOr(m, bit);
    else atomicAnd(m, ~bit);
}

// popcount(not_f[i] | not_r[j]) over the resident masks: one wave per pair
__global__ __launch_bounds__(kBlock) void mask_pair_kernel(const unsigned long long *__restrict__ f, const unsigned long long *__restrict__ r2,
                                                           int n_words, size_t stride, long long n_pairs, const int32_t *__restrict__ pairs,
                                                           int32_t *__restrict__ out) {
    const long long p = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (p >= n_pairs) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long *A = f + (size_t)pairs[2 * p] * stride, *B = r2 + (size_t)pairs[2 * p + 1] * stride;
    int cnt = 0;
    for (int w = lane; w < n_words; w += 64) cnt += __popcll(A[w] | B[w]);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) cnt += __shfl_xor(cnt, s);
    if (lane == 0) out[p] = cnt;
}

__global__ __launch_bounds__(kBlock) void zero_kernel(unsigned long long *__restrict__ p, size_t n) {
    const size_t i0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * 4;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (i0 + j < n) p[i0 + j] = 0ull;
}

typedef void (*EvalBitsFn)(const EvalBitsArgs);
typedef void (*EvalChainFn)(const EvalChainArgs);

typedef void (*EvalFn)(const EvalArgs);
struct EvalVariant { const char *name; EvalFn fn[3]; };     // fn[VMODE]
#define EVAL_VARIANT(name, COUNT, PREFETCH, FORM)                                                         \
    { name, { eval_kernel<kEvalCC, 0, COUNT, PREFETCH, FORM>, eval_kernel<kEvalCC, 1, COUNT, PREFETCH, FORM>, \
              eval_kernel<kEvalCC, 2, COUNT, PREFETCH, FORM> } }
// variant 0 is the default; the others exist to be measured (tools/variant_bench.py, MP_EVAL_VARIANT)
const EvalVariant kEvalVariants[] = {
    EVAL_VARIANT("ballot+prefetch/onehot", 1, true, 2),      // default: fastest measured (profiles/r01_variants.txt)
    EVAL_VARIANT("ballot+prefetch/bfi-vgpr", 1, true, 1),
    EVAL_VARIANT("ballot+prefetch/bfi-sgpr", 1, true, 0),
    EVAL_VARIANT("lane-acc+prefetch/onehot", 0, true, 2),
    EVAL_VARIANT("ballot/onehot", 1, false, 2),
};
constexpr int kNumEvalVariants = (int)(sizeof(kEvalVariants) / sizeof(kEvalVariants[0]));



// (Re)builds the patch planes after mp_build_windows / mp_set_extra_rows changed the lists they mirror.
int ensure_patch_planes(mp_ctx *c) {
    if (!c->pp_dirty) return MP_OK;
    dev_free(c, &c->pplanes, c->pp_words); dev_free(c, &c->pvalid, c->pv_words); dev_free(c, &c->pwin, (size_t)c->n_win);
    c->pp_words = c->pv_words = 0;
    c->max_npw = 0;
    const size_t W = (size_t)c->n_win;
    std::vector<PatchWin> pw(W);
    std::vector<PatchRun> runs;
    size_t poff = 0, voff = 0;
    for (size_t w = 0; w < W; w++) {
        const int n = (c->h_patch_off[w + 1] - c->h_patch_off[w]) + (c->h_extra_off[w + 1] - c->h_extra_off[w]);
        const int npw = ((n + 31) / 32 + 7) / 8 * 8;
        if (poff + (size_t)c->k * 4 * npw > 0x7fffffffu) return fail(c, MP_ERR_NOMEM, "patch planes too large");
        pw[w] = PatchWin{(int32_t)poff, (int32_t)voff, npw};
        poff += (size_t)c->k * 4 * npw;
        voff += (size_t)npw;
        c->max_npw = std::max(c->max_npw, npw);
        for (int r0 = 0; r0 < npw * 32; r0 += kPatchRun) runs.push_back(PatchRun{(int32_t)w, r0});
    }
    int rc;
    if ((rc = dev_alloc(c, &c->pwin, W))) return rc;
    HIPCK(c, hipMemcpyAsync(c->pwin, pw.data(), sizeof(PatchWin) * W, hipMemcpyHostToDevice, c->stream));
    PatchRun *d_runs = nullptr;
    if (poff) {
        if ((rc = dev_alloc(c, &c->pplanes, poff))) return rc;
        if ((rc = dev_alloc(c, &c->pvalid, voff))) return rc;
        c->pp_words = poff; c->pv_words = voff;
        if ((rc = dev_alloc(c, &d_runs, runs.size()))) return rc;
        HIPCK(c, hipMemcpyAsync(d_runs, runs.data(), sizeof(PatchRun) * runs.size(), hipMemcpyHostToDevice, c->stream));
        if (c->wide)
            hipLaunchKernelGGL(patch_planes_kernel<uint64_t>, dim3((unsigned)runs.size()), dim3(kBlock), 0, c->stream,
                               c->n_patch ? c->patch_off : (const int32_t *)nullptr, reinterpret_cast<const uint64_t *>(c->patch_words),
                               c->n_extra ? c->extra_off : (const int32_t *)nullptr, reinterpret_cast<const uint64_t *>(c->extra_words), c->pwin, d_runs,
                               c->k, c->v, c->pplanes, c->pvalid);
        else
            hipLaunchKernelGGL(patch_planes_kernel<uint32_t>, dim3((unsigned)runs.size()), dim3(kBlock), 0, c->stream,
                               c->n_patch ? c->patch_off : (const int32_t *)nullptr, (const uint32_t *)c->patch_words,
                               c->n_extra ? c->extra_off : (const int32_t *)nullptr, (const uint32_t *)c->extra_words, c->pwin, d_runs, c->k, c->v,
                               c->pplanes, c->pvalid);
        HIPCK(c, hipGetLastError());
    }
    HIPCK(c, hipStreamSynchronize(c->stream));            // pw and runs are host temporaries
    dev_free(c, &d_runs, runs.size());
    c->pp_dirty = false;
    return MP_OK;
}

// The plain-slice planes of the patch-list rows (same sizes and offsets as the patch planes): built when the sliding evaluation first
// needs them after the window lists changed.
int ensure_plain_planes(mp_ctx *c) {
    int rc = ensure_patch_planes(c);
    if (rc) return rc;
    if (!c->qp_dirty) return MP_OK;
    dev_free(c, &c->qplanes, c->pp_words); dev_free(c, &c->qvalid, c->pv_words);
    if (c->pp_words == 0) { c->qp_dirty = false; return MP_OK; }
    if ((rc = dev_alloc(c, &c->qplanes, c->pp_words))) return rc;
    if ((rc = dev_alloc(c, &c->qvalid, c->pv_words))) return rc;
    std::vector<PatchRun> runs;
    for (size_t w = 0; w < (size_t)c->n_win; w++) {
        const int n = (c->h_patch_off[w + 1] - c->h_patch_off[w]) + (c->h_extra_off[w + 1] - c->h_extra_off[w]);
        const int npw = ((n + 31) / 32 + 7) / 8 * 8;
        for (int r0 = 0; r0 < npw * 32; r0 += kPatchRun) runs.push_back(PatchRun{(int32_t)w, r0});
    }
    PatchRun *d_runs = nullptr;
    if ((rc = dev_alloc(c, &d_runs, runs.size()))) return rc;
    HIPCK(c, hipMemcpyAsync(d_runs, runs.data(), sizeof(PatchRun) * runs.size(), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(plain_planes_kernel, dim3((unsigned)runs.size()), dim3(kBlock), 0, c->stream,
                       c->n_patch ? c->patch_off : (const int32_t *)nullptr, (const int32_t *)c->patch_rows, c->pwin, d_runs, c->planes, c->n_pad,
                       c->n_chunks, c->p0, c->k, c->qplanes, c->qvalid);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipStreamSynchronize(c->stream));
    dev_free(c, &d_runs, runs.size());
    c->qp_dirty = false;
    return MP_OK;
}

// workgroups of a bit-sliced launch: n_items items x the row slices of nw 64-bit words at GW 32-bit words per thread (bitslice.hpp)
BlockMap make_block_map(int nw, int GW, int n_items, unsigned &grid) {
    BlockMap m;
    m.ny = std::max(1, (2 * nw / GW + kBlock - 1) / kBlock);
    m.ny_pad = m.ny > 4 ? (m.ny + 7) / 8 * 8 : (m.ny > 2 ? 4 : m.ny);
    m.n_items = n_items;
    const int bands = m.ny_pad >= 8 ? 1 : 8 / m.ny_pad;
    m.per_band = (n_items + bands - 1) / bands;
    grid = m.ny_pad >= 8 ? (unsigned)((size_t)n_items * m.ny_pad) : (unsigned)(8 * (size_t)m.per_band);
    return m;
}

// patch units of a launch over n_items items with GW words per thread and unit_threads threads per unit
PatchArgs patch_args(const mp_ctx *c, int GW, int n_items, int unit_threads) {
    PatchArgs pa{c->pplanes, c->pvalid, c->pwin, 0, 0, nullptr, nullptr, 0};
    static const bool skip = getenv("MP_EXPERIMENT_SKIP_PATCH") != nullptr;      // TIMING EXPERIMENTS ONLY (tools/): the counts come out wrong
    if (c->max_npw > 0 && n_items > 0 && !skip) {
        pa.per_item = (c->max_npw + unit_threads * GW - 1) / (unit_threads * GW);
        const long long units = (long long)pa.per_item * n_items, per_block = kBlock / unit_threads;
        pa.n_blocks = (int)(((units + per_block - 1) / per_block + 7) / 8 * 8);
    }
    return pa;
}

}  // namespace

// [r6] The chain items of eval_chain_x_kernel (evalx.hpp) for a staged candidate set of primers of 32..63 bases or of v = 4, 5: every
// maximal run of at most 8 consecutive candidates of a window that is nested in one direction (a refinement chain read either way) is
// one item — a single candidate is a run of one: its first pass alone is ~80 times cheaper than the row-per-lane comparison.  Built
// BESIDE the row-per-lane arrays (the coverage masks and MP_EVAL_MODE=rows still use those); not built when a candidate holds an empty
// symbol (it matches nothing: the row-per-lane form handles that).
static int upload_x(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes) {
    c->x_n = 0;
    if (c->v > 5 || c->k > 63 || getenv("MP_EVAL_NO_X")) return MP_OK;
    const int k = c->k;
    auto sym_at = [&](int ci, int p) { return (uint32_t)(codes[(size_t)ci * k + p] & 15u); };
    for (int ci = 0; ci < n_cand; ci++)
        for (int p = 0; p < k; p++)
            if (!sym_at(ci, p)) return MP_OK;
    std::vector<ChainItemX> items;
    std::vector<uint32_t> events;
    std::vector<int32_t> co;
    std::vector<int> order;
    for (int i = 0; i < n_cand;) {
        const int w = cw[i];
        int j = i;
        while (j < n_cand && cw[j] == w) j++;
        for (int b = i; b < j;) {
            int e = b + 1, dir = 3;
            while (e < j && e - b < kEvalCC) {
                int rel = 3;
                for (int p = 0; p < k; p++) {
                    const uint32_t x = sym_at(e - 1, p), y = sym_at(e, p);
                    if (x & ~y) rel &= ~1;           // not "previous within next"
                    if (y & ~x) rel &= ~2;           // not "next within previous"
                }
                if (!(dir & rel)) break;
                dir &= rel; e++;
            }
            order.clear();
            if ((dir & 2) == 0) for (int ci = e - 1; ci >= b; ci--) order.push_back(ci);      // ascending run: most degenerate member last
            else for (int ci = b; ci < e; ci++) order.push_back(ci);
            ChainItemX ch{};
            ch.win = w; ch.cand0 = (int32_t)co.size(); ch.n_steps = (int32_t)order.size(); ch.ev0 = (int32_t)events.size();
            for (int p = 0; p < k; p++) {
                const uint32_t sy = sym_at(order[0], p);
                ch.sym[p >> 3] |= sy << (4 * (p & 7));
                const int nb = __builtin_popcount(sy);
                (nb == 1 ? ch.pos1 : (nb == 2 ? ch.pos2 : ch.pos4)) |= 1ull << p;
            }
            for (size_t t = 1; t < order.size(); t++)
                for (int p = 0; p < k; p++) {
                    const uint32_t lost = sym_at(order[t - 1], p) & ~sym_at(order[t], p);
                    for (uint32_t bit = 1; bit < 16; bit <<= 1)
                        if (lost & bit) events.push_back((uint32_t)p | (bit << 8) | ((uint32_t)t << 16));
                }
            ch.n_ev = (int32_t)events.size() - ch.ev0;
            items.push_back(ch);
            for (int t = 0; t < kEvalCC; t++) co.push_back(t < (int)order.size() ? order[(size_t)t] : -1);
            b = e;
        }
        i = j;
    }
    if (items.empty()) return MP_OK;
    int rc;
    ChainItemX *d_items = nullptr;
    if ((rc = dev_alloc(c, &d_items, items.size()))) return rc;
    c->x_items = d_items;
    c->x_n = (int)items.size(); c->x_n_events = (int)events.size();
    if ((rc = dev_alloc(c, &c->x_events, events.size() + 1))) return rc;
    if ((rc = dev_alloc(c, &c->x_cand_out, co.size()))) return rc;
    HIPCK(c, hipMemcpyAsync(d_items, items.data(), sizeof(ChainItemX) * items.size(), hipMemcpyHostToDevice, c->stream));
    if (!events.empty()) HIPCK(c, hipMemcpyAsync(c->x_events, events.data(), sizeof(uint32_t) * events.size(), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(c->x_cand_out, co.data(), sizeof(int32_t) * co.size(), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

typedef void (*EvalXFn)(const EvalXArgs);
static int launch_eval_x(mp_ctx *c, unsigned long long *out) {
    { int rc = ensure_patch_planes(c); if (rc) return rc; }
    const int nw = c->n_pad / 64, nw32 = 2 * nw;
    // words per thread x positions in flight by depth (as eval_chain_kernel chooses them); five and six counter levels: at most 4 words
#define X_ROW(LV) {eval_chain_x_kernel<LV, 1, 6>, eval_chain_x_kernel<LV, 2, 6>, eval_chain_x_kernel<LV, 4, 3>, eval_chain_x_kernel<LV, (LV <= 4 ? 8 : 4), (LV <= 4 ? 2 : 3)>}
    static const EvalXFn fn[6][4] = {X_ROW(1), X_ROW(2), X_ROW(3), X_ROW(4), X_ROW(5), X_ROW(6)};
#undef X_ROW
    int shape = nw32 >= 4 * kBlock ? 3 : (nw32 >= 2 * kBlock ? 2 : (nw32 >= kBlock ? 1 : 0));
    if (const char *e = getenv("MP_EVAL_X_SHAPE")) { const int sh = atoi(e); if (sh >= 0 && sh < 4) shape = sh; }
    static const int gw_of[4] = {1, 2, 4, 8};
    const int GW = shape == 3 && c->v >= 4 ? 4 : gw_of[shape];
    unsigned grid;
    const BlockMap bm = make_block_map(nw, GW, c->x_n, grid);
    EvalXArgs xa{c->cols, c->excl, nw, c->p0, c->k, c->v, reinterpret_cast<const ChainItemX *>(c->x_items), c->x_events, c->x_cand_out, c->sF, c->sR, out, bm,
                 patch_args(c, GW, c->x_n, 64)};
    hipLaunchKernelGGL(fn[c->v][shape], dim3(grid + (unsigned)xa.patch.n_blocks), dim3(kBlock), 0, c->stream, xa);
    return MP_OK;
}

extern "C" {

// Primers of 32..63 bases: 8 consecutive candidates of a window per item, 64-bit candidate words, the row-per-lane kernels only.
static int upload_wide(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint64_t sF, uint64_t sR) {
    const int k = c->k;
    const uint64_t kmask = kmask_of<uint64_t>(k);
    std::vector<EvalItem> items;
    std::vector<CandN<uint64_t>> cn;
    std::vector<int32_t> co;
    for (int i = 0; i < n_cand;) {
        const int w = cw[i];
        if (w < 0 || w >= c->n_win || (i && w < cw[i - 1])) return fail(c, MP_ERR_ARG, "candidate windows must be ascending and in range");
        int j = i;
        while (j < n_cand && cw[j] == w) j++;
        for (int b = i; b < j; b += kEvalCC) {
            items.push_back(EvalItem{w, (int32_t)cn.size()});
            for (int t = 0; t < kEvalCC; t++) {
                const int ci = b + t;
                if (ci >= j) { cn.push_back(CandN<uint64_t>{kmask, kmask, kmask, kmask}); co.push_back(-1); continue; }      // an unused slot matches nothing, reports nowhere
                CandN<uint64_t> q{0, 0, 0, 0};
                for (int p = 0; p < k; p++) {
                    const uint32_t m = codes[(size_t)ci * k + p] & 15u;
                    if (!(m & 1)) q.x |= 1ull << p;
                    if (!(m & 2)) q.y |= 1ull << p;
                    if (!(m & 4)) q.z |= 1ull << p;
                    if (!(m & 8)) q.w |= 1ull << p;
                }
                cn.push_back(q);
                co.push_back(ci);
            }
        }
        i = j;
    }
    c->n_cand = n_cand; c->sF = sF; c->sR = sR;
    c->n_items = (int)items.size();
    c->n_padded = (int)cn.size();
    if (c->n_items == 0) return MP_OK;
    int rc;
    if ((rc = dev_alloc(c, &c->items, items.size()))) return rc;
    if ((rc = dev_alloc(c, &c->cand_n, 2 * cn.size()))) return rc;                   // two uint4 per 64-bit candidate record
    if ((rc = dev_alloc(c, &c->cand_out, co.size()))) return rc;
    HIPCK(c, hipMemcpy(c->items, items.data(), sizeof(EvalItem) * items.size(), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(c->cand_n, cn.data(), sizeof(CandN<uint64_t>) * cn.size(), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(c->cand_out, co.data(), sizeof(int32_t) * co.size(), hipMemcpyHostToDevice));
    return upload_x(c, n_cand, cw, codes);           // [r6] the bit-sliced chain items beside the row-per-lane arrays
}

int mp_eval_upload(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint64_t sF64, uint64_t sR64) {
    if (!c) return MP_ERR_ARG;
    if (!c->excl) return fail(c, MP_ERR_ARG, "no windows built");
    if (n_cand < 0 || (n_cand && (!cw || !codes))) return fail(c, MP_ERR_ARG, "bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    free_eval(c);
    if (c->wide) return upload_wide(c, n_cand, cw, codes, sF64, sR64);
    const uint32_t sF = (uint32_t)sF64, sR = (uint32_t)sR64;      // k <= 31: positions above k never mismatch
    const int k = c->k;
    const uint32_t kmask = (1u << k) - 1u;
    std::vector<EvalItem> items;
    std::vector<uint4> cn;
    std::vector<int32_t> co, table_ids;
    std::vector<uint32_t> symT, diffm, events;
    std::vector<ChainItem> chains;
    // grouping policy (MP_EVAL_GROUP): "plain" = 8 consecutive candidates per item, symbol-table kernel; "nested" =
    // always split into nested runs; default = whichever the per-window cost model (VALU per 32 sequences) prefers
    const char *genv = getenv("MP_EVAL_GROUP");
    const int policy = genv ? (!strcmp(genv, "plain") ? 1 : (!strcmp(genv, "nested") ? 2 : 0)) : 0;
    struct Run { int b, e; bool asc; int n_ev; };
    constexpr int kMaxRun = 64;                  // members of one chain item (8 groups of output slots)
    auto sym_at = [&](int ci, int p) { return (uint32_t)(codes[(size_t)ci * k + p] & 15u); };
    auto diff_of = [&](int b, int e) {
        uint32_t dm = 0;
        for (int p = 0; p < k; p++)
            for (int ci = b + 1; ci < e; ci++)
                if (sym_at(ci, p) != sym_at(b, p)) dm |= 1u << p;
        return dm;
    };
    // one item = up to 8 slots; `order` lists its candidates (for a nested run: most degenerate first)
    // 8 output slots (one EvalItem) from `order[g0 .. g0+8)`; `order` lists candidates (for a nested run: most degenerate first)
    auto emit_slots = [&](int w, const std::vector<int> &order, int g0) {
        const int n = std::min(kEvalCC, (int)order.size() - g0);
        const int item = (int)items.size();
        items.push_back(EvalItem{w, (int32_t)cn.size()});
        symT.resize(items.size() * 32, 0u);
        uint32_t dm = 0;
        for (int p = 0; p < k; p++)
            for (int t = 1; t < n; t++)
                if (sym_at(order[g0 + t], p) != sym_at(order[g0], p)) dm |= 1u << p;
        diffm.push_back(dm);
        for (int t = 0; t < kEvalCC; t++) {
            const int ci = order[g0 + (t < n ? t : n - 1)];      // an unused slot repeats the last candidate, reports nowhere
            uint32_t nA = 0, nC = 0, nG = 0, nT = 0;
            for (int p = 0; p < k; p++) {
                const uint32_t m = sym_at(ci, p);
                if (!(m & 1)) nA |= 1u << p;
                if (!(m & 2)) nC |= 1u << p;
                if (!(m & 4)) nG |= 1u << p;
                if (!(m & 8)) nT |= 1u << p;
                symT[(size_t)item * 32 + (size_t)p] |= m << (4 * t);
            }
            if (t < n) { cn.push_back(uint4{nA, nC, nG, nT}); co.push_back(ci); }
            else { cn.push_back(uint4{kmask, kmask, kmask, kmask}); co.push_back(-1); }
        }
        return item;
    };
    // a nested run of any length up to kMaxRun is ONE chain item over consecutive 8-slot groups: the counters carry over
    auto emit = [&](int w, const std::vector<int> &order, bool nested) {
        const int n = (int)order.size();
        if (nested) {
            ChainItem ch{w, (int32_t)cn.size(), n, (int32_t)events.size(), 0, {0u, 0u, 0u, 0u}, 0u, 0u, 0u};
            for (int p = 0; p < k; p++) {
                const uint32_t sy = sym_at(order[0], p);
                ch.sym[p >> 3] |= sy << (4 * (p & 7));
                const int nb = __builtin_popcount(sy);
                (nb == 1 ? ch.pos1 : (nb == 2 ? ch.pos2 : ch.pos4)) |= 1u << p;
            }
            for (int t = 1; t < n; t++)
                for (int p = 0; p < k; p++) {
                    const uint32_t lost = sym_at(order[t - 1], p) & ~sym_at(order[t], p);
                    for (uint32_t bit = 1; bit < 16; bit <<= 1)      // one event per lost base: its column plane IS the increment
                        if (lost & bit) events.push_back((uint32_t)p | (bit << 8) | ((uint32_t)t << 16));
                }
            ch.n_ev = (int32_t)events.size() - ch.ev0;
            chains.push_back(ch);
            for (int g0 = 0; g0 < n; g0 += kEvalCC) emit_slots(w, order, g0);
        } else {
            for (int g0 = 0; g0 < n; g0 += kEvalCC) table_ids.push_back(emit_slots(w, order, g0));
        }
    };
    int i = 0;
    std::vector<Run> runs;
    std::vector<int> order;
    while (i < n_cand) {
        int w = cw[i];
        if (w < 0 || w >= c->n_win || (i && w < cw[i - 1])) return fail(c, MP_ERR_ARG, "candidate windows must be ascending and in range");
        int j = i;
        while (j < n_cand && cw[j] == w) j++;
        // maximal runs of consecutive candidates that are nested in one direction (a refinement chain, either way round)
        runs.clear();
        long cost_nested = 0, cost_plain = 0;
        bool has_empty = false;                     // a candidate with an empty symbol (matches nothing): symbol-table kernel only
        for (int b = i; b < j;) {
            int e = b + 1, dir = 3, n_ev = 0;
            while (e < j && e - b < kMaxRun) {
                int rel = 3, changed = 0;
                for (int p = 0; p < k; p++) {
                    const uint32_t x = sym_at(e - 1, p), y = sym_at(e, p);
                    if (x & ~y) rel &= ~1;           // not "previous within next"
                    if (y & ~x) rel &= ~2;           // not "next within previous"
                    changed += x != y;
                }
                if (!(dir & rel)) break;
                dir &= rel; n_ev += changed; e++;
            }
            for (int ci = b; ci < e; ci++)
                for (int p = 0; p < k; p++)
                    if (!sym_at(ci, p)) has_empty = true;
            runs.push_back(Run{b, e, (dir & 2) == 0, n_ev});
            cost_nested += 4L * k + 4L * n_ev + 6L * (e - b) + 40L * ((e - b + kEvalCC - 1) / kEvalCC);
            b = e;
        }
        for (int b = i; b < j; b += kEvalCC) {
            const int nd = __builtin_popcount(diff_of(b, std::min(j, b + kEvalCC)));
            cost_plain += 45L * nd + 8L * (k - nd) + 136 + 40;
        }
        const bool use_nested = !has_empty && (policy == 2 || (policy == 0 && cost_nested <= cost_plain));
        if (use_nested) {
            for (const Run &r : runs) {
                order.clear();
                if (r.asc) for (int ci = r.e - 1; ci >= r.b; ci--) order.push_back(ci);
                else for (int ci = r.b; ci < r.e; ci++) order.push_back(ci);
                emit(w, order, true);
            }
        } else {
            for (int b = i; b < j; b += kEvalCC) {
                order.clear();
                for (int ci = b; ci < j && ci < b + kEvalCC; ci++) order.push_back(ci);
                emit(w, order, false);
            }
        }
        i = j;
    }
    c->n_cand = n_cand; c->sF = sF; c->sR = sR;
    c->n_items = (int)items.size();
    c->n_padded = (int)cn.size();
    if (c->n_items == 0) return MP_OK;
    int rc;
    if ((rc = dev_alloc(c, &c->items, items.size()))) return rc;
    if ((rc = dev_alloc(c, &c->cand_n, cn.size()))) return rc;
    if ((rc = dev_alloc(c, &c->cand_out, co.size()))) return rc;
    if ((rc = dev_alloc(c, &c->cand_symT, symT.size()))) return rc;
    HIPCK(c, hipMemcpy(c->cand_symT, symT.data(), sizeof(uint32_t) * symT.size(), hipMemcpyHostToDevice));
    if ((rc = dev_alloc(c, &c->cand_diff, diffm.size()))) return rc;
    HIPCK(c, hipMemcpy(c->cand_diff, diffm.data(), sizeof(uint32_t) * diffm.size(), hipMemcpyHostToDevice));
    c->n_chain = (int)chains.size(); c->n_table = (int)table_ids.size(); c->n_events = (int)events.size();
    c->max_steps = 0;
    for (const ChainItem &ch : chains) c->max_steps = std::max(c->max_steps, (int)ch.n_steps);
    c->h_chains = chains;
    c->h_events = events;
    c->h_cand_out = co;
    // Which kernel walks the chains.  eval_prog_kernel (evalprog.hip: host-written fetch programs, buffer loads, event planes parked
    // in LDS) is faster from about 400 000 rows up, where the planes no longer come out of L2 (shape 11: 1.122 vs 1.193 ms / 10 steps at
    // 524 288 rows, 0.217-0.223 vs 0.240-0.242 ms at 1 048 576; slower below: 0.0549 vs 0.0524 ms at 262 144 —
    // profiles/r03_prog_keep.txt); eval_chain_kernel otherwise.  MP_EVAL_PROG=1 / 0 forces one of them, MP_EVAL_CHAIN the shape.
    // The sliding kernel (evalslide.hip) takes the chain items it can (all of them in a refinement run); what it leaves out stays with
    // the first-pass kernel below.  MP_EVAL_SLIDE=0 / 1 forbids / forces it.
    if (c->n_chain && (rc = upload_eval_slide(c, chains, events, co))) return rc;
    c->prog_shape = -1;
    if (c->n_chain && c->max_steps <= kEvalCC && c->slide_items == 0) {
        const char *pe = getenv("MP_EVAL_PROG");
        if (pe ? atoi(pe) == 1 : c->n_pad >= 393216) {
            c->prog_shape = pe ? 7 : 11;
            if (const char *e = getenv("MP_EVAL_CHAIN")) { const int sh = atoi(e); if (pe && sh >= 0 && sh < kProgShapes) c->prog_shape = sh; }
        }
    }
    if (c->prog_shape >= 0) {
        std::vector<uint32_t> prog;                // the programs carry the LDS slots of the shape that will run them
        build_eval_programs(chains, events, co, k, sF, sR, kProgKeep[c->prog_shape], prog);
        if ((rc = dev_alloc(c, &c->chain_prog, prog.size()))) return rc;
        c->chain_prog_n = prog.size();
        HIPCK(c, hipMemcpy(c->chain_prog, prog.data(), sizeof(uint32_t) * prog.size(), hipMemcpyHostToDevice));
    }
    if (c->n_chain) {
        if ((rc = dev_alloc(c, &c->chain_items, chains.size()))) return rc;
        HIPCK(c, hipMemcpy(c->chain_items, chains.data(), sizeof(ChainItem) * chains.size(), hipMemcpyHostToDevice));
        if ((rc = dev_alloc(c, &c->chain_events, events.size() + 1))) return rc;      // + 1: never a null pointer
        if (!events.empty()) HIPCK(c, hipMemcpy(c->chain_events, events.data(), sizeof(uint32_t) * events.size(), hipMemcpyHostToDevice));
    }
    if (c->n_table) {
        if ((rc = dev_alloc(c, &c->table_ids, table_ids.size()))) return rc;
        HIPCK(c, hipMemcpy(c->table_ids, table_ids.data(), sizeof(int32_t) * table_ids.size(), hipMemcpyHostToDevice));
    }
    HIPCK(c, hipMemcpy(c->items, items.data(), sizeof(EvalItem) * items.size(), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(c->cand_n, cn.data(), sizeof(uint4) * cn.size(), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(c->cand_out, co.data(), sizeof(int32_t) * co.size(), hipMemcpyHostToDevice));
    if (c->v > 3) return upload_x(c, n_cand, cw, codes);     // [r6] v = 4, 5: the four-level kernels above do not count that far
    return MP_OK;
}

// zero_out: the launch clears device_out itself first (mp_eval_launch).  Otherwise the caller vouches for a zeroed device_out
// (mp_eval_launch_rotating) and `device_clear`, if given, is zeroed by the first evaluation kernel of the step inside its own grid.
static int eval_launch_impl(mp_ctx *c, int64_t *device_out, int64_t *device_clear, bool zero_out) {
    if (!c) return MP_ERR_ARG;
    if (!c->excl) return fail(c, MP_ERR_ARG, "no windows built");
    if (!device_out) return fail(c, MP_ERR_ARG, "null output");
    if (device_clear == device_out) return fail(c, MP_ERR_ARG, "the block to clear is the block to fill");
    HIPCK(c, hipSetDevice(c->dev));
    if (c->n_cand == 0) return MP_OK;
    const size_t n_counters = 3 * (size_t)c->n_cand;
    auto zero = [&](int64_t *p) {
        hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((n_counters + 4 * kBlock - 1) / (4 * kBlock))), dim3(kBlock), 0, c->stream,
                           reinterpret_cast<unsigned long long *>(p), n_counters);
    };
    // the counters are summed with atomics: they start at zero.  A launch of our own: the runtime's fill kernel takes 6 us per call
    // at this size (profiles/r02_pipeline_kernels.txt), a fifth of the evaluation itself
    if (zero_out) zero(device_out);
    // [r6, advisor] a rotating launch clears 3 x n_cand counters of the block it is handed — the size of ITS staged set.  When that block comes
    // back as device_out under a larger set, its tail was never cleared: refuse instead of adding to stale counts.
    if (!zero_out && device_out == c->rot_block && n_counters > c->rot_cleared)
        return fail(c, MP_ERR_ARG, "mp_eval_launch_rotating: this block was cleared for %zu counters by the launch before, the staged set needs %zu",
                    c->rot_cleared, n_counters);
    if (!zero_out) { c->rot_block = device_clear; c->rot_cleared = device_clear ? n_counters : 0; }
    // the rotating form's side job goes to the first kernel of the step that can take it (a kernel that cannot leaves it pending)
    unsigned long long *pending_clear = reinterpret_cast<unsigned long long *>(device_clear);
    auto take_clear = [&]() { unsigned long long *p = pending_clear; pending_clear = nullptr; return p; };
    const uint32_t n_clear = (uint32_t)n_counters;
    // enough blocks to fill 256 CUs several times over, each with at least 1024 sequences
    int max_split = (c->n_pad + 1023) / 1024;
    int want = (4096 + c->n_items - 1) / c->n_items;
    int split = std::max(1, std::min(max_split, want));
    int rows = ((c->n_pad + split - 1) / split + 1023) / 1024 * 1024;
    split = (c->n_pad + rows - 1) / rows;
    // HIP-event timing of the launch (mp_eval_timing).  An event pair idles the stream for ~6 us, so
    // MP_EVAL_TIMING_EVERY=n times every n-th launch only, counted from the last timing reset (0: none); default: all.
    int every = 1;
    if (const char *e = getenv("MP_EVAL_TIMING_EVERY")) every = std::max(0, atoi(e));
    const bool timed = every > 0 && (c->launch_seq++ % (unsigned)every) == 0;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (timed) {
        if (!c->ev_free.empty()) { ev = c->ev_free.back(); c->ev_free.pop_back(); }
        else { HIPCK(c, hipEventCreate(&ev.first)); HIPCK(c, hipEventCreate(&ev.second)); }
        HIPCK(c, hipEventRecord(ev.first, c->stream));
    }
    const char *mode_env = getenv("MP_EVAL_MODE");
    const bool bits = c->v <= 3 && !c->wide && !(mode_env && !strcmp(mode_env, "rows"));      // (the bit-sliced kernels hold 32 positions per item)
    const bool xbits = !bits && c->x_n > 0 && !(mode_env && !strcmp(mode_env, "rows"));       // [r6] 32..63 positions / v = 4, 5: eval_chain_x_kernel
    const int vmode = c->v == 0 ? 0 : (c->v == 1 ? 1 : 2);     // predicate specialisation of the row-per-lane code
    if (xbits) {
        const int rc = launch_eval_x(c, (unsigned long long *)device_out);
        if (rc) return rc;
    } else if (bits) {
        // bit-sliced pass over the column planes and over the windows' patch planes
        // MP_EVAL_BITS: 0 (default) = nested-chain kernel on the nested items + symbol-table kernel (with the shared-
        // position shortcut) on the others; 1 = symbol-table kernel on every item, no shortcut; 2 = the same with it
        int shape = 0;
        if (const char *e = getenv("MP_EVAL_BITS")) { shape = atoi(e); if (shape < 0 || shape > 2) shape = 0; }
        const int nw = c->n_pad / 64;
        auto block_map = [&](int GW, int n_items, unsigned &grid) { return make_block_map(nw, GW, n_items, grid); };
        { int rc = ensure_patch_planes(c); if (rc) return rc; }
        static const EvalBitsFn tfn[4][2] = {{eval_bits_kernel<1, 2, true, 1>, eval_bits_kernel<1, 2, false, 1>},
                                             {eval_bits_kernel<2, 2, true, 1>, eval_bits_kernel<2, 2, false, 1>},
                                             {eval_bits_kernel<3, 2, true, 1>, eval_bits_kernel<3, 2, false, 1>},
                                             {eval_bits_kernel<4, 2, true, 1>, eval_bits_kernel<4, 2, false, 1>}};
        if (shape == 0 && c->n_chain) {
            // words per thread x positions in flight: the more sequences a block covers, the further its fixed costs
            // (24 popcount totals, item and event fetches) are spread — 8 x 4 from 32768 sequences up (half a block of
            // threads at that size), 4 x 3 from 16384, else 2 x 6 / 1 x 6.  MP_EVAL_CHAIN overrides (tools/variant_bench.py).
            const int nw32 = 2 * nw;
            int cshape = nw32 >= 4 * kBlock ? 7 : (nw32 >= 2 * kBlock ? 3 : (nw32 >= kBlock ? 0 : 5));
            const bool use_prog = c->chain_prog && c->prog_shape >= 0;
            if (use_prog) cshape = c->prog_shape;
            else if (const char *e = getenv("MP_EVAL_CHAIN")) { cshape = atoi(e); if (cshape < 0 || cshape > 8) cshape = 0; }
            // (plane rows and patch planes are padded to multiples of 8 words: 8 words per thread is the widest shape of
            // eval_chain_kernel; shapes 9-12 — 16 words per thread, event planes parked in LDS — exist in the program-driven kernel only)
            const int *cgw = kProgWords;
#define CHAIN_ROW(LV) {eval_chain_kernel<LV, 2, 6>, eval_chain_kernel<LV, 2, 3>, eval_chain_kernel<LV, 2, 9>, eval_chain_kernel<LV, 4, 3>, \
                       eval_chain_kernel<LV, 4, 6>, eval_chain_kernel<LV, 1, 6>, eval_chain_kernel<LV, 8, 2>, eval_chain_kernel<LV, 8, 4>, \
                       eval_chain_kernel<LV, 8, 1>}
            static const EvalChainFn cfn[4][9] = {CHAIN_ROW(1), CHAIN_ROW(2), CHAIN_ROW(3), CHAIN_ROW(4)};
#undef CHAIN_ROW
#define CHAIN_ROW(LV) {eval_chain_long_kernel<LV, 2, 6>, eval_chain_long_kernel<LV, 2, 3>, eval_chain_long_kernel<LV, 2, 9>, \
                       eval_chain_long_kernel<LV, 4, 3>, eval_chain_long_kernel<LV, 4, 6>, eval_chain_long_kernel<LV, 1, 6>, \
                       eval_chain_long_kernel<LV, 8, 2>, eval_chain_long_kernel<LV, 8, 4>, eval_chain_long_kernel<LV, 8, 1>}
            static const EvalChainFn lfn[4][9] = {CHAIN_ROW(1), CHAIN_ROW(2), CHAIN_ROW(3), CHAIN_ROW(4)};
#undef CHAIN_ROW
            unsigned grid;
            const BlockMap bm = block_map(cgw[cshape], c->n_chain, grid);
            EvalChainArgs ca{c->cols, c->excl, nw, c->p0, c->k, c->v, c->chain_items, c->chain_events, c->cand_out, (uint32_t)c->sF, (uint32_t)c->sR,
                             (unsigned long long *)device_out, bm, patch_args(c, cgw[cshape], c->n_chain, 64), nullptr, 0};
            if (c->slide_items > 0) {
                // sliding evaluation: the patch planes of ALL chain items and the column planes of the items the plan left out run on
                // the first-pass kernel, the rest slides
                // (the sliding kernel counts every row of a window as a plain column slice: the patch planes add the patch-list rows'
                // real k-mers, their plain-slice planes take the plain counts back — the exclusion words are not read at all)
                EvalChainArgs pa2 = ca;
                int patch_blocks = 0, pgw = 8;
                if (ca.patch.n_blocks) {
                    { int rc = ensure_plain_planes(c); if (rc) return rc; }
                    // chains of more than 8 members: the long-chain kernel, a launch of its own, units of 8 words per thread; else the
                    // units are the tail of the sliding launch, one word per lane (chainbody.hpp: eval_patch_wave)
                    const bool long_chains = c->max_steps > kEvalCC;
                    pgw = long_chains || c->max_npw > 64 ? 8 : 1;
                    pa2.patch = patch_args(c, pgw, c->n_chain, 64);
                    const PatchArgs neg = patch_args(c, pgw, c->slide_items, 64);
                    pa2.patch.qplanes = c->qplanes; pa2.patch.qvalid = c->qvalid; pa2.patch.neg_blocks = neg.n_blocks;
                    pa2.neg_items = c->chain_slid; pa2.n_neg = c->slide_items;
                    patch_blocks = pa2.patch.n_blocks + neg.n_blocks;
                    if (long_chains) {
                        hipLaunchKernelGGL(lfn[c->v][7], dim3((unsigned)patch_blocks), dim3(kBlock), 0, c->stream, pa2);
                        patch_blocks = 0;
                    }
                }
                if (c->n_rest) {
                    unsigned rgrid;
                    const BlockMap rbm = block_map(cgw[cshape], c->n_rest, rgrid);
                    PatchArgs none{nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0};
                    EvalChainArgs ra{c->cols, c->excl, nw, c->p0, c->k, c->v, c->chain_rest, c->chain_events, c->cand_out, (uint32_t)c->sF, (uint32_t)c->sR,
                                     (unsigned long long *)device_out, rbm, none, nullptr, 0};
                    hipLaunchKernelGGL((c->rest_max_steps > kEvalCC ? lfn : cfn)[c->v][cshape], dim3(rgrid), dim3(kBlock), 0, c->stream, ra);
                }
                int rc = launch_eval_slide(c, (unsigned long long *)device_out, patch_blocks ? &pa2 : nullptr, patch_blocks, pgw, take_clear(), n_clear);
                if (rc) return rc;
            } else {
            if (use_prog) {
                // program-driven kernel (evalprog.hip): same arithmetic and block map (chosen at upload time, see there)
                int rc = launch_eval_prog(c, cshape, bm, ca.patch, grid, (unsigned long long *)device_out);
                if (rc) return rc;
            } else {
                ca.clear = take_clear(); ca.n_clear = n_clear;
                hipLaunchKernelGGL((c->max_steps > kEvalCC ? lfn : cfn)[c->v][cshape], dim3(grid + (unsigned)ca.patch.n_blocks), dim3(kBlock), 0,
                                   c->stream, ca);
            }
            }
        }
        const int n_tab = shape == 0 ? c->n_table : c->n_items;
        if (n_tab) {
            unsigned grid;
            const BlockMap bm = block_map(2, n_tab, grid);
            EvalBitsArgs ba{c->cols, c->excl, nw, c->p0, c->k, c->v, c->items, c->cand_symT, c->cand_out, (uint32_t)c->sF, (uint32_t)c->sR,
                            (unsigned long long *)device_out, bm, c->cand_diff, shape == 0 ? c->table_ids : (const int32_t *)nullptr,
                            patch_args(c, 2, n_tab, 64), nullptr, nullptr, c->n_rows, take_clear(), n_clear};
            hipLaunchKernelGGL(tfn[c->v][shape == 1 ? 1 : 0], dim3(grid + (unsigned)ba.patch.n_blocks), dim3(kBlock), 0, c->stream, ba);
        }
    } else if (c->wide) {
        const EvalArgsT<uint64_t> ea{msa_args(c), c->p0, c->n_pad, c->k, c->items, reinterpret_cast<const CandN<uint64_t> *>(c->cand_n), c->cand_out,
                                     c->n_extra ? c->extra_off : (const int32_t *)nullptr, reinterpret_cast<const uint64_t *>(c->extra_words), c->sF, c->sR,
                                     c->v, kmask_of<uint64_t>(c->k), rows, (unsigned long long *)device_out};
        const dim3 grid((unsigned)c->n_items, (unsigned)split);
        if (vmode == 0) hipLaunchKernelGGL((eval_kernel<kEvalCC, 0, 1, true, 2, uint64_t>), grid, dim3(kBlock), 0, c->stream, ea);
        else if (vmode == 1 && !getenv("MP_EVAL_GENERIC_V")) hipLaunchKernelGGL((eval_kernel<kEvalCC, 1, 1, true, 2, uint64_t>), grid, dim3(kBlock), 0, c->stream, ea);
        else hipLaunchKernelGGL((eval_kernel<kEvalCC, 2, 1, true, 2, uint64_t>), grid, dim3(kBlock), 0, c->stream, ea);
    } else {
    EvalArgs ea{msa_args(c), c->p0, c->n_pad, c->k, c->items, reinterpret_cast<const CandN<uint32_t> *>(c->cand_n), c->cand_out,
                c->n_extra ? c->extra_off : (const int32_t *)nullptr, c->extra_words, (uint32_t)c->sF, (uint32_t)c->sR, c->v, (1u << c->k) - 1u, rows,
                (unsigned long long *)device_out};
    int variant = c->eval_variant;
    if (const char *e = getenv("MP_EVAL_VARIANT")) variant = atoi(e);
    if (variant < 0 || variant >= kNumEvalVariants) variant = 0;
    hipLaunchKernelGGL(kEvalVariants[variant].fn[getenv("MP_EVAL_GENERIC_V") ? 2 : vmode], dim3((unsigned)c->n_items, (unsigned)split),
                       dim3(kBlock), 0, c->stream, ea);
    }
    if (pending_clear) zero(device_clear);                  // no kernel of this step could take the side job (row-per-lane / program-driven forms)
    if (timed) {
        HIPCK(c, hipEventRecord(ev.second, c->stream));
        c->ev_busy.push_back(ev);
    }
    HIPCK(c, hipGetLastError());
    return MP_OK;
}

int mp_eval_launch(mp_ctx *c, int64_t *device_out) { return eval_launch_impl(c, device_out, nullptr, true); }

int mp_eval_launch_rotating(mp_ctx *c, int64_t *device_out, int64_t *device_clear) { return eval_launch_impl(c, device_out, device_clear, false); }

int mp_eval_launch_alt(mp_ctx *c, int64_t *device_out) {
    if (!c) return MP_ERR_ARG;
    HIPCK(c, hipSetDevice(c->dev));
    if (c->pp_dirty || (c->slide_items > 0 && c->qp_dirty && c->n_patch))
        return fail(c, MP_ERR_ARG, "mp_eval_launch_alt: the first launch of a staged candidate set has to be mp_eval_launch");
    if (!c->alt_stream) {
        HIPCK(c, hipStreamSynchronize(c->stream));            // whatever the first launches built is there before the second stream exists
        HIPCK(c, hipStreamCreateWithFlags(&c->alt_stream, hipStreamNonBlocking));
    }
    hipStream_t keep = c->stream;
    c->stream = c->alt_stream;                                // (everything a launch enqueues goes to the context's stream)
    const int rc = eval_launch_impl(c, device_out, nullptr, true);
    c->stream = keep;
    return rc;
}

int mp_eval_sync(mp_ctx *c) {
    if (!c) return MP_ERR_ARG;
    HIPCK(c, hipSetDevice(c->dev));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (c->alt_stream) HIPCK(c, hipStreamSynchronize(c->alt_stream));
    return MP_OK;
}

int mp_eval_timing(mp_ctx *c, int32_t reset, double *total_ms, int32_t *n_launches) {
    if (!c) return MP_ERR_ARG;
    HIPCK(c, hipSetDevice(c->dev));
    for (auto &p : c->ev_busy) {
        HIPCK(c, hipEventSynchronize(p.second));
        float ms = 0;
        HIPCK(c, hipEventElapsedTime(&ms, p.first, p.second));
        c->ev_ms += ms;
        c->ev_n++;
        if (c->ev_samples.size() < 4096) c->ev_samples.push_back(ms);
        c->ev_free.push_back(p);
    }
    c->ev_busy.clear();
    if (total_ms) *total_ms = c->ev_ms;
    if (n_launches) *n_launches = c->ev_n;
    c->ev_last = c->ev_samples;
    if (reset) { c->ev_ms = 0; c->ev_n = 0; c->launch_seq = 0; c->ev_samples.clear(); }      // the first launch after a reset is a timed one
    return MP_OK;
}

int mp_eval_plan_info(mp_ctx *c, int32_t *info) {
    if (!c || !info) return MP_ERR_ARG;
    info[0] = c->n_chain; info[1] = c->n_table; info[2] = c->slide_items; info[3] = c->slide_items ? c->n_rest : 0;
    return MP_OK;
}

int mp_eval_timing_samples(mp_ctx *c, int32_t cap, float *ms, int32_t *n) {
    if (!c || !n) return MP_ERR_ARG;
    *n = (int32_t)c->ev_last.size();
    if (ms) for (int i = 0; i < *n && i < cap; i++) ms[i] = c->ev_last[(size_t)i];
    return MP_OK;
}

// the statistics launch on the context's current stream into stats_buf (n_f frequency + n_t pair counters)
static int window_stats_launch(mp_ctx *c, size_t &n_f, size_t &n_t) {
    const size_t W = (size_t)c->n_win, k = (size_t)c->k;
    n_f = W * 4 * k; n_t = W * (k - 1) * 16;
    int rc;
    if (c->stats_buf_n < n_f + n_t) {
        dev_free(c, &c->stats_buf, c->stats_buf_n);
        c->stats_buf_n = 0;
        if ((rc = dev_alloc(c, &c->stats_buf, n_f + n_t))) return rc;
        c->stats_buf_n = n_f + n_t;
    }
    unsigned long long *d = c->stats_buf;
    HIPCK(c, hipMemsetAsync(d, 0, sizeof(unsigned long long) * (n_f + n_t), c->stream));
    const int nw = c->n_pad / 64;
    // words per thread: 4 from 32768 rows up (8: 0.64 ms against 0.34 at 131072 x 1000 — half the waves, 2: 0.36)
    const int GW = 2 * nw >= 4 * kBlock ? 4 : (2 * nw >= 2 * kBlock ? 2 : 1);
    BlockMap m;
    m.ny = std::max(1, (2 * nw / GW + kBlock - 1) / kBlock);
    m.ny_pad = m.ny > 4 ? (m.ny + 7) / 8 * 8 : (m.ny > 2 ? 4 : m.ny);
    m.n_items = c->n_win;
    const int bands = m.ny_pad >= 8 ? 1 : 8 / m.ny_pad;
    m.per_band = (c->n_win + bands - 1) / bands;
    const unsigned grid = m.ny_pad >= 8 ? (unsigned)((size_t)c->n_win * m.ny_pad) : (unsigned)(8 * (size_t)m.per_band);
    StatsArgs sa{c->cols, c->excl, nw, c->p0, c->k, c->v, m, c->n_win, d, d + n_f, patch_args(c, GW, c->n_win, kBlock)};
    // [r6] the plain rows of G consecutive windows per workgroup (window_stats_group_kernel) where the alignment is deep enough for the
    // per-window form to be bound by its L2 re-reads; the patch planes stay with the per-window kernel (its patch blocks only then).
    // MP_STATS_GROUP=0 keeps the per-window kernel for everything, =4 / 8 picks G.
    int G = GW == 4 ? 4 : 0;
    if (const char *e = getenv("MP_STATS_GROUP")) { const int g = atoi(e); G = (g == 4 || g == 8) && GW == 4 ? g : 0; }
    if (G) {
        const int n_groups = (c->n_win + G - 1) / G;
        unsigned ggrid;
        StatsArgs sg = sa;
        sg.map = make_block_map(nw, GW, n_groups, ggrid);
        const dim3 gfull(ggrid + (unsigned)sg.patch.n_blocks);          // (the patch blocks first, as in the per-window launch)
        if (G == 8) hipLaunchKernelGGL((window_stats_group_kernel<4, 8>), gfull, dim3(kBlock), 0, c->stream, sg);
        else hipLaunchKernelGGL((window_stats_group_kernel<4, 4>), gfull, dim3(kBlock), 0, c->stream, sg);
        HIPCK(c, hipGetLastError());
        return MP_OK;
    }
    const dim3 full(grid + (unsigned)sa.patch.n_blocks);
    if (GW == 4) hipLaunchKernelGGL(window_stats_kernel<4>, full, dim3(kBlock), 0, c->stream, sa);
    else if (GW == 2) hipLaunchKernelGGL(window_stats_kernel<2>, full, dim3(kBlock), 0, c->stream, sa);
    else hipLaunchKernelGGL(window_stats_kernel<1>, full, dim3(kBlock), 0, c->stream, sa);
    HIPCK(c, hipGetLastError());
    return MP_OK;
}

int mp_window_stats(mp_ctx *c, int64_t *freq, int64_t *nn) {
    if (!c) return MP_ERR_ARG;
    if (!c->excl) return fail(c, MP_ERR_ARG, "no windows built");
    if (!freq || !nn) return fail(c, MP_ERR_ARG, "null output");
    HIPCK(c, hipSetDevice(c->dev));
    int rc;
    if ((rc = ensure_patch_planes(c))) return rc;
    size_t n_f = 0, n_t = 0;
    if ((rc = window_stats_launch(c, n_f, n_t))) return rc;
    unsigned long long *d = c->stats_buf;
    HIPCK(c, hipMemcpyAsync(freq, d, sizeof(int64_t) * n_f, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(nn, d + n_f, sizeof(int64_t) * n_t, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

// [r6] The same in two halves, for a caller that has something else to put on the device's copy engines meanwhile — the streamed planning
// (mp_plan_create_streamed) reads 50-100 MB of histogram entries back while the statistics kernel (2.8 ms at 10^6 rows) runs: begin
// launches the kernel and the read-back of its counters into a registered buffer of the context on the SECOND stream and returns at once;
// end waits for them and hands the counters over.  mp_plan_create_streamed calls end itself (before its planners read a counter) when a
// begin is pending and is given the arrays the counters belong in.
int mp_window_stats_begin(mp_ctx *c) {
    if (!c) return MP_ERR_ARG;
    if (!c->excl) return fail(c, MP_ERR_ARG, "no windows built");
    HIPCK(c, hipSetDevice(c->dev));
    int rc;
    if ((rc = ensure_patch_planes(c))) return rc;                      // (on the first stream)
    if (!c->alt_stream) HIPCK(c, hipStreamCreateWithFlags(&c->alt_stream, hipStreamNonBlocking));
    if (!c->stats_ev) HIPCK(c, hipEventCreateWithFlags(&c->stats_ev, hipEventDisableTiming));
    // the second stream starts behind everything the first has queued so far (planes, windows, patch planes)
    HIPCK(c, hipEventRecord(c->stats_ev, c->stream));
    HIPCK(c, hipStreamWaitEvent(c->alt_stream, c->stats_ev, 0));
    const size_t W = (size_t)c->n_win, k = (size_t)c->k, n = W * 4 * k + W * (k - 1) * 16, bytes = sizeof(int64_t) * n;
    if (c->h_stats_bytes < bytes) {
        if (c->h_stats) { if (c->h_stats_pinned) (void)hipHostUnregister(c->h_stats); host_unmap(c->h_stats, c->h_stats_bytes); }
        c->h_stats_pinned = false; c->h_stats_bytes = 0;
        const size_t room = (bytes + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
        c->h_stats = static_cast<uint8_t *>(host_map(room));
        if (!c->h_stats) return fail(c, MP_ERR_NOMEM, "mp_window_stats_begin: out of host memory");
        c->h_stats_bytes = room;
        prefault_host(c->h_stats, room);
        if (!getenv("MP_NO_PIN") && hipHostRegister(c->h_stats, room, hipHostRegisterDefault) == hipSuccess) c->h_stats_pinned = true;
        else (void)hipGetLastError();
    }
    hipStream_t keep = c->stream;
    c->stream = c->alt_stream;
    size_t n_f = 0, n_t = 0;
    rc = window_stats_launch(c, n_f, n_t);
    hipError_t e = hipSuccess;
    if (rc == MP_OK) e = hipMemcpyAsync(c->h_stats, c->stats_buf, bytes, hipMemcpyDeviceToHost, c->stream);
    if (rc == MP_OK && e == hipSuccess) e = hipEventRecord(c->stats_ev, c->stream);
    c->stream = keep;
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, MP_ERR_DEVICE, "mp_window_stats_begin: %s", hipGetErrorString(e));
    c->stats_pending_f = n_f; c->stats_pending_t = n_t;
    return MP_OK;
}

int mp_window_stats_end(mp_ctx *c, int64_t *freq, int64_t *nn) {
    if (!c) return MP_ERR_ARG;
    if (!c->stats_pending_f) return fail(c, MP_ERR_ARG, "mp_window_stats_end: no mp_window_stats_begin is pending");
    if (!freq || !nn) return fail(c, MP_ERR_ARG, "null output");
    HIPCK(c, hipSetDevice(c->dev));
    HIPCK(c, hipEventSynchronize(c->stats_ev));
    memcpy(freq, c->h_stats, sizeof(int64_t) * c->stats_pending_f);
    memcpy(nn, c->h_stats + sizeof(int64_t) * c->stats_pending_f, sizeof(int64_t) * c->stats_pending_t);
    c->stats_pending_f = c->stats_pending_t = 0;
    return MP_OK;
}

int mp_eval_candidates(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint64_t sF, uint64_t sR,
                       int64_t *out) {
    if (!c) return MP_ERR_ARG;
    int rc = mp_eval_upload(c, n_cand, cw, codes, sF, sR);
    if (rc) return rc;
    if (n_cand == 0) return MP_OK;
    if (!out) return fail(c, MP_ERR_ARG, "null output");
    if (c->tmp_out_n < 3 * n_cand) {
        dev_free(c, &c->tmp_out, (size_t)c->tmp_out_n);
        c->tmp_out_n = 0;
        if ((rc = dev_alloc(c, &c->tmp_out, (size_t)3 * n_cand))) return rc;
        c->tmp_out_n = 3 * n_cand;
    }
    if ((rc = mp_eval_launch(c, (int64_t *)c->tmp_out))) return rc;
    HIPCK(c, hipMemcpyAsync(out, c->tmp_out, sizeof(int64_t) * 3 * (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}


int mp_eval_masks_resident(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint64_t sF, uint64_t sR) {
    if (!c) return MP_ERR_ARG;
    int rc = mp_eval_upload(c, n_cand, cw, codes, sF, sR);
    if (rc) return rc;
    const size_t nw = (size_t)c->n_pad / 64;
    dev_free(c, &c->mask_f, c->mask_words); dev_free(c, &c->mask_r, c->mask_words);
    c->mask_words = 0; c->n_masks = 0;
    if (n_cand == 0) return MP_OK;
    if ((rc = dev_alloc(c, &c->mask_f, (size_t)n_cand * nw))) return rc;
    if ((rc = dev_alloc(c, &c->mask_r, (size_t)n_cand * nw))) return rc;
    c->mask_words = (size_t)n_cand * nw;
    c->n_masks = n_cand;
    if (c->wide) {                                 // primers of 32..63 bases: the row-per-thread kernels on 64-bit words
        const EvalArgsT<uint64_t> ew{msa_args(c), c->p0, c->n_pad, c->k, c->items, reinterpret_cast<const CandN<uint64_t> *>(c->cand_n), c->cand_out,
                                     nullptr, nullptr, c->sF, c->sR, c->v, kmask_of<uint64_t>(c->k), 0, nullptr};
        const dim3 grid((unsigned)c->n_items, (unsigned)(c->n_pad / kBlock));
        hipLaunchKernelGGL((mask_rows_kernel<kEvalCC, uint64_t>), grid, dim3(kBlock), 0, c->stream, ew, c->n_rows, c->mask_f, c->mask_r);
        if (c->n_patch)
            hipLaunchKernelGGL((mask_patch_kernel<kEvalCC, uint64_t>), dim3((unsigned)c->n_items), dim3(kBlock), 0, c->stream, ew, (const int32_t *)c->patch_off,
                               (const int32_t *)c->patch_rows, reinterpret_cast<const uint64_t *>(c->patch_words), c->mask_f, c->mask_r);
        HIPCK(c, hipGetLastError());
        HIPCK(c, hipStreamSynchronize(c->stream));
        return MP_OK;
    }
    EvalArgs ea{msa_args(c), c->p0, c->n_pad, c->k, c->items, reinterpret_cast<const CandN<uint32_t> *>(c->cand_n), c->cand_out, nullptr, nullptr,
                (uint32_t)c->sF, (uint32_t)c->sR, c->v, (1u << c->k) - 1u, 0, nullptr};
    // the plain rows: bit-sliced on the column planes (the evaluation pass itself, its final words stored instead of counted);
    // MP_MASK_MODE=rows keeps the row-per-thread kernel of rounds 1-2 (0.32 ms against 0.0x ms at 131072 rows x 410 windows)
    const char *mm = getenv("MP_MASK_MODE");
    if ((mm && !strcmp(mm, "rows")) || c->v > 3) {                 // (four counter levels in the bit-sliced kernels: v <= 3)
        const dim3 grid((unsigned)c->n_items, (unsigned)(c->n_pad / kBlock));
        hipLaunchKernelGGL((mask_rows_kernel<kEvalCC>), grid, dim3(kBlock), 0, c->stream, ea, c->n_rows, c->mask_f, c->mask_r);
    } else {
        static const EvalBitsFn mfn[4] = {eval_bits_kernel<1, 2, true, 1, true>, eval_bits_kernel<2, 2, true, 1, true>,
                                          eval_bits_kernel<3, 2, true, 1, true>, eval_bits_kernel<4, 2, true, 1, true>};
        unsigned grid;
        const BlockMap bm = make_block_map((int)nw, 2, c->n_items, grid);
        PatchArgs none{nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0};
        EvalBitsArgs ba{c->cols, c->excl, (int)nw, c->p0, c->k, c->v, c->items, c->cand_symT, c->cand_out, (uint32_t)c->sF, (uint32_t)c->sR, nullptr, bm, c->cand_diff,
                        nullptr, none, reinterpret_cast<uint32_t *>(c->mask_f), reinterpret_cast<uint32_t *>(c->mask_r), c->n_rows};
        hipLaunchKernelGGL(mfn[c->v], dim3(grid), dim3(kBlock), 0, c->stream, ba);
    }
    if (c->n_patch)
        hipLaunchKernelGGL((mask_patch_kernel<kEvalCC>), dim3((unsigned)c->n_items), dim3(kBlock), 0, c->stream, ea, (const int32_t *)c->patch_off,
                           (const int32_t *)c->patch_rows, (const uint32_t *)c->patch_words, c->mask_f, c->mask_r);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

int mp_masks_set_bits(mp_ctx *c, int64_t n, const int32_t *cand, const int32_t *row, const uint8_t *which, const uint8_t *value) {
    if (!c) return MP_ERR_ARG;
    if (n < 0 || (n && (!cand || !row || !which || !value))) return fail(c, MP_ERR_ARG, "mp_masks_set_bits: bad arguments");
    if (n == 0) return MP_OK;
    if (!c->mask_f) return fail(c, MP_ERR_ARG, "no resident masks (mp_eval_masks_resident has not run)");
    for (int64_t i = 0; i < n; i++)
        if (cand[i] < 0 || cand[i] >= c->n_masks || row[i] < 0 || row[i] >= c->n_rows) return fail(c, MP_ERR_ARG, "assignment %lld out of range", (long long)i);
    HIPCK(c, hipSetDevice(c->dev));
    int32_t *d_c = nullptr, *d_r = nullptr;
    uint8_t *d_w = nullptr, *d_v = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &d_c, (size_t)n)) || (rc = dev_alloc(c, &d_r, (size_t)n)) || (rc = dev_alloc(c, &d_w, (size_t)n)) || (rc = dev_alloc(c, &d_v, (size_t)n))) {
        dev_free(c, &d_c, (size_t)n); dev_free(c, &d_r, (size_t)n); dev_free(c, &d_w, (size_t)n); dev_free(c, &d_v, (size_t)n);
        return rc;
    }
    hipError_t e = hipMemcpyAsync(d_c, cand, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_r, row, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_w, which, (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_v, value, (size_t)n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mask_set_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, (long long)n, (const int32_t *)d_c,
                           (const int32_t *)d_r, (const uint8_t *)d_w, (const uint8_t *)d_v, (size_t)c->n_pad / 64, c->mask_f, c->mask_r);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(c, &d_c, (size_t)n); dev_free(c, &d_r, (size_t)n); dev_free(c, &d_w, (size_t)n); dev_free(c, &d_v, (size_t)n);
    if (e != hipSuccess) return fail(c, MP_ERR_DEVICE, "mp_masks_set_bits: %s", hipGetErrorString(e));
    return MP_OK;
}

int mp_masks_fetch(mp_ctx *c, uint64_t *not_f, uint64_t *not_r) {
    if (!c) return MP_ERR_ARG;
    if (!c->mask_f) return fail(c, MP_ERR_ARG, "no resident masks (mp_eval_masks_resident has not run)");
    if (!not_f || !not_r) return fail(c, MP_ERR_ARG, "null output");
    HIPCK(c, hipSetDevice(c->dev));
    // rows are padded to a multiple of 256 on the device: copy the (n_rows+63)/64 meaningful words of each mask
    const size_t nw = (size_t)c->n_pad / 64, nwo = ((size_t)c->n_rows + 63) / 64;
    HIPCK(c, hipMemcpy2DAsync(not_f, nwo * 8, c->mask_f, nw * 8, nwo * 8, (size_t)c->n_masks, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpy2DAsync(not_r, nwo * 8, c->mask_r, nw * 8, nwo * 8, (size_t)c->n_masks, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

int mp_pair_coverage_resident(mp_ctx *c, int64_t n_pairs, const int32_t *pairs, int32_t *out) {
    if (!c) return MP_ERR_ARG;
    if (n_pairs < 0 || (n_pairs && (!pairs || !out))) return fail(c, MP_ERR_ARG, "mp_pair_coverage_resident: bad arguments");
    if (n_pairs == 0) return MP_OK;
    if (!c->mask_f) return fail(c, MP_ERR_ARG, "no resident masks (mp_eval_masks_resident has not run)");
    for (int64_t p = 0; p < 2 * n_pairs; p++)
        if (pairs[p] < 0 || pairs[p] >= c->n_masks) return fail(c, MP_ERR_ARG, "pair %lld out of range", (long long)(p / 2));
    HIPCK(c, hipSetDevice(c->dev));
    int32_t *d_pairs = nullptr, *d_out = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &d_pairs, (size_t)2 * n_pairs))) return rc;
    if ((rc = dev_alloc(c, &d_out, (size_t)n_pairs))) { dev_free(c, &d_pairs, (size_t)2 * n_pairs); return rc; }
    hipError_t e = hipMemcpyAsync(d_pairs, pairs, sizeof(int32_t) * 2 * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const long long per_block = kBlock / 64;
        hipLaunchKernelGGL(mask_pair_kernel, dim3((unsigned)((n_pairs + per_block - 1) / per_block)), dim3(kBlock), 0, c->stream,
                           (const unsigned long long *)c->mask_f, (const unsigned long long *)c->mask_r, (int)((c->n_rows + 63) / 64),
                           (size_t)c->n_pad / 64, (long long)n_pairs, (const int32_t *)d_pairs, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(int32_t) * (size_t)n_pairs, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(c, &d_pairs, (size_t)2 * n_pairs); dev_free(c, &d_out, (size_t)n_pairs);
    if (e != hipSuccess) return fail(c, MP_ERR_DEVICE, "mp_pair_coverage_resident: %s", hipGetErrorString(e));
    return MP_OK;
}

int mp_eval_masks(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint64_t sF, uint64_t sR,
                  uint64_t *not_f, uint64_t *not_r) {
    if (!c) return MP_ERR_ARG;
    int rc = mp_eval_masks_resident(c, n_cand, cw, codes, sF, sR);
    if (rc) return rc;
    if (n_cand == 0) return MP_OK;
    if (!not_f || !not_r) return fail(c, MP_ERR_ARG, "null output");
    return mp_masks_fetch(c, not_f, not_r);
}

}  // extern "C"
