// eval.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of
// include/mprime.h.  Candidate x sequence coverage evaluation: bit-sliced kernel, list kernel, row-per-lane kernel (mp_eval_*).
#include "common.hpp"

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
__device__ inline uint32_t bfi(uint32_t s, uint32_t a, uint32_t b) { return (s & a) | (~s & b); }

struct EvalArgs {
    const void *win;
    int n_pad, k;
    const EvalItem *items;
    const uint4 *cand_n;        // [padded cand] nA,nC,nG,nT
    const int32_t *cand_out;    // [padded cand] index into out or -1
    const int32_t *extra_off;   // [W+1] or nullptr
    const uint32_t *extra_words;
    uint32_t sF, sR;
    int v;
    uint32_t kmask;
    int rows_per_split;
    unsigned long long *out;
};

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
template <int CC, int VMODE, int COUNT, int FORM>
__device__ inline void eval_row(uint32_t b0, uint32_t b1, uint32_t g, const EvalArgs &A,
                                const uint32_t (&nA)[CC], const uint32_t (&nC)[CC], const uint32_t (&nG)[CC],
                                const uint32_t (&nT)[CC], EvalAcc<CC, COUNT> &acc) {
    uint32_t gk = g & A.kmask;
    // rows outside the universe (SKIP slots and k-mers with more than v gaps, V20:689) get an
    // all-ones mismatch word: 32 mismatches, counted nowhere
    if ((int)__popc(gk) > A.v) gk = 0xFFFFFFFFu;
    uint32_t eA = 0, eC = 0, eG = 0, eT = 0;
    if (FORM == 2) { eA = ~(b0 | b1 | gk); eC = b0 & ~b1; eG = b1 & ~b0; eT = b0 & b1; }
#pragma unroll
    for (int c = 0; c < CC; c++) {
        uint32_t mm;
        if (FORM == 2) mm = (eT & nT[c]) | ((eG & nG[c]) | ((eC & nC[c]) | ((eA & nA[c]) | gk)));
        else mm = bfi(b1, bfi(b0, nT[c], nG[c]), bfi(b0, nC[c], nA[c])) | gk;
        bool pp = mm == 0, ff, rr;
        if (VMODE == 0) {
            ff = rr = pp;
        } else if (VMODE == 1) {
            uint32_t t;
            if (FORM == 0) t = mm - 1u;
            else asm("v_add_u32_e32 %0, -1, %1" : "=v"(t) : "v"(mm));   // no carry-out: keeps `mm == 0` a plain v_cmp
            ff = (mm & (t | A.sF)) == 0;
            rr = (mm & (t | A.sR)) == 0;
        } else {
            bool le = (int)__popc(mm) <= A.v;
            ff = le && (mm & A.sF) == 0;
            rr = le && (mm & A.sR) == 0;
        }
        acc.add(c, pp, ff, rr);
    }
}

template <int CC, int VMODE, int COUNT, bool PREFETCH, int FORM, bool P64>
__global__ __launch_bounds__(kBlock) void eval_kernel(const EvalArgs A) {
    __shared__ uint32_t s_acc[3 * CC];
    const EvalItem it = A.items[blockIdx.x];
    uint32_t nA[CC], nC[CC], nG[CC], nT[CC];
    EvalAcc<CC, COUNT> acc;
    acc.clear();
#pragma unroll
    for (int c = 0; c < CC; c++) {
        uint4 q = A.cand_n[it.cand0 + c];
        nA[c] = q.x; nC[c] = q.y; nG[c] = q.z; nT[c] = q.w;
        if (FORM == 1) {
            asm volatile("" : "+v"(nA[c]));
            asm volatile("" : "+v"(nC[c]));
            asm volatile("" : "+v"(nG[c]));
            asm volatile("" : "+v"(nT[c]));
        }
    }
    if (threadIdx.x < 3 * CC) s_acc[threadIdx.x] = 0;
    const size_t np = (size_t)A.n_pad;
    const WinView<P64> V(A.win, it.win, np, A.k, A.kmask);
    typedef typename WinView<P64>::Raw4 Raw4;
    const int r0 = blockIdx.y * A.rows_per_split;
    const int r1 = r0 + A.rows_per_split < A.n_pad ? r0 + A.rows_per_split : A.n_pad;
    // 4 consecutive sequences per lane and iteration, 16-byte loads (n_pad % 4 == 0)
    int r = r0 + threadIdx.x * 4;
    Raw4 cur;
    if (PREFETCH && r < r1) cur = V.load4(r);
#pragma unroll 1
    while (r < r1) {
        const int rn = r + kBlock * 4;
        Raw4 now;
        if (PREFETCH) {
            now = cur;
            if (rn < r1) cur = V.load4(rn);          // next group in flight while this one computes
        } else {
            now = V.load4(r);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t b0, b1, g;
            V.unpack(now, i, b0, b1, g);
            eval_row<CC, VMODE, COUNT, FORM>(b0, b1, g, A, nA, nC, nG, nT, acc);
        }
        r = rn;
    }
    if (blockIdx.y == 0 && A.extra_off) {      // host-expanded IUPAC rows of this window
        const int e0 = A.extra_off[it.win], e1 = A.extra_off[it.win + 1];
        for (int eb = e0; eb < e1; eb += kBlock) {   // uniform trip count: COUNT 1 ballots need every lane
            int e = eb + threadIdx.x;
            uint32_t b0 = 0, b1 = 0, g = 0xFFFFFFFFu;
            if (e < e1) { b0 = A.extra_words[3 * e]; b1 = A.extra_words[3 * e + 1]; g = A.extra_words[3 * e + 2]; }
            eval_row<CC, VMODE, COUNT, FORM>(b0, b1, g, A, nA, nC, nG, nT, acc);
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
// (4b) bit-sliced evaluation: 64 sequences per register word
// ----------------------------------------------------------------------------------------------
// For the sequences whose k-mer at window w is the plain column slice (everything the patch list
// does not hold) the symbol at window position j is column p0+w+j of the alignment, so the
// evaluation can run on the COLUMN planes: a thread owns G words of 64 sequences; for every
// position it loads the three plane words once, and for every candidate the mismatch word of 64
// sequences is ONE v_bitop3 of (g,b1,b0) whose truth table is fixed by the candidate's symbol
// (a wave-uniform 16-way dispatch).  Mismatch counts are bit-sliced saturating counters
// (t1 = ">= 1", t2 = ">= 2", t3 = ">= 3": LV = v+1 levels), the strict-position sets are two more
// words, and the three coverage counters are popcounts at the end.  ~2.5 VALU per evaluation
// instead of ~13, and the inputs (N*L*3/8 bytes) stay in L2 / Infinity Cache.
struct EvalBitsArgs {
    const unsigned long long *cols;    // [n_cols][3][nw]
    const unsigned long long *excl;    // [W][nw]
    int nw, p0, k, v;
    const EvalItem *items;
    const uint32_t *cand_symT;         // [item][32] u32: nibble c of word j = symbol of candidate c at position j
    const int32_t *cand_out;
    uint32_t sF, sR;
    unsigned long long *out;
    int ny, ny_pad;                    // row slices per item; ny_pad = ny rounded up to a multiple of 8 (XCDs)
};

// truth table of v_bitop3_b32 D = f(S0,S1,S2): bit (S0<<2 | S1<<1 | S2) of the immediate
constexpr int bs_lut_mismatch(int sym) {      // inputs (g, b1, b0): gap, or base not in the symbol's set
    int t = 0;
    for (int idx = 0; idx < 8; idx++) {
        int g = idx >> 2, base = idx & 3;
        if (g || !((sym >> base) & 1)) t |= 1 << idx;
    }
    return t;
}
constexpr int kLutOrAnd = 0xF8;               // S0 | (S1 & S2)
constexpr int kLutAndNotNot = 0x10;           // S0 & ~S1 & ~S2

template <int SYM, int GW>
__device__ inline void bs_mismatch(const uint32_t (&b0)[GW], const uint32_t (&b1)[GW], const uint32_t (&g)[GW], uint32_t (&m)[GW]) {
    constexpr int lut = bs_lut_mismatch(SYM);
#pragma unroll
    for (int i = 0; i < GW; i++) m[i] = __builtin_amdgcn_bitop3_b32(g[i], b1[i], b0[i], lut);
}

// CP candidates per pass over the k positions (8 / CP passes), GW 32-bit words (32 sequences each)
// per thread, LV = v + 1 saturating counter levels.
template <int CP, int LV, int GW, bool PREFETCH>
__global__ __launch_bounds__(kBlock) void eval_bits_kernel(const EvalBitsArgs A) {
    constexpr int CC = 8;
    __shared__ uint32_t s_acc[3 * CC];
    // XCD-aware block mapping: workgroup b runs on XCD b % 8 (observed dispatch order), so all blocks of
    // one row slice land on the same XCD and consecutive windows re-read their 17 shared columns from
    // that XCD's L2 (a slice of the planes is 1/ny of N*L*3/8 bytes)
    const int slice = blockIdx.x % A.ny_pad, item = blockIdx.x / A.ny_pad;
    if (slice >= A.ny) return;
    const EvalItem it = A.items[item];
    if (threadIdx.x < 3 * CC) s_acc[threadIdx.x] = 0;
    const size_t nw32 = (size_t)A.nw * 2;             // 32-bit words per plane row
    const int word0 = (slice * kBlock + threadIdx.x) * GW;
    const bool live = word0 < (int)nw32;              // nw32 % GW == 0 (n_pad % 256 == 0, GW <= 8)
    const uint32_t *cols = reinterpret_cast<const uint32_t *>(A.cols);
    uint32_t accP[CC], accF[CC], accR[CC];
#pragma unroll
    for (int c = 0; c < CC; c++) accP[c] = accF[c] = accR[c] = 0;
    if (live) {
        uint32_t valid[GW];
        bool have_valid = false;
#pragma unroll 1
        for (int pass = 0; pass < CC / CP; pass++) {
            uint32_t t1[CP][GW], t2[CP][GW], t3[CP][GW], sf[CP][GW], sr[CP][GW];
            uint32_t g1[GW], g2[GW], g3[GW];
#pragma unroll
            for (int i = 0; i < GW; i++) {
                g1[i] = g2[i] = g3[i] = 0;
#pragma unroll
                for (int c = 0; c < CP; c++) t1[c][i] = t2[c][i] = t3[c][i] = sf[c][i] = sr[c][i] = 0;
            }
            uint32_t n0[GW], n1[GW], ng[GW];               // next position's planes, in flight during this one
            if (PREFETCH) {
                const uint32_t *P = cols + ((size_t)(A.p0 + it.win) * 3) * nw32 + word0;
#pragma unroll
                for (int i = 0; i < GW; i++) { n0[i] = P[i]; n1[i] = P[nw32 + i]; ng[i] = P[2 * nw32 + i]; }
            }
#pragma unroll 1
            for (int j = 0; j < A.k; j++) {
                uint32_t b0[GW], b1[GW], g[GW];
                if (PREFETCH) {
#pragma unroll
                    for (int i = 0; i < GW; i++) { b0[i] = n0[i]; b1[i] = n1[i]; g[i] = ng[i]; }
                    if (j + 1 < A.k) {
                        const uint32_t *P = cols + ((size_t)(A.p0 + it.win + j + 1) * 3) * nw32 + word0;
#pragma unroll
                        for (int i = 0; i < GW; i++) { n0[i] = P[i]; n1[i] = P[nw32 + i]; ng[i] = P[2 * nw32 + i]; }
                    }
                } else {
                    const uint32_t *P = cols + ((size_t)(A.p0 + it.win + j) * 3) * nw32 + word0;
#pragma unroll
                    for (int i = 0; i < GW; i++) { b0[i] = P[i]; b1[i] = P[nw32 + i]; g[i] = P[2 * nw32 + i]; }
                }
                if (!have_valid) {
#pragma unroll
                    for (int i = 0; i < GW; i++) {            // gaps per k-mer, saturating (V20:689 needs "> v")
                        if (LV >= 3) g3[i] = __builtin_amdgcn_bitop3_b32(g3[i], g2[i], g[i], kLutOrAnd);
                        if (LV >= 2) g2[i] = __builtin_amdgcn_bitop3_b32(g2[i], g1[i], g[i], kLutOrAnd);
                        g1[i] |= g[i];
                    }
                }
                const uint32_t sw = __builtin_amdgcn_readfirstlane(A.cand_symT[(size_t)item * 32 + j]) >> (4 * CP * pass);
                // the mismatch word of every possible candidate symbol at this position (15 x GW v_bitop3,
                // shared by all candidates); a candidate then picks its word by a wave-uniform register index
                uint32_t tab[16][GW];
#pragma unroll
                for (int i = 0; i < GW; i++) tab[0][i] = 0xFFFFFFFFu;
                bs_mismatch<1, GW>(b0, b1, g, tab[1]); bs_mismatch<2, GW>(b0, b1, g, tab[2]); bs_mismatch<3, GW>(b0, b1, g, tab[3]);
                bs_mismatch<4, GW>(b0, b1, g, tab[4]); bs_mismatch<5, GW>(b0, b1, g, tab[5]); bs_mismatch<6, GW>(b0, b1, g, tab[6]);
                bs_mismatch<7, GW>(b0, b1, g, tab[7]); bs_mismatch<8, GW>(b0, b1, g, tab[8]); bs_mismatch<9, GW>(b0, b1, g, tab[9]);
                bs_mismatch<10, GW>(b0, b1, g, tab[10]); bs_mismatch<11, GW>(b0, b1, g, tab[11]); bs_mismatch<12, GW>(b0, b1, g, tab[12]);
                bs_mismatch<13, GW>(b0, b1, g, tab[13]); bs_mismatch<14, GW>(b0, b1, g, tab[14]); bs_mismatch<15, GW>(b0, b1, g, tab[15]);
                uint32_t m[CP][GW];
#pragma unroll
                for (int c = 0; c < CP; c++) {
                    const uint32_t sy = (sw >> (4 * c)) & 15u;
#pragma unroll
                    for (int i = 0; i < GW; i++) m[c][i] = tab[sy][i];
#pragma unroll
                    for (int i = 0; i < GW; i++) {
                        if (LV >= 3) t3[c][i] = __builtin_amdgcn_bitop3_b32(t3[c][i], t2[c][i], m[c][i], kLutOrAnd);
                        if (LV >= 2) t2[c][i] = __builtin_amdgcn_bitop3_b32(t2[c][i], t1[c][i], m[c][i], kLutOrAnd);
                        t1[c][i] |= m[c][i];
                    }
                }
                if (__builtin_amdgcn_readfirstlane((A.sF >> j) & 1u)) {
#pragma unroll
                    for (int c = 0; c < CP; c++)
#pragma unroll
                        for (int i = 0; i < GW; i++) sf[c][i] |= m[c][i];
                }
                if (__builtin_amdgcn_readfirstlane((A.sR >> j) & 1u)) {
#pragma unroll
                    for (int c = 0; c < CP; c++)
#pragma unroll
                        for (int i = 0; i < GW; i++) sr[c][i] |= m[c][i];
                }
            }
            if (!have_valid) {
                const uint32_t *E = reinterpret_cast<const uint32_t *>(A.excl) + (size_t)it.win * nw32 + word0;
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    const uint32_t gapbad = LV == 1 ? g1[i] : (LV == 2 ? g2[i] : g3[i]);
                    valid[i] = ~(E[i] | gapbad);
                }
                have_valid = true;
            }
#pragma unroll
            for (int c = 0; c < CP; c++) {
                uint32_t p = 0, f = 0, r = 0;
#pragma unroll
                for (int i = 0; i < GW; i++) {
                    const uint32_t far = LV == 1 ? t1[c][i] : (LV == 2 ? t2[c][i] : t3[c][i]);
                    p += __popc(valid[i] & ~t1[c][i]);
                    f += __popc(__builtin_amdgcn_bitop3_b32(valid[i], far, sf[c][i], kLutAndNotNot));
                    r += __popc(__builtin_amdgcn_bitop3_b32(valid[i], far, sr[c][i], kLutAndNotNot));
                }
                // static index into the accumulators: the pass loop is not unrolled, so select by pass
#pragma unroll
                for (int q = 0; q < CC / CP; q++)
                    if (pass == q) { accP[q * CP + c] += p; accF[q * CP + c] += f; accR[q * CP + c] += r; }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CC; c++) {
        uint32_t x = accP[c], y = accF[c], z = accR[c];
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            x += __shfl_xor(x, sft);
            y += __shfl_xor(y, sft);
            z += __shfl_xor(z, sft);
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

// Row-per-lane evaluation of two compact per-window lists of window words: the patch list (rows with
// edge-gap repair / ragged ends, built on the device) and the host-expanded IUPAC rows.
struct EvalListArgs {
    const EvalItem *items;
    const uint4 *cand_n;
    const int32_t *cand_out;
    const int32_t *off_a;
    const uint32_t *words_a;
    const int32_t *off_b;       // may be nullptr
    const uint32_t *words_b;
    uint32_t sF, sR;
    int v;
    uint32_t kmask;
    unsigned long long *out;
};

template <int CC, int VMODE>
__global__ __launch_bounds__(kBlock) void eval_list_kernel(const EvalListArgs L) {
    __shared__ uint32_t s_acc[3 * CC];
    const EvalItem it = L.items[blockIdx.x];
    uint32_t nA[CC], nC[CC], nG[CC], nT[CC];
    EvalAcc<CC, 1> acc;
    acc.clear();
#pragma unroll
    for (int c = 0; c < CC; c++) {
        uint4 q = L.cand_n[it.cand0 + c];
        nA[c] = q.x; nC[c] = q.y; nG[c] = q.z; nT[c] = q.w;
    }
    if (threadIdx.x < 3 * CC) s_acc[threadIdx.x] = 0;
    EvalArgs A;
    A.sF = L.sF; A.sR = L.sR; A.v = L.v; A.kmask = L.kmask;
    for (int which = 0; which < 2; which++) {
        const int32_t *off = which ? L.off_b : L.off_a;
        const uint32_t *words = which ? L.words_b : L.words_a;
        if (!off) continue;
        const int e0 = off[it.win], e1 = off[it.win + 1];
        for (int eb = e0 + blockIdx.y * kBlock; eb < e1; eb += gridDim.y * kBlock) {     // uniform per wave
            int e = eb + threadIdx.x;
            uint32_t b0 = 0, b1 = 0, g = 0xFFFFFFFFu;
            if (e < e1) { b0 = words[3 * (size_t)e]; b1 = words[3 * (size_t)e + 1]; g = words[3 * (size_t)e + 2]; }
            eval_row<CC, VMODE, 1, 2>(b0, b1, g, A, nA, nC, nG, nT, acc);
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CC; c++) {
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&s_acc[3 * c], acc.p[c]);
            atomicAdd(&s_acc[3 * c + 1], acc.f[c] - acc.p[c]);
            atomicAdd(&s_acc[3 * c + 2], acc.r[c] - acc.p[c]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * CC) {
        int oc = L.cand_out[it.cand0 + threadIdx.x / 3];
        uint32_t val = s_acc[threadIdx.x];
        if (oc >= 0 && val) atomicAdd(&L.out[(size_t)oc * 3 + threadIdx.x % 3], (unsigned long long)val);
    }
}


// Per-sequence coverage masks (mp_eval_masks): thread = sequence, the wave's 64 "not covered" bits
// go out as one 64-bit word per candidate straight from the ballot (blocks are 64-row aligned), so
// there are no atomics; works on the window words, i.e. after edge-gap repair, for any v.
template <int CC, bool P64>
__global__ __launch_bounds__(kBlock) void mask_rows_kernel(const EvalArgs A, int n_rows, unsigned long long *__restrict__ not_f,
                                                           unsigned long long *__restrict__ not_r) {
    const EvalItem it = A.items[blockIdx.x];
    const int r = blockIdx.y * kBlock + threadIdx.x;             // n_pad is a multiple of kBlock
    const size_t nw = (size_t)A.n_pad / 64;
    const WinView<P64> V(A.win, it.win, (size_t)A.n_pad, A.k, A.kmask);
    uint32_t b0, b1, g;
    V.load(r, b0, b1, g);
    const bool skip = (g & MP_WIN_SKIP) || r >= n_rows;
    const uint32_t gk = g & A.kmask;
    const bool gap_row = (int)__popc(gk) > A.v;
#pragma unroll
    for (int c = 0; c < CC; c++) {
        const uint4 q = A.cand_n[it.cand0 + c];
        const uint32_t mm = bfi(b1, bfi(b0, q.w, q.z), bfi(b0, q.y, q.x)) | gk;
        const int d = __popc(mm);
        const bool near = d <= A.v;
        const bool bad_f = !skip && (gap_row || !(near && (d == 0 || !(mm & A.sF))));
        const bool bad_r = !skip && (gap_row || !(near && (d == 0 || !(mm & A.sR))));
        const unsigned long long wf = __ballot(bad_f), wr = __ballot(bad_r);
        const int oc = A.cand_out[it.cand0 + c];
        if ((threadIdx.x & 63) == 0 && oc >= 0) {
            not_f[(size_t)oc * nw + (size_t)(r >> 6)] = wf;
            not_r[(size_t)oc * nw + (size_t)(r >> 6)] = wr;
        }
    }
}

typedef void (*EvalBitsFn)(const EvalBitsArgs);
typedef void (*EvalListFn)(const EvalListArgs);

typedef void (*EvalFn)(const EvalArgs);
struct EvalVariant { const char *name; EvalFn fn[2][3]; };     // fn[P64][VMODE]
#define EVAL_VARIANT(name, COUNT, PREFETCH, FORM)                                                         \
    { name, { { eval_kernel<kEvalCC, 0, COUNT, PREFETCH, FORM, false>, eval_kernel<kEvalCC, 1, COUNT, PREFETCH, FORM, false>, \
                eval_kernel<kEvalCC, 2, COUNT, PREFETCH, FORM, false> },                                   \
              { eval_kernel<kEvalCC, 0, COUNT, PREFETCH, FORM, true>, eval_kernel<kEvalCC, 1, COUNT, PREFETCH, FORM, true>,   \
                eval_kernel<kEvalCC, 2, COUNT, PREFETCH, FORM, true> } } }
// variant 0 is the default; the others exist to be measured (tools/variant_bench.py, MP_EVAL_VARIANT)
const EvalVariant kEvalVariants[] = {
    EVAL_VARIANT("ballot+prefetch/onehot", 1, true, 2),      // default: fastest measured (profiles/r01_variants.txt)
    EVAL_VARIANT("ballot+prefetch/bfi-vgpr", 1, true, 1),
    EVAL_VARIANT("ballot+prefetch/bfi-sgpr", 1, true, 0),
    EVAL_VARIANT("lane-acc+prefetch/onehot", 0, true, 2),
    EVAL_VARIANT("ballot/onehot", 1, false, 2),
};
constexpr int kNumEvalVariants = (int)(sizeof(kEvalVariants) / sizeof(kEvalVariants[0]));



}  // namespace

extern "C" {

int mp_eval_upload(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint32_t sF, uint32_t sR) {
    if (!c) return MP_ERR_ARG;
    if (!c->win) return fail(c, MP_ERR_ARG, "no windows built");
    if (n_cand < 0 || (n_cand && (!cw || !codes))) return fail(c, MP_ERR_ARG, "bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    free_eval(c);
    const int k = c->k;
    const uint32_t kmask = (1u << k) - 1u;
    std::vector<EvalItem> items;
    std::vector<uint4> cn;
    std::vector<int32_t> co;
    std::vector<uint32_t> symT;
    int i = 0;
    while (i < n_cand) {
        int w = cw[i];
        if (w < 0 || w >= c->n_win || (i && w < cw[i - 1])) return fail(c, MP_ERR_ARG, "candidate windows must be ascending and in range");
        int j = i;
        while (j < n_cand && cw[j] == w) j++;
        for (int b = i; b < j; b += kEvalCC) {
            items.push_back(EvalItem{w, (int32_t)cn.size()});
            symT.resize(items.size() * 32, 0u);
            for (int t = 0; t < kEvalCC; t++) {
                int ci = b + t;
                if (ci < j) {
                    uint32_t nA = 0, nC = 0, nG = 0, nT = 0;
                    for (int p = 0; p < k; p++) {
                        uint8_t m = codes[(size_t)ci * k + p];
                        if (!(m & 1)) nA |= 1u << p;
                        if (!(m & 2)) nC |= 1u << p;
                        if (!(m & 4)) nG |= 1u << p;
                        if (!(m & 8)) nT |= 1u << p;
                    }
                    cn.push_back(uint4{nA, nC, nG, nT});
                    co.push_back(ci);
                    for (int p = 0; p < k; p++)
                        symT[(items.size() - 1) * 32 + (size_t)p] |= (uint32_t)(codes[(size_t)ci * k + p] & 15u) << (4 * t);
                } else {
                    cn.push_back(uint4{kmask, kmask, kmask, kmask});
                    co.push_back(-1);
                }
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
    HIPCK(c, hipMemcpy(c->items, items.data(), sizeof(EvalItem) * items.size(), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(c->cand_n, cn.data(), sizeof(uint4) * cn.size(), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(c->cand_out, co.data(), sizeof(int32_t) * co.size(), hipMemcpyHostToDevice));
    return MP_OK;
}

int mp_eval_launch(mp_ctx *c, int64_t *device_out) {
    if (!c) return MP_ERR_ARG;
    if (!c->win) return fail(c, MP_ERR_ARG, "no windows built");
    if (!device_out) return fail(c, MP_ERR_ARG, "null output");
    HIPCK(c, hipSetDevice(c->dev));
    if (c->n_cand == 0) return MP_OK;
    HIPCK(c, hipMemsetAsync(device_out, 0, sizeof(int64_t) * 3 * (size_t)c->n_cand, c->stream));
    // enough blocks to fill 256 CUs several times over, each with at least 1024 sequences
    int max_split = (c->n_pad + 1023) / 1024;
    int want = (4096 + c->n_items - 1) / c->n_items;
    int split = std::max(1, std::min(max_split, want));
    int rows = ((c->n_pad + split - 1) / split + 1023) / 1024 * 1024;
    split = (c->n_pad + rows - 1) / rows;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!c->ev_free.empty()) { ev = c->ev_free.back(); c->ev_free.pop_back(); }
    else { HIPCK(c, hipEventCreate(&ev.first)); HIPCK(c, hipEventCreate(&ev.second)); }
    HIPCK(c, hipEventRecord(ev.first, c->stream));
    const char *mode_env = getenv("MP_EVAL_MODE");
    const bool bits = c->v <= 2 && !(mode_env && !strcmp(mode_env, "rows"));
    const int vmode = c->v == 0 ? 0 : (c->v == 1 ? 1 : 2);     // predicate specialisation of the row-per-lane code
    if (bits) {
        // bit-sliced pass over the column planes + row-per-lane pass over the patch / IUPAC lists
        // kernel shape (MP_EVAL_BITS): 0 = 8 candidates x 2 words (64 sequences) per thread (default),
        // 1 = the same with the next position's planes prefetched, 2 = 8 x 1 word
        int shape = 0;
        if (const char *e = getenv("MP_EVAL_BITS")) { shape = atoi(e); if (shape < 0 || shape > 2) shape = 0; }
        static const int shape_gw[3] = {2, 2, 1};
        const int nw = c->n_pad / 64;
        const int GW = shape_gw[shape];
        const int ny = std::max(1, (2 * nw / GW + kBlock - 1) / kBlock);
        const int ny_pad = (ny + 7) / 8 * 8;
        EvalBitsArgs ba{c->cols, c->excl, nw, c->p0, c->k, c->v, c->items, c->cand_symT, c->cand_out, c->sF, c->sR,
                        (unsigned long long *)device_out, ny, ny_pad};
#define BITS_ROW(LV) {eval_bits_kernel<8, LV, 2, false>, eval_bits_kernel<8, LV, 2, true>, eval_bits_kernel<8, LV, 1, false>}
        static const EvalBitsFn bfn[3][3] = {BITS_ROW(1), BITS_ROW(2), BITS_ROW(3)};
#undef BITS_ROW
        hipLaunchKernelGGL(bfn[c->v][shape], dim3((unsigned)((size_t)c->n_items * ny_pad)), dim3(kBlock), 0, c->stream, ba);
        if (c->n_patch || c->n_extra) {
            EvalListArgs la{c->items, c->cand_n, c->cand_out, c->n_patch ? c->patch_off : (const int32_t *)nullptr, c->patch_words,
                            c->n_extra ? c->extra_off : (const int32_t *)nullptr, c->extra_words, c->sF, c->sR, c->v,
                            (1u << c->k) - 1u, (unsigned long long *)device_out};
            static const EvalListFn lfn[3] = {eval_list_kernel<kEvalCC, 0>, eval_list_kernel<kEvalCC, 1>, eval_list_kernel<kEvalCC, 2>};
            int ly = std::max(1, std::min(64, (c->max_patch + 2047) / 2048));
            hipLaunchKernelGGL(lfn[vmode], dim3((unsigned)c->n_items, (unsigned)ly), dim3(kBlock), 0, c->stream, la);
        }
    } else {
    EvalArgs ea{c->win, c->n_pad, c->k, c->items, c->cand_n, c->cand_out, c->n_extra ? c->extra_off : (const int32_t *)nullptr,
                c->extra_words, c->sF, c->sR, c->v, (1u << c->k) - 1u, rows, (unsigned long long *)device_out};
    int variant = c->eval_variant;
    if (const char *e = getenv("MP_EVAL_VARIANT")) variant = atoi(e);
    if (variant < 0 || variant >= kNumEvalVariants) variant = 0;
    hipLaunchKernelGGL(kEvalVariants[variant].fn[c->p64 ? 1 : 0][getenv("MP_EVAL_GENERIC_V") ? 2 : vmode], dim3((unsigned)c->n_items, (unsigned)split),
                       dim3(kBlock), 0, c->stream, ea);
    }
    HIPCK(c, hipEventRecord(ev.second, c->stream));
    c->ev_busy.push_back(ev);
    HIPCK(c, hipGetLastError());
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
        c->ev_free.push_back(p);
    }
    c->ev_busy.clear();
    if (total_ms) *total_ms = c->ev_ms;
    if (n_launches) *n_launches = c->ev_n;
    if (reset) { c->ev_ms = 0; c->ev_n = 0; }
    return MP_OK;
}

int mp_eval_candidates(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint32_t sF, uint32_t sR,
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


int mp_eval_masks(mp_ctx *c, int32_t n_cand, const int32_t *cw, const uint8_t *codes, uint32_t sF, uint32_t sR,
                  uint64_t *not_f, uint64_t *not_r) {
    if (!c) return MP_ERR_ARG;
    int rc = mp_eval_upload(c, n_cand, cw, codes, sF, sR);
    if (rc) return rc;
    if (n_cand == 0) return MP_OK;
    if (!not_f || !not_r) return fail(c, MP_ERR_ARG, "null output");
    const size_t nw = (size_t)c->n_pad / 64, nwo = ((size_t)c->n_rows + 63) / 64;
    unsigned long long *d_f = nullptr, *d_r = nullptr;
    if ((rc = dev_alloc(c, &d_f, (size_t)n_cand * nw))) return rc;
    if ((rc = dev_alloc(c, &d_r, (size_t)n_cand * nw))) return rc;
    EvalArgs ea{c->win, c->n_pad, c->k, c->items, c->cand_n, c->cand_out, nullptr, nullptr, c->sF, c->sR, c->v,
                (1u << c->k) - 1u, 0, nullptr};
    const dim3 grid((unsigned)c->n_items, (unsigned)(c->n_pad / kBlock));
    if (c->p64) hipLaunchKernelGGL((mask_rows_kernel<kEvalCC, true>), grid, dim3(kBlock), 0, c->stream, ea, c->n_rows, d_f, d_r);
    else hipLaunchKernelGGL((mask_rows_kernel<kEvalCC, false>), grid, dim3(kBlock), 0, c->stream, ea, c->n_rows, d_f, d_r);
    HIPCK(c, hipGetLastError());
    // rows are padded to a multiple of 256 on the device: copy the (n_rows+63)/64 meaningful words of each mask
    HIPCK(c, hipMemcpy2DAsync(not_f, nwo * 8, d_f, nw * 8, nwo * 8, (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpy2DAsync(not_r, nwo * 8, d_r, nw * 8, nwo * 8, (size_t)n_cand, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    dev_free(c, &d_f, (size_t)n_cand * nw);
    dev_free(c, &d_r, (size_t)n_cand * nw);
    return MP_OK;
}

}  // extern "C"
