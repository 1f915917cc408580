// pack.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of
// include/mprime.h.  Alignment -> bit planes: pack, row scan, gap-free strings, column planes (mp_load_msa).
#include "common.hpp"
#include "../../include/mprime_host.h"

using namespace mp;

namespace {

// ----------------------------------------------------------------------------------------------
// (1) alignment -> planes
// ----------------------------------------------------------------------------------------------
__constant__ uint8_t c_code_lut[256];

// V20:453: upper-case, keep ACGTRYMKSWHBVD (as 4-bit base sets), everything else (N included) -> '-' = 0
static void host_code_lut(uint8_t *lut) {
    memset(lut, 0, 256);
    const char *sym = "ACGTRYMKSWHBVD";
    const uint8_t code[] = {1, 2, 4, 8, 5, 10, 3, 12, 6, 9, 11, 14, 7, 13};
    for (int i = 0; sym[i]; i++) {
        lut[(uint8_t)sym[i]] = code[i];
        lut[(uint8_t)(sym[i] + 32)] = code[i];
    }
}

// thread = (row, group of 4 chunks = 128 columns); lanes run along rows so that the plane stores coalesce.  A thread reads its 128
// residues as eight 16-byte loads (rows start at any byte: the hardware takes unaligned global loads) — whole 128-byte lines of a
// row per thread instead of 32 byte loads — and maps them through a copy of the symbol table in LDS.
constexpr int kPackChunks = 4;
__global__ __launch_bounds__(kBlock) void pack_kernel(const uint8_t *__restrict__ bytes, const int64_t *__restrict__ row_off,
                                                      int n_rows, int n_pad, int n_chunks, uint32_t *__restrict__ planes) {
    __shared__ uint8_t s_lut[256];
    s_lut[threadIdx.x] = c_code_lut[threadIdx.x];              // kBlock == 256
    __syncthreads();
    const int r = blockIdx.x * kBlock + threadIdx.x;
    const int c0 = blockIdx.y * kPackChunks;
    if (r >= n_pad) return;
    int64_t o = 0, len = 0;
    if (r < n_rows) { o = row_off[r]; len = row_off[r + 1] - o; }
    for (int q = 0; q < kPackChunks; q++) {
        const int c = c0 + q;
        if (c >= n_chunks) break;
        uint32_t mA = 0, mC = 0, mG = 0, mT = 0;
        const int64_t col0 = (int64_t)c * 32;
        const int n = (int)(len - col0 < 32 ? (len - col0 < 0 ? 0 : len - col0) : 32);
        if (n > 0) {
            uint32_t w[8];                                      // (the byte buffer ends 64 bytes after the last row: reading past a row's end is in bounds)
            __builtin_memcpy(w, bytes + o + col0, 32);
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const uint32_t code = j < n ? s_lut[(w[j >> 2] >> (8 * (j & 3))) & 255u] : 0u;
                mA |= (code & 1u) << j;
                mC |= ((code >> 1) & 1u) << j;
                mG |= ((code >> 2) & 1u) << j;
                mT |= ((code >> 3) & 1u) << j;
            }
        }
        const size_t base = ((size_t)c * 4) * n_pad + r;
        planes[base] = mA;
        planes[base + n_pad] = mC;
        planes[base + 2 * (size_t)n_pad] = mG;
        planes[base + 3 * (size_t)n_pad] = mT;
    }
}

// thread = (row, one of kScanSegs runs of chunks): prefix count of residues per chunk, leading-gap length and right-stripped length
// (V20:625-627), and the row's gap-free residue string `ung` (what `.replace("-", "")` gives, V20:673/679): residues are appended in a
// 64-bit nibble buffer and stored a word at a time into the word-major array, so a wave's stores of one word index coalesce.
// (Round 1 appended per (row, chunk) with atomicOr into a row-major array: 0.89 ms at 131072 x 1000; round 2: one thread per row,
// 0.145 ms — 2048 waves of a long dependent loop, two per SIMD.)  A segment first counts the residues in front of it (plane words only),
// then packs its own chunks from that nibble offset on; the words it shares with its neighbours (the first and the last it touches)
// are ORed into the zero-filled array, the others stored.  lead / rstrip: the segment that sees the row's first residue writes `lead`,
// every segment with a residue raises `rstrip` (zero-filled) with atomicMax.
constexpr int kScanSegs = 4;
__global__ __launch_bounds__(kBlock) void row_scan_kernel(const uint32_t *__restrict__ planes, const int64_t *__restrict__ row_off,
                                                          int n_rows, int n_pad, int n_chunks, uint32_t *__restrict__ cum,
                                                          int32_t *__restrict__ lead, int32_t *__restrict__ rstrip,
                                                          int32_t *__restrict__ rlen, uint32_t *__restrict__ ung) {
    const int r = blockIdx.x * kBlock + threadIdx.x, seg = blockIdx.y;
    if (r >= n_pad) return;
    const size_t np = (size_t)n_pad;
    const int per = (n_chunks + kScanSegs - 1) / kScanSegs;
    const int c0 = min(n_chunks, seg * per), c1 = min(n_chunks, c0 + per);
    auto residues = [&](int c) {
        const size_t base = ((size_t)c * 4) * np + r;
        return planes[base] | planes[base + np] | planes[base + 2 * np] | planes[base + 3 * np];
    };
    uint32_t run = 0;
    for (int c = 0; c < c0; c++) run += __popc(residues(c));
    const uint32_t before = run;
    int first = -1, last = 0;
    unsigned long long buf = 0;
    int nb = (int)(before & 7u);                          // the segment's first word starts at this nibble
    size_t widx = before >> 3;
    bool shared = true;                                   // the next word to leave is the segment's first: a neighbour may hold part of it
    for (int c = c0; c < c1; c++) {
        const size_t base = ((size_t)c * 4) * np + r;
        const uint32_t mA = planes[base], mC = planes[base + np], mG = planes[base + 2 * np], mT = planes[base + 3 * np];
        uint32_t ng = mA | mC | mG | mT;
        cum[(size_t)c * np + r] = run;
        run += __popc(ng);
        if (ng) {
            if (first < 0) first = c * 32 + (__ffs(ng) - 1);
            last = c * 32 + 32 - __clz(ng);
        }
        while (ng) {
            const int j = __ffs(ng) - 1;
            ng &= ng - 1;
            const unsigned long long code = ((mA >> j) & 1u) | (((mC >> j) & 1u) << 1) | (((mG >> j) & 1u) << 2) | (((mT >> j) & 1u) << 3);
            buf |= code << (4 * nb);
            if (++nb == 16) {
                if (shared) atomicOr(&ung[widx * np + r], (uint32_t)buf);
                else ung[widx * np + r] = (uint32_t)buf;
                ung[(widx + 1) * np + r] = (uint32_t)(buf >> 32);
                shared = false;
                widx += 2; buf = 0; nb = 0;
            }
        }
    }
    // the tail: up to two words, the last of which the next segment may continue
    if (nb > 0) atomicOr(&ung[widx * np + r], (uint32_t)buf);
    if (nb > 8) atomicOr(&ung[(widx + 1) * np + r], (uint32_t)(buf >> 32));
    if (c1 == n_chunks && c0 < c1) cum[(size_t)n_chunks * np + r] = run;
    if (r < n_rows) {
        const int len = (int)(row_off[r + 1] - row_off[r]);
        if (seg == 0) rlen[r] = len;
        if (before == 0 && first >= 0) lead[r] = first;                               // nothing in front of this segment's first residue
        if (c1 == n_chunks && c0 < c1 && run == 0) lead[r] = len;                     // a row without residues
        if (first >= 0) atomicMax(&rstrip[r], last);
    }
}

// Column planes for the bit-sliced evaluation: cols[col][4][Npad/64] u64, bit r%64 of word r/64 =
// sequence r; one plane per base (A, C, G, T), all four clear where the sequence has a gap or has
// ended.  A concrete candidate symbol then needs ONE plane per position ("matches" = that plane), a
// degenerate one the OR of its bases' planes.  An IUPAC residue sets none either: windows that touch
// one are always routed to the general path (patch list), never to the bit-sliced pass.
constexpr int kColBlock = 1024;           // 16 waves = 1024 rows per workgroup: a (column, base) row of 16 words leaves as one 128-byte line
__global__ __launch_bounds__(kColBlock) void colplane_kernel(const uint32_t *__restrict__ planes, int n_pad, int /*n_chunks*/,
                                                             unsigned long long *__restrict__ cols) {
    __shared__ unsigned long long s_tile[32 * 4][kColBlock / 64];
    const int r = blockIdx.x * kColBlock + threadIdx.x;
    const int c = blockIdx.y;
    const size_t np = (size_t)n_pad, nw = np / 64;
    uint32_t mA = 0, mC = 0, mG = 0, mT = 0;
    if (r < n_pad) {
        const size_t base = ((size_t)c * 4) * np + r;
        mA = planes[base]; mC = planes[base + np]; mG = planes[base + 2 * np]; mT = planes[base + 3 * np];
        // an IUPAC cell (several bases) leaves NO plane set, like a gap: every consumer keeps windows that touch one away from these
        // planes anyway (patch list), and the sliding evaluation's arithmetic wants at most one base per cell (evalslide.hip)
        const uint32_t multi = (mA & (mC | mG | mT)) | (mC & (mG | mT)) | (mG & mT);
        mA &= ~multi; mC &= ~multi; mG &= ~multi; mT &= ~multi;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane j of a wave ends up holding the four words of column j (the ballots over the wave's 64 rows)
    unsigned long long kA = 0, kC = 0, kG = 0, kT = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const unsigned long long xA = __ballot((mA >> j) & 1u), xC = __ballot((mC >> j) & 1u);
        const unsigned long long xG = __ballot((mG >> j) & 1u), xT = __ballot((mT >> j) & 1u);
        if (lane == j) { kA = xA; kC = xC; kG = xG; kT = xT; }
    }
    if (lane < 32) {
        s_tile[lane * 4 + 0][wave] = kA; s_tile[lane * 4 + 1][wave] = kC;
        s_tile[lane * 4 + 2][wave] = kG; s_tile[lane * 4 + 3][wave] = kT;
    }
    __syncthreads();
    // 128 (column, base) rows x 16 words: 8 threads per row store 16 bytes each
    const int row = threadIdx.x >> 3, w2 = (threadIdx.x & 7) * 2;
    const size_t word = (size_t)blockIdx.x * (kColBlock / 64) + w2;
    if (word < nw) {                                             // nw is a multiple of 4 (n_pad of 256): pairs never straddle the end
        unsigned long long *dst = cols + ((size_t)(c * 32 + (row >> 2)) * 4 + (row & 3)) * nw + word;
        dst[0] = s_tile[row][w2];
        dst[1] = s_tile[row][w2 + 1];
    }
}



// histograms of two per-row integers.  Most rows share a value (0 leading gaps, the alignment width), and a device atomic per row on
// one address would serialise the launch: equal values inside a wave are added once, a workgroup keeps the sums of the values it
// meets in a small direct-mapped LDS table over its whole share of the rows (a value whose slot is taken goes to HBM directly), and
// only the table leaves through device atomics — a handful per workgroup.
constexpr int kHistCache = 256;
__global__ __launch_bounds__(kBlock) void row_hist_kernel(const int32_t *__restrict__ lead, const int32_t *__restrict__ rstrip, int n_rows,
                                                          int n_bins, unsigned long long *__restrict__ hist, int *__restrict__ overflow) {
    __shared__ int s_tag[2][kHistCache];
    __shared__ unsigned int s_cnt[2][kHistCache];
    for (int i = threadIdx.x; i < 2 * kHistCache; i += kBlock) { (&s_tag[0][0])[i] = -1; (&s_cnt[0][0])[i] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int rounds = (n_rows + (int)(gridDim.x * kBlock) - 1) / (int)(gridDim.x * kBlock);       // the same trip count for every wave
    for (int it = 0; it < rounds; it++) {
        const int r = (it * (int)gridDim.x + (int)blockIdx.x) * kBlock + (int)threadIdx.x;
        const bool active = r < n_rows;
        for (int which = 0; which < 2; which++) {
            const int v = active ? (which ? rstrip[r] : lead[r]) : -1;
            unsigned long long todo = __ballot(active);
            while (todo) {
                const int first = __ffsll((long long)todo) - 1;
                const int v0 = __shfl(v, first);
                const unsigned long long same = __ballot(active && v == v0);
                if (lane == first) {
                    const unsigned int n = (unsigned int)__popcll(same);
                    if (v0 < 0 || v0 >= n_bins) atomicExch(overflow, 1);
                    else {
                        const int slot = v0 & (kHistCache - 1);
                        const int seen = atomicCAS(&s_tag[which][slot], -1, v0);
                        if (seen == -1 || seen == v0) atomicAdd(&s_cnt[which][slot], n);
                        else atomicAdd(&hist[(size_t)which * n_bins + v0], (unsigned long long)n);
                    }
                }
                todo &= ~same;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * kHistCache; i += kBlock) {
        const int which = i / kHistCache, tag = (&s_tag[0][0])[i];
        const unsigned int n = (&s_cnt[0][0])[i];
        if (tag >= 0 && n) atomicAdd(&hist[(size_t)which * n_bins + tag], (unsigned long long)n);
    }
}
}  // namespace

namespace mp {
int pack_init() {
    uint8_t lut[256];
    host_code_lut(lut);
    return hipMemcpyToSymbol(HIP_SYMBOL(c_code_lut), lut, 256) == hipSuccess ? MP_OK : MP_ERR_DEVICE;
}
}  // namespace mp

extern "C" {

int mp_reserve_columns(mp_ctx *c, int32_t n_columns) {
    if (!c) return MP_ERR_ARG;
    if (n_columns < 0 || n_columns > 0x3fffffff) return fail(c, MP_ERR_ARG, "mp_reserve_columns: bad width %d", n_columns);
    c->reserve_cols = n_columns;
    return MP_OK;
}

}  // extern "C"

namespace {
// the device side of a load: allocations, the residue bytes (brought by `upload`), the packing kernels
template <typename Upload>
int load_msa_impl(mp_ctx *c, const int64_t *row_off, int32_t n_rows, Upload &&upload) {
    HIPCK(c, hipSetDevice(c->dev));
    free_msa(c);
    int64_t max_len = 0;
    for (int r = 0; r < n_rows; r++) {
        int64_t l = row_off[r + 1] - row_off[r];
        if (l < 0 || l > 0x3fffffff) return fail(c, MP_ERR_ARG, "row %d has bad length", r);
        max_len = std::max(max_len, l);
    }
    c->n_rows = n_rows;
    c->n_pad = (n_rows + kBlock - 1) / kBlock * kBlock;
    max_len = std::max<int64_t>(max_len, c->reserve_cols);      // row shards: the alignment is wider than the local rows
    c->max_len = (int)max_len;
    c->n_chunks = (int)((max_len + 31) / 32) + 2;
    c->ustride = (int)(max_len / 8) + 3;
    size_t np = (size_t)c->n_pad;
    int64_t total = row_off[n_rows] - row_off[0];
    uint8_t *d_bytes = nullptr;
    int64_t *d_off = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &d_bytes, (size_t)total + 64))) return rc;
    if ((rc = dev_alloc(c, &d_off, (size_t)n_rows + 1))) return rc;
    if ((rc = dev_alloc(c, &c->planes, (size_t)c->n_chunks * 4 * np))) return rc;
    // + one all-zero plane row behind the last column's: the sliding evaluation points plane-less fetches at it (evalslide.hip)
    if ((rc = dev_alloc(c, &c->cols, ((size_t)c->n_chunks * 32 * 4 + 1) * (np / 64)))) return rc;
    if ((rc = dev_alloc(c, &c->cum, ((size_t)c->n_chunks + 1) * np))) return rc;
    if ((rc = dev_alloc(c, &c->ung, np * c->ustride))) return rc;
    if ((rc = dev_alloc(c, &c->lead, np))) return rc;
    if ((rc = dev_alloc(c, &c->rstrip, np))) return rc;
    if ((rc = dev_alloc(c, &c->rlen, np))) return rc;
    std::vector<int64_t> off0(n_rows + 1);
    for (int r = 0; r <= n_rows; r++) off0[r] = row_off[r] - row_off[0];
    HIPCK(c, hipMemcpyAsync(d_off, off0.data(), sizeof(int64_t) * (n_rows + 1), hipMemcpyHostToDevice, c->stream));
    const FillSeg init[4] = {{c->ung, sizeof(uint32_t) * np * c->ustride, 0u}, {c->rlen, sizeof(int32_t) * np, 0u}, {c->rstrip, sizeof(int32_t) * np, 0u},
                             {c->cols + (size_t)c->n_chunks * 32 * 4 * (np / 64), sizeof(unsigned long long) * (np / 64), 0u}};
    if ((rc = fill_segments(c, init, 4))) return rc;
    if ((rc = upload(d_bytes, total))) return rc;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)(c->n_pad / kBlock), (unsigned)((c->n_chunks + kPackChunks - 1) / kPackChunks)), dim3(kBlock), 0,
                       c->stream, d_bytes, d_off, n_rows, c->n_pad, c->n_chunks, c->planes);
    hipLaunchKernelGGL(row_scan_kernel, dim3((unsigned)(c->n_pad / kBlock), (unsigned)kScanSegs), dim3(kBlock), 0, c->stream, c->planes, d_off, n_rows,
                       c->n_pad, c->n_chunks, c->cum, c->lead, c->rstrip, c->rlen, c->ung);
    hipLaunchKernelGGL(colplane_kernel, dim3((unsigned)((c->n_pad + kColBlock - 1) / kColBlock), (unsigned)c->n_chunks), dim3(kColBlock), 0, c->stream,
                       c->planes, c->n_pad, c->n_chunks, c->cols);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipStreamSynchronize(c->stream));
    dev_free(c, &d_bytes, (size_t)total + 64);
    dev_free(c, &d_off, (size_t)n_rows + 1);
    return MP_OK;
}
}  // namespace

extern "C" {

int mp_load_msa(mp_ctx *c, const uint8_t *bytes, const int64_t *row_off, int32_t n_rows) {
    if (!c) return MP_ERR_ARG;
    if (!bytes || !row_off || n_rows <= 0) return fail(c, MP_ERR_ARG, "mp_load_msa: bad arguments");
    return load_msa_impl(c, row_off, n_rows, [&](uint8_t *d_bytes, int64_t total) -> int {
        // MP_EXPERIMENT_PIN_LOAD (tools/load_stress.py only): round 4's registration of the CALLER's residue bytes for the duration of this
        // transfer — withdrawn after an unexplained SIGABRT in one GPU-suite run of four; kept behind the switch so that the stress tool
        // exercises exactly that code path
        struct PinForLoad {
            void *p = nullptr;
            PinForLoad(void *ptr, size_t n) { if (getenv("MP_EXPERIMENT_PIN_LOAD") && n >= ((size_t)4 << 20) && hipHostRegister(ptr, n, hipHostRegisterDefault) == hipSuccess) p = ptr; else (void)hipGetLastError(); }
            ~PinForLoad() { if (p) { (void)hipStreamSynchronize(nullptr); (void)hipHostUnregister(p); } }
        } pin_for_load(getenv("MP_EXPERIMENT_PIN_LOAD") ? const_cast<uint8_t *>(bytes) + row_off[0] : nullptr, (size_t)total);
        HIPCK(c, hipMemcpyAsync(d_bytes, bytes + row_off[0], (size_t)total, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));          // (a registration ends with this scope)
        return MP_OK;
    });
}

// mprime_host.h: the residue bytes straight from the parsed file through the context's ring of registered transfer buffers
int mp_load_msa_fasta(mp_ctx *c, const mp_fasta *f) {
    if (!c) return MP_ERR_ARG;
    if (!f) return fail(c, MP_ERR_ARG, "mp_load_msa_fasta: no parsed file");
    int32_t n_rows = 0;
    int64_t n_bytes = 0, n_ids = 0;
    if (mp_fasta_sizes(f, &n_rows, &n_bytes, &n_ids) != MP_OK || n_rows <= 0) return fail(c, MP_ERR_ARG, "mp_load_msa_fasta: no sequence records");
    std::vector<int64_t> row_off((size_t)n_rows + 1);
    if (mp_fasta_rows(f, nullptr, row_off.data()) != MP_OK) return fail(c, MP_ERR_ARG, "mp_load_msa_fasta: mp_fasta_rows");
    constexpr size_t kSlot = (size_t)32 << 20;
    constexpr int kSlots = 3;
    return load_msa_impl(c, row_off.data(), n_rows, [&](uint8_t *d_bytes, int64_t total) -> int {
        HIPCK(c, hipSetDevice(c->dev));
        if (!c->h_ring) {
            c->h_ring = static_cast<uint8_t *>(host_map(kSlot * kSlots));
            if (!c->h_ring) return fail(c, MP_ERR_NOMEM, "mp_load_msa_fasta: out of host memory");
            prefault_host(c->h_ring, kSlot * kSlots);
            if (!getenv("MP_NO_PIN") && hipHostRegister(c->h_ring, kSlot * kSlots, hipHostRegisterDefault) == hipSuccess) c->h_ring_pinned = true;
            else (void)hipGetLastError();
            for (int i = 0; i < kSlots; i++) HIPCK(c, hipEventCreateWithFlags(&c->h_ring_ev[i], hipEventDisableTiming));
        }
        int slot = 0;
        bool used[kSlots] = {false, false, false};
        for (int64_t at = 0; at < total; at += (int64_t)kSlot, slot = (slot + 1) % kSlots) {
            const int64_t n = std::min<int64_t>((int64_t)kSlot, total - at);
            if (used[slot]) HIPCK(c, hipEventSynchronize(c->h_ring_ev[slot]));       // the copy that last read this slot has finished
            uint8_t *buf = c->h_ring + (size_t)slot * kSlot;
            if (mp_fasta_gather(f, at, at + n, buf, 0) != MP_OK) return fail(c, MP_ERR_ARG, "mp_load_msa_fasta: gather");
            HIPCK(c, hipMemcpyAsync(d_bytes + at, buf, (size_t)n, hipMemcpyHostToDevice, c->stream));
            HIPCK(c, hipEventRecord(c->h_ring_ev[slot], c->stream));
            used[slot] = true;
        }
        return MP_OK;
    });
}

int mp_row_attributes(mp_ctx *c, int32_t *lead, int32_t *rstrip, int32_t *rowlen) {
    if (!c) return MP_ERR_ARG;
    if (!c->planes) return fail(c, MP_ERR_ARG, "no alignment loaded");
    HIPCK(c, hipSetDevice(c->dev));
    size_t n = sizeof(int32_t) * (size_t)c->n_rows;
    if (lead) HIPCK(c, hipMemcpyAsync(lead, c->lead, n, hipMemcpyDeviceToHost, c->stream));
    if (rstrip) HIPCK(c, hipMemcpyAsync(rstrip, c->rstrip, n, hipMemcpyDeviceToHost, c->stream));
    if (rowlen) HIPCK(c, hipMemcpyAsync(rowlen, c->rlen, n, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return MP_OK;
}

int mp_row_histograms(mp_ctx *c, int32_t n_bins, int64_t *lead_hist, int64_t *rstrip_hist) {
    if (!c) return MP_ERR_ARG;
    if (!c->planes) return fail(c, MP_ERR_ARG, "no alignment loaded");
    if (n_bins <= 0 || !lead_hist || !rstrip_hist) return fail(c, MP_ERR_ARG, "mp_row_histograms: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    unsigned long long *d_hist = nullptr;
    int *d_over = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &d_hist, (size_t)2 * n_bins))) return rc;
    if ((rc = dev_alloc(c, &d_over, 1))) { dev_free(c, &d_hist, (size_t)2 * n_bins); return rc; }
    std::vector<unsigned long long> h((size_t)2 * n_bins);
    int over = 0;
    hipError_t e = hipMemsetAsync(d_hist, 0, sizeof(unsigned long long) * 2 * (size_t)n_bins, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_over, 0, sizeof(int), c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(row_hist_kernel, dim3((unsigned)std::min(c->n_pad / kBlock, 256)), dim3(kBlock), 0, c->stream, (const int32_t *)c->lead,
                           (const int32_t *)c->rstrip, c->n_rows, (int)n_bins, d_hist, d_over);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_hist, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&over, d_over, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(c, &d_hist, (size_t)2 * n_bins);
    dev_free(c, &d_over, 1);
    if (e != hipSuccess) return fail(c, MP_ERR_DEVICE, "mp_row_histograms: %s", hipGetErrorString(e));
    if (over) return fail(c, MP_ERR_CAPACITY, "mp_row_histograms: a row is longer than %d", n_bins - 1);
    for (int i = 0; i < n_bins; i++) { lead_hist[i] = (int64_t)h[(size_t)i]; rstrip_hist[i] = (int64_t)h[(size_t)n_bins + i]; }
    return MP_OK;
}

}  // extern "C"
