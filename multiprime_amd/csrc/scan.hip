// scan.hip — part of libmprime_hip.so: hand-written HIP (gfx950 / MI355X, wave64) behind the C ABI of include/mprime.h.
// (8) k-mismatch primer-site scan over unaligned sequences (mp_kmm_scan) — the GPU replacement of the bowtie2 + samtools
// mapping step of scripts/primer_coverage_validation_by_BWT_V9.py (BWT:264-300) and of its MD:Z filter (BWT:241-262).
//
// Shape: a workgroup owns one segment of one sequence.  It packs the segment once into LDS — 2 bits per base plus a
// "never matches" bit for everything outside ACGT — and every thread then slides over its start positions: the 32-base
// window at a position is two LDS words and a funnel shift (a second 32-base window for patterns of 33..64 bases: the
// two-word kernel), and one pattern costs an XOR, a fold of the two bits of each base, a mask, a popcount and two compares
// per word.  The pattern table (code word, length mask, trailing-run mask per strand) is
// wave-uniform and arrives through scalar loads.  HBM traffic is the text once (1 byte per base); the work is integer VALU:
// ~10 wave-instructions per (64 positions, pattern).
#include "common.hpp"

using namespace mp;

namespace {

constexpr int kSeg = 8192;                       // start positions per workgroup

// NW = 64-bit words of a pattern: 1 up to 32 bases, 2 up to MP_PATTERN_MAX_LEN = 64 (adaptor-tailed primers)
template <int NW>
struct KmmPat {
    unsigned long long word[NW];      // base j of the aligned text at bits 2 (j % 32) of word j / 32 (A0 C1 G2 T3)
    unsigned long long lenmask[NW];   // bit 2j set for j < len
    unsigned long long termmask[NW];  // bit 2j set for the last `term` positions in reference orientation (all of lenmask if term > len)
    int32_t len, id, strand, never;   // never: term > len — the trailing match run cannot reach the threshold
};

// RES: the text comes from the context's resident store (mp_seq_load: `code` / `flag` words at word offsets `woff`) — a segment is
// kSegWords coalesced 8-byte loads per plane instead of 32 byte loads and ~130 instructions per word
template <int NW, bool RES>
__global__ __launch_bounds__(kBlock) void kmm_kernel(const uint8_t *__restrict__ bytes, const int64_t *__restrict__ row_off,
                                                     const unsigned long long *__restrict__ code, const unsigned long long *__restrict__ flag,
                                                     const int64_t *__restrict__ woff,
                                                     const int32_t *__restrict__ blk_row, const int32_t *__restrict__ blk_seg,
                                                     const KmmPat<NW> *__restrict__ pats, int n_pats, int max_mm, long long cap,
                                                     int32_t *__restrict__ hits, unsigned long long *__restrict__ n_hits) {
    constexpr int kSegWords = kSeg / 32 + 1 + NW;      // 64-bit words of 32 bases, with the overhang of the longest pattern
    __shared__ unsigned long long s_b[kSegWords];      // 2-bit codes
    __shared__ unsigned long long s_n[kSegWords];      // 0b01 at positions that match nothing (non-ACGT, past the end)
    const int row = blk_row[blockIdx.x], seg = blk_seg[blockIdx.x];
    const uint8_t *s = bytes + row_off[row];
    const long long len = row_off[row + 1] - row_off[row];
    const long long base = (long long)seg * kSeg;
    // pack: thread t builds word t (32 bases)
    for (int w = threadIdx.x; w < kSegWords; w += kBlock) {
        unsigned long long b = 0, n = 0;
        const long long p0 = base + (long long)w * 32;
        if (RES) {
            const long long gw = base / 32 + w, nwords = woff[row + 1] - woff[row];
            if (gw < nwords) { b = code[woff[row] + gw]; n = flag[woff[row] + gw] & 0x5555555555555555ull; }     // (the scan upper-cases: bit 2j alone)
            else n = 0x5555555555555555ull;
            s_b[w] = b; s_n[w] = n;
            continue;
        }
        for (int j = 0; j < 32; j++) {
            const long long p = p0 + j;
            unsigned long long code = 0, bad = 1;
            if (p < len) {
                uint8_t ch = s[p];
                if (ch >= 'a' && ch <= 'z') ch -= 32;
                if (ch == 'A') { code = 0; bad = 0; }
                else if (ch == 'C') { code = 1; bad = 0; }
                else if (ch == 'G') { code = 2; bad = 0; }
                else if (ch == 'T') { code = 3; bad = 0; }
            }
            b |= code << (2 * j);
            n |= bad << (2 * j);
        }
        s_b[w] = b;
        s_n[w] = n;
    }
    __syncthreads();
    const unsigned long long kOdd = 0x5555555555555555ull;
    for (int q = threadIdx.x; q < kSeg; q += kBlock) {
        const long long p = base + q;
        if (p >= len) break;
        const int w = q >> 5, sh = (q & 31) * 2;
        unsigned long long win[NW], nw[NW];
#pragma unroll
        for (int t = 0; t < NW; t++) {
            win[t] = s_b[w + t] >> sh; nw[t] = s_n[w + t] >> sh;
            if (sh) { win[t] |= s_b[w + t + 1] << (64 - sh); nw[t] |= s_n[w + t + 1] << (64 - sh); }
        }
        for (int i = 0; i < n_pats; i++) {
            const KmmPat<NW> P = pats[i];               // uniform index: scalar loads
            int n_mm = 0;
            unsigned long long in_term = 0;
#pragma unroll
            for (int t = 0; t < NW; t++) {
                const unsigned long long x = win[t] ^ P.word[t];
                const unsigned long long mm = (((x | (x >> 1)) & kOdd) | nw[t]) & P.lenmask[t];
                n_mm += (int)__popcll(mm);
                in_term |= mm & P.termmask[t];
            }
            if (n_mm <= max_mm && in_term == 0 && !P.never && p + P.len <= len) {
                const unsigned long long idx = atomicAdd(n_hits, 1ull);
                if ((long long)idx < cap) {
                    hits[4 * idx] = row; hits[4 * idx + 1] = (int32_t)p; hits[4 * idx + 2] = P.id; hits[4 * idx + 3] = P.strand;
                }
            }
        }
    }
}

// both strands of every pattern as kernel table entries
template <int NW>
void kmm_patterns(int32_t n_pat, const uint8_t *pat_codes, const int32_t *pat_off, int32_t term, std::vector<KmmPat<NW>> &pats) {
    for (int32_t i = 0; i < n_pat; i++) {
        const int len = pat_off[i + 1] - pat_off[i];
        int b[MP_PATTERN_MAX_LEN];
        for (int j = 0; j < len; j++) {
            const uint8_t m = pat_codes[pat_off[i] + j];
            b[j] = m == 1 ? 0 : m == 2 ? 1 : m == 4 ? 2 : 3;
        }
        for (int strand = 0; strand < 2; strand++) {
            KmmPat<NW> P{};
            for (int j = 0; j < len; j++) {
                const int code = strand == 0 ? b[j] : 3 - b[len - 1 - j];      // the text reads the pattern / its reverse complement
                P.word[j >> 5] |= (unsigned long long)code << (2 * (j & 31));
                P.lenmask[j >> 5] |= 1ull << (2 * (j & 31));
                if (j >= len - term) P.termmask[j >> 5] |= 1ull << (2 * (j & 31));
            }
            P.len = len; P.id = i; P.strand = strand; P.never = term > len;
            pats.push_back(P);
        }
    }
}

// pack the characters of the store into words (one thread per word; once per mp_seq_load)
__global__ __launch_bounds__(kBlock) void seq_pack_kernel(const uint8_t *__restrict__ bytes, const int64_t *__restrict__ row_off,
                                                          const int64_t *__restrict__ woff, int n_rows, unsigned long long *__restrict__ code,
                                                          unsigned long long *__restrict__ flag) {
    const int row = blockIdx.x;
    if (row >= n_rows) return;
    const uint8_t *s = bytes + row_off[row];
    const long long len = row_off[row + 1] - row_off[row], nw = woff[row + 1] - woff[row];
    for (long long w = threadIdx.x; w < nw; w += kBlock) {
        unsigned long long b = 0, f = 0;
        for (int j = 0; j < 32; j++) {
            const long long p = w * 32 + j;
            unsigned long long c = 0, bad = 1, low = 0;
            if (p < len) {
                uint8_t ch = s[p];
                if (ch >= 'a' && ch <= 'z') { ch -= 32; low = 1; }
                if (ch == 'A') { c = 0; bad = 0; }
                else if (ch == 'C') { c = 1; bad = 0; }
                else if (ch == 'G') { c = 2; bad = 0; }
                else if (ch == 'T') { c = 3; bad = 0; }
            }
            b |= c << (2 * j);
            f |= (bad | (low << 1)) << (2 * j);
        }
        code[woff[row] + w] = b;
        flag[woff[row] + w] = f;
    }
}

// the scan itself on device text (bytes of this call, or the resident store when `code` is set)
int kmm_scan_device(mp_ctx *c, const uint8_t *d_bytes, const int64_t *d_roff, const unsigned long long *code, const unsigned long long *flag,
                    const int64_t *d_woff, const int64_t *roff_host, int32_t n_rows, int32_t n_pat, const uint8_t *pat_codes, const int32_t *pat_off,
                    int32_t max_mm, int32_t term, int64_t cap, int32_t *hits, int64_t *n_hits) {
    int longest = 0;
    for (int32_t i = 0; i < n_pat; i++) {
        const int len = pat_off[i + 1] - pat_off[i];
        if (len < 4 || len > MP_PATTERN_MAX_LEN) return fail(c, MP_ERR_ARG, "pattern %d has length %d (4..%d supported)", i, len, MP_PATTERN_MAX_LEN);
        for (int j = 0; j < len; j++) {
            const uint8_t m = pat_codes[pat_off[i] + j];
            if (m != 1 && m != 2 && m != 4 && m != 8) return fail(c, MP_ERR_ARG, "pattern %d is not a concrete A/C/G/T sequence", i);
        }
        longest = std::max(longest, len);
    }
    const bool two_words = longest > 32;             // one pattern longer than 32 bases: the two-word kernel for the whole call
    std::vector<KmmPat<1>> pats1;
    std::vector<KmmPat<2>> pats2;
    if (two_words) kmm_patterns<2>(n_pat, pat_codes, pat_off, term, pats2);
    else kmm_patterns<1>(n_pat, pat_codes, pat_off, term, pats1);
    const size_t n_entries = two_words ? pats2.size() : pats1.size();
    const size_t pat_bytes = two_words ? sizeof(KmmPat<2>) * pats2.size() : sizeof(KmmPat<1>) * pats1.size();
    const void *pat_src = two_words ? (const void *)pats2.data() : (const void *)pats1.data();
    std::vector<int32_t> blk_row, blk_seg;
    for (int32_t r = 0; r < n_rows; r++) {
        const int64_t len = roff_host[r + 1] - roff_host[r];
        if (len < 0 || len > 0x7fffffffLL) return fail(c, MP_ERR_ARG, "sequence %d has bad length", r);
        for (int64_t sgm = 0; sgm * kSeg < len; sgm++) { blk_row.push_back(r); blk_seg.push_back((int32_t)sgm); }
    }
    if (blk_row.empty()) return MP_OK;
    const size_t nb = blk_row.size();
    int32_t *d_brow = nullptr, *d_bseg = nullptr, *d_hits = nullptr;
    uint8_t *d_pats = nullptr;
    unsigned long long *d_n = nullptr;
    const size_t hcap = (size_t)std::max<int64_t>(cap, 1) * 4;
    int rc = MP_OK;
    auto cleanup = [&]() {
        dev_free(c, &d_brow, nb); dev_free(c, &d_bseg, nb); dev_free(c, &d_hits, hcap); dev_free(c, &d_pats, pat_bytes); dev_free(c, &d_n, 1);
    };
    hipError_t e = hipSuccess;
    if ((rc = dev_alloc(c, &d_brow, nb)) || (rc = dev_alloc(c, &d_bseg, nb)) || (rc = dev_alloc(c, &d_hits, hcap)) ||
        (rc = dev_alloc(c, &d_pats, pat_bytes)) || (rc = dev_alloc(c, &d_n, 1))) { cleanup(); return rc; }
    if (e == hipSuccess) e = hipMemcpyAsync(d_brow, blk_row.data(), sizeof(int32_t) * nb, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_bseg, blk_seg.data(), sizeof(int32_t) * nb, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pats, pat_src, pat_bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_n, 0, sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) {
#define MP_KMM_LAUNCH(NW, RES)                                                                                                               \
    hipLaunchKernelGGL((kmm_kernel<NW, RES>), dim3((unsigned)nb), dim3(kBlock), 0, c->stream, d_bytes, d_roff, code, flag, d_woff, d_brow, \
                       d_bseg, reinterpret_cast<const KmmPat<NW> *>(d_pats), (int)n_entries, (int)max_mm, (long long)cap, d_hits, d_n)
        if (code) { if (two_words) MP_KMM_LAUNCH(2, true); else MP_KMM_LAUNCH(1, true); }
        else { if (two_words) MP_KMM_LAUNCH(2, false); else MP_KMM_LAUNCH(1, false); }
#undef MP_KMM_LAUNCH
        e = hipGetLastError();
    }
    unsigned long long n = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&n, d_n, sizeof n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && cap) {
        const size_t got = (size_t)std::min<unsigned long long>(n, (unsigned long long)cap);
        if (got) e = hipMemcpyAsync(hits, d_hits, sizeof(int32_t) * 4 * got, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    cleanup();
    if (e != hipSuccess) return fail(c, MP_ERR_DEVICE, "mp_kmm_scan: %s", hipGetErrorString(e));
    *n_hits = (int64_t)n;
    return MP_OK;
}

}  // namespace

namespace mp {
void free_seq(mp_ctx *c) {
    dev_free(c, &c->sq_bytes, c->sq_total + 16); dev_free(c, &c->sq_roff, (size_t)c->sq_n + 1);
    dev_free(c, &c->sq_code, c->sq_words); dev_free(c, &c->sq_flag, c->sq_words); dev_free(c, &c->sq_woff, (size_t)c->sq_n + 1);
    c->sq_n = 0; c->sq_total = c->sq_words = 0;
    c->sq_roff_host.clear();
}
}  // namespace mp

extern "C" {

int mp_seq_load(mp_ctx *c, const uint8_t *bytes, const int64_t *row_off, int32_t n_rows) {
    if (!c) return MP_ERR_ARG;
    if (n_rows < 0 || (n_rows && (!bytes || !row_off))) return fail(c, MP_ERR_ARG, "mp_seq_load: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    free_seq(c);
    if (n_rows == 0) return MP_OK;
    std::vector<int64_t> roff((size_t)n_rows + 1), woff((size_t)n_rows + 1);
    woff[0] = 0;
    for (int32_t r = 0; r <= n_rows; r++) roff[(size_t)r] = row_off[r] - row_off[0];
    for (int32_t r = 0; r < n_rows; r++) {
        const int64_t len = roff[(size_t)r + 1] - roff[(size_t)r];
        if (len < 0 || len > 0x7fffffffLL) return fail(c, MP_ERR_ARG, "sequence %d has bad length", r);
        woff[(size_t)r + 1] = woff[(size_t)r] + (len + 31) / 32;
    }
    const size_t total = (size_t)roff[(size_t)n_rows], words = (size_t)woff[(size_t)n_rows];
    int rc;
    c->sq_n = n_rows; c->sq_total = total; c->sq_words = words;
    if ((rc = dev_alloc(c, &c->sq_bytes, total + 16)) || (rc = dev_alloc(c, &c->sq_roff, (size_t)n_rows + 1)) || (rc = dev_alloc(c, &c->sq_code, words)) ||
        (rc = dev_alloc(c, &c->sq_flag, words)) || (rc = dev_alloc(c, &c->sq_woff, (size_t)n_rows + 1))) { free_seq(c); return rc; }
    hipError_t e = hipSuccess;
    if (total) e = hipMemcpyAsync(c->sq_bytes, bytes + row_off[0], total, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->sq_roff, roff.data(), sizeof(int64_t) * roff.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->sq_woff, woff.data(), sizeof(int64_t) * woff.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(seq_pack_kernel, dim3((unsigned)n_rows), dim3(kBlock), 0, c->stream, (const uint8_t *)c->sq_bytes, (const int64_t *)c->sq_roff,
                           (const int64_t *)c->sq_woff, (int)n_rows, c->sq_code, c->sq_flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);            // (the host vectors above leave scope)
    if (e != hipSuccess) { free_seq(c); return fail(c, MP_ERR_DEVICE, "mp_seq_load: %s", hipGetErrorString(e)); }
    c->sq_roff_host = std::move(roff);
    return MP_OK;
}

int mp_seq_free(mp_ctx *c) {
    if (!c) return MP_ERR_ARG;
    HIPCK(c, hipSetDevice(c->dev));
    free_seq(c);
    return MP_OK;
}

int mp_seq_info(mp_ctx *c, int32_t *n_rows, int64_t *n_bases, int64_t *device_bytes) {
    if (!c) return MP_ERR_ARG;
    if (n_rows) *n_rows = c->sq_n;
    if (n_bases) *n_bases = (int64_t)c->sq_total;
    if (device_bytes) *device_bytes = c->sq_n ? (int64_t)(c->sq_total + 16 + 16 * c->sq_words + 16 * ((size_t)c->sq_n + 1)) : 0;
    return MP_OK;
}

int mp_kmm_scan(mp_ctx *c, const uint8_t *bytes, const int64_t *row_off, int32_t n_rows, int32_t n_pat, const uint8_t *pat_codes,
                const int32_t *pat_off, int32_t max_mm, int32_t term, int64_t cap, int32_t *hits, int64_t *n_hits) {
    if (!c) return MP_ERR_ARG;
    if (n_rows < 0 || n_pat < 0 || !n_hits || cap < 0 || (cap && !hits) || (n_rows && (!bytes || !row_off)) ||
        (n_pat && (!pat_codes || !pat_off)) || max_mm < 0 || term < 0)
        return fail(c, MP_ERR_ARG, "mp_kmm_scan: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    *n_hits = 0;
    if (n_rows == 0 || n_pat == 0) return MP_OK;
    const size_t total = (size_t)(row_off[n_rows] - row_off[0]);
    std::vector<int64_t> roff((size_t)n_rows + 1);
    for (int32_t r = 0; r <= n_rows; r++) roff[(size_t)r] = row_off[r] - row_off[0];
    uint8_t *d_bytes = nullptr;
    int64_t *d_roff = nullptr;
    int rc;
    if ((rc = dev_alloc(c, &d_bytes, total + 16)) || (rc = dev_alloc(c, &d_roff, (size_t)n_rows + 1))) {
        dev_free(c, &d_bytes, total + 16); dev_free(c, &d_roff, (size_t)n_rows + 1);
        return rc;
    }
    hipError_t e = hipMemcpyAsync(d_bytes, bytes + row_off[0], total, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_roff, roff.data(), sizeof(int64_t) * roff.size(), hipMemcpyHostToDevice, c->stream);
    rc = e == hipSuccess ? kmm_scan_device(c, d_bytes, d_roff, nullptr, nullptr, nullptr, roff.data(), n_rows, n_pat, pat_codes, pat_off, max_mm, term, cap, hits, n_hits)
                         : fail(c, MP_ERR_DEVICE, "mp_kmm_scan: %s", hipGetErrorString(e));
    (void)hipStreamSynchronize(c->stream);
    dev_free(c, &d_bytes, total + 16); dev_free(c, &d_roff, (size_t)n_rows + 1);
    return rc;
}

int mp_kmm_scan_resident(mp_ctx *c, int32_t n_pat, const uint8_t *pat_codes, const int32_t *pat_off, int32_t max_mm, int32_t term, int64_t cap,
                         int32_t *hits, int64_t *n_hits) {
    if (!c) return MP_ERR_ARG;
    if (n_pat < 0 || !n_hits || cap < 0 || (cap && !hits) || (n_pat && (!pat_codes || !pat_off)) || max_mm < 0 || term < 0)
        return fail(c, MP_ERR_ARG, "mp_kmm_scan_resident: bad arguments");
    HIPCK(c, hipSetDevice(c->dev));
    *n_hits = 0;
    if (c->sq_n == 0 || n_pat == 0) return MP_OK;
    return kmm_scan_device(c, c->sq_bytes, c->sq_roff, c->sq_code, c->sq_flag, c->sq_woff, c->sq_roff_host.data(), c->sq_n, n_pat, pat_codes, pat_off,
                           max_mm, term, cap, hits, n_hits);
}

}  // extern "C"
