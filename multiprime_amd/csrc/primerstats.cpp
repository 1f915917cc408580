// primerstats.cpp — part of libmprime_hip.so: the per-primer numbers of the core step's TSV behind include/mprime_host.h (H4) — melting
// temperature and the "Information" column's inputs of ALL output primers of an alignment in one call.  Plain C++17 on the host (a few
// hundred primers, a few thousand expansions).  "V20" = scripts/multiPrime-core_V20.py.
//
// Everything that decides a digit is done the way the reference's Python does it, so that the TSV stays byte-identical:
//   * the nearest-neighbour sums are the same left-to-right double additions per expansion (V20:249-261), the closing formula is plain
//     IEEE arithmetic (V20:328-336; the translation unit is built with -ffp-contract=off), the tables and constants come from the caller
//     (multiprime_amd/thermo.py evaluates them with the reference's own expressions);
//   * round(x, 2) is Python's: the double nearest to the correctly rounded two-decimal value of the exact binary x (printf's %.2f is
//     correctly rounded too; rint(100 x) / 100 is the same double away from a tie and is what runs unless 100 x lies within 1e-6 of one);
//   * statistics.mean over a primer's expansions is exact: the doubles are integers times one power of two, summed in 128 bits, and the
//     one division is rounded to nearest-even like Python's int / int.
#include "../../include/mprime.h"
#include "workers.hpp"
#include "../../include/mprime_host.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <thread>
#include <vector>
#include <vector>

namespace {

constexpr int kMaxLen = 64;

// members of a symbol in the reference's enumeration order (degenerate_base, V20:105-107), as base indices A=0 C=1 G=2 T=3
struct Members { int n; int b[4]; };
const Members kMembers[16] = {
    {0, {0, 0, 0, 0}},        // '-' (not a primer symbol)
    {1, {0, 0, 0, 0}},        // A
    {1, {1, 0, 0, 0}},        // C
    {2, {0, 1, 0, 0}},        // M = AC
    {1, {2, 0, 0, 0}},        // G
    {2, {0, 2, 0, 0}},        // R = AG
    {2, {2, 1, 0, 0}},        // S = GC
    {3, {2, 0, 1, 0}},        // V = GAC
    {1, {3, 0, 0, 0}},        // T
    {2, {0, 3, 0, 0}},        // W = AT
    {2, {1, 3, 0, 0}},        // Y = CT
    {3, {0, 3, 1, 0}},        // H = ATC
    {2, {2, 3, 0, 0}},        // K = GT
    {3, {2, 0, 3, 0}},        // D = GAT
    {3, {2, 3, 1, 0}},        // B = GTC
    {4, {0, 3, 2, 1}},        // N = ATGC
};

double py_round2(double x) {
    if (!std::isfinite(x)) return x;
    const double y = x * 100.0;
    if (std::fabs(y) < 4e9) {                               // below 2^32 the product is off by < 1e-6
        const double fr = std::fabs(y - std::floor(y));
        if (std::fabs(fr - 0.5) > 1e-6) return std::rint(y) / 100.0;
    }
    char buf[64];
    snprintf(buf, sizeof buf, "%.2f", x);
    return strtod(buf, nullptr);
}

// exact mean of finite doubles: false when the values do not fit the 128-bit sum (the caller — batchfilters.py — takes such a primer through its numpy form and Python's rationals)
bool exact_mean(const std::vector<double> &v, double &out) {
    if (v.empty()) return false;
    int emin = 0;
    bool any = false;
    std::vector<int64_t> mant(v.size());
    std::vector<int> ex(v.size());
    for (size_t i = 0; i < v.size(); i++) {
        if (!std::isfinite(v[i])) return false;
        int e = 0;
        const double f = std::frexp(v[i], &e);
        mant[i] = (int64_t)std::ldexp(f, 53);                // exact: |f| < 1
        ex[i] = e - 53;
        if (mant[i] != 0 && (!any || ex[i] < emin)) { emin = ex[i]; any = true; }
    }
    if (!any) { out = 0.0; return true; }
    __int128 sum = 0;
    for (size_t i = 0; i < v.size(); i++) {
        if (mant[i] == 0) continue;
        const int sh = ex[i] - emin;
        if (sh > 40) return false;
        sum += (__int128)mant[i] << sh;
    }
    if (v.size() >= ((size_t)1 << 30)) return false;
    const bool neg = sum < 0;
    unsigned __int128 a = neg ? (unsigned __int128)(-sum) : (unsigned __int128)sum;
    if (a == 0) { out = 0.0; return true; }
    const uint64_t n = (uint64_t)v.size();
    auto bitlen = [](unsigned __int128 x) { int b = 0; while (x) { b++; x >>= 1; } return b; };
    const int la = bitlen(a);
    int s = 64 + bitlen(n) - la;
    if (s < 0) s = 0;
    if (la + s > 126) return false;
    const unsigned __int128 num = a << s;
    unsigned __int128 q = num / n;
    const bool rem = (num % n) != 0;
    const int lq = bitlen(q);
    const int drop = lq - 53;
    if (drop <= 0) return false;                              // (cannot happen: the quotient has at least 63 bits)
    const unsigned __int128 half = (unsigned __int128)1 << (drop - 1), low = q & (((unsigned __int128)1 << drop) - 1);
    uint64_t q53 = (uint64_t)(q >> drop);
    if (low > half || (low == half && (rem || (q53 & 1)))) q53++;
    const double r = std::ldexp((double)q53, drop - s + emin);
    out = neg ? -r : r;
    return true;
}

// every expansion of codes[0..k): f(base indices) — order does not matter to the callers (they take exact means)
template <typename F>
void for_each_expansion(const uint8_t *codes, int k, F &&f) {
    int idx[kMaxLen] = {0}, cur[kMaxLen];
    for (int j = 0; j < k; j++) cur[j] = kMembers[codes[j]].b[0];
    for (;;) {
        f(cur);
        int j = k - 1;
        for (; j >= 0; j--) {
            const Members &m = kMembers[codes[j]];
            if (++idx[j] < m.n) { cur[j] = m.b[idx[j]]; break; }
            idx[j] = 0;
            cur[j] = m.b[0];
        }
        if (j < 0) break;
    }
}

bool usable(const uint8_t *codes, int k, double limit) {
    double d = 1;
    for (int j = 0; j < k; j++) {
        if (codes[j] == 0 || codes[j] > 15) return false;
        d *= kMembers[codes[j]].n;
    }
    return d <= limit;
}

}  // namespace

extern "C" {

// params: [0..16) dH[cur][prev], [16..32) dS[cur][prev], [32..36) dH of an end base, [36..40) dS of an end base, [40] dS symmetry term,
// [41] R ln(c) of a self-complementary sequence, [42] of any other, [43] salt correction, [44] 273.15
int mp_primer_tm(int32_t k, int64_t n, const uint8_t *codes, const double *params, double *tm) {
    if (k < 2 || k > kMaxLen || n < 0 || (n && (!codes || !tm)) || !params) return MP_ERR_ARG;
    const double *DH = params, *DS = params + 16, *DHE = params + 32, *DSE = params + 36;
    const double ds_sym = params[40], ln_a = params[41], ln_b = params[42], salt = params[43], kelvin = params[44];
    std::vector<double> vals;
    for (int64_t i = 0; i < n; i++) {
        const uint8_t *c = codes + (size_t)i * k;
        if (!usable(c, k, 1 << 22)) return MP_ERR_ARG;
        vals.clear();
        for_each_expansion(c, k, [&](const int *b) {
            double dh = 0, ds = 0;
            for (int t = 1; t < k; t++) { dh += DH[b[t] * 4 + b[t - 1]]; ds += DS[b[t] * 4 + b[t - 1]]; }      // V20:253-256
            dh += DHE[b[0]] + DHE[b[k - 1]];
            ds += DSE[b[0]] + DSE[b[k - 1]];
            bool sym = (k % 2) == 0;                                                                            // V20:237-246
            for (int t = 0; sym && t < k / 2; t++) sym = b[t] == 3 - b[k / 2 + t];
            if (sym) ds = ds + ds_sym;
            dh = dh * 1000;
            const double ln_c = sym ? ln_a : ln_b;
            vals.push_back(py_round2(1 / ((1 / (dh / (ds + ln_c))) + salt) - kelvin));                          // V20:336, rounded per expansion
        });
        double mean;
        if (!exact_mean(vals, mean)) return MP_ERR_CAPACITY;
        tm[i] = py_round2(mean);                                                                                 // V20:852
    }
    return MP_OK;
}

// gc[i] = round(mean over the expansions of r3[number of G/C], 2) with r3[g] = round(g / k, 3) from the caller (V20:401-407);
// repeat[i] = di_nucleotide (V20:410-416), hairpin[i] = hairpin_check with `distance` (V20:387-398), both as statements about the
// positions' base sets (an expansion picks one base per position independently)
int mp_primer_filters(int32_t k, int64_t n, const uint8_t *codes, const double *r3, int32_t distance, double *gc, uint8_t *repeat, uint8_t *hairpin) {
    if (k < 1 || k > kMaxLen || n < 0 || distance < 0 || (n && (!codes || !gc || !repeat || !hairpin)) || !r3) return MP_ERR_ARG;
    auto comp = [](uint32_t m) { return ((m & 1u) << 3) | ((m & 2u) << 1) | ((m & 4u) >> 1) | ((m & 8u) >> 3); };
    std::vector<double> vals;
    for (int64_t i = 0; i < n; i++) {
        const uint8_t *M = codes + (size_t)i * k;
        if (!usable(M, k, 1 << 22)) return MP_ERR_ARG;
        vals.clear();
        for_each_expansion(M, k, [&](const int *b) {
            int g = 0;
            for (int j = 0; j < k; j++) g += b[j] == 1 || b[j] == 2;
            vals.push_back(r3[g]);
        });
        double mean;
        if (!exact_mean(vals, mean)) return MP_ERR_CAPACITY;
        gc[i] = py_round2(mean);
        bool rep = false;
        for (int o = 0; o + 3 < k && !rep; o++) rep = (M[o] & M[o + 1] & M[o + 2] & M[o + 3]) != 0;                       // XXXX
        for (int o = 0; o + 7 < k && !rep; o++) {                                                                          // (XY) x 4, X != Y
            const uint32_t a = M[o] & M[o + 2] & M[o + 4] & M[o + 6], b = M[o + 1] & M[o + 3] & M[o + 5] & M[o + 7];
            rep = a != 0 && b != 0 && !(a == b && __builtin_popcount(a) == 1);
        }
        for (int o = 0; o + 8 < k && !rep; o++) {                                                                          // (XYZ) x 3, X != Y, Y != Z
            const uint32_t a = M[o] & M[o + 3] & M[o + 6], b = M[o + 1] & M[o + 4] & M[o + 7], c = M[o + 2] & M[o + 5] & M[o + 8];
            for (uint32_t y = 1; y < 16 && !rep; y <<= 1) rep = (b & y) != 0 && (a & ~y & 15u) != 0 && (c & ~y & 15u) != 0;
        }
        repeat[i] = rep;
        bool hp = false;
        for (int s = 0; s <= k - 5 - 5 - distance && !hp; s++)
            for (int o = s + 5 + distance; o < k - 4 && !hp; o++) {
                bool ok = true;
                for (int j = 0; j < 5 && ok; j++) ok = (comp(M[s + 4 - j]) & M[o + j]) != 0;      // RC(stem)[j] = comp(stem[4 - j]) lies in the set at o + j
                hp = ok;
            }
        hairpin[i] = hp;
    }
    return MP_OK;
}

// (H5) the coverage-bitset verdicts of the rows whose window held an IUPAC code (V20:701-707 puts the row's id under EVERY expansion's
// k-mer, V20:1107-1127 decides per k-mer): the row is NOT reached when some expansion is neither matched perfectly nor admissible.  The
// expansions need not be listed: position j CAN mismatch when the row has '-' there or some member of its symbol lies outside the
// primer's; the expansion that takes a mismatching member wherever there is one has the most mismatches, so the row is bad when that
// count exceeds v, or else when a strict position can mismatch at all; a row with more than v gaps is in gap_seq_id: bad for both.
int mp_exception_verdicts(int32_t k, int32_t v, int64_t n, const uint8_t *xc, const int64_t *primer_of, int64_t n_primers, const uint8_t *primers,
                          uint64_t strictF, uint64_t strictR, uint8_t *bad) {
    if (k < 1 || k > kMaxLen || v < 0 || n < 0 || (n && (!xc || !primer_of || !primers || !bad))) return MP_ERR_ARG;
    for (int64_t i = 0; i < n; i++)
        if (primer_of[i] < 0 || primer_of[i] >= n_primers) return MP_ERR_ARG;
    auto body = [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; i++) {
            const uint8_t *row = xc + (size_t)i * k, *pr = primers + (size_t)primer_of[i] * k;
            int gaps = 0, miss = 0;
            uint64_t can = 0;
            for (int j = 0; j < k; j++) {
                const bool gap = row[j] == 0, m = gap || (row[j] & ~pr[j] & 15u) != 0;
                gaps += gap;
                miss += m;
                can |= (uint64_t)m << j;
            }
            const bool both = gaps > v || miss > v;
            bad[2 * i] = both || (can & strictF) != 0;
            bad[2 * i + 1] = both || (can & strictR) != 0;
        }
    };
    int n_thr = n >= 16384 ? (int)std::min<int64_t>(16, std::min<int64_t>((int64_t)std::max(1u, std::thread::hardware_concurrency()), n / 8192)) : 1;
    if (const char *e = getenv("MP_HOST_THREADS")) n_thr = std::max(1, std::min(n_thr, atoi(e)));
    if (n_thr <= 1) { body(0, n); return MP_OK; }
    mp::run_on_threads(n_thr, [&](int t) { body(n * t / n_thr, n * (t + 1) / n_thr); });
    return MP_OK;
}

// (H4b) mp_expand_kmer_words for the exception list as it comes from mp_get_exceptions: the rows with more than v gaps dropped (V20:701-707
// expands only k-mers that stay in the universe), every expansion with its row's WINDOW — what mp_set_extra_rows takes.
int mp_expand_exception_words(int32_t k, int32_t v, int64_t n, const int32_t *x_window, const uint8_t *codes, int64_t cap, void *out_words,
                              int32_t *out_window, int64_t *n_out) {
    if (k < 1 || k > kMaxLen || v < 0 || n < 0 || cap < 0 || !n_out || (n && (!x_window || !codes)) || (cap && (!out_words || !out_window))) return MP_ERR_ARG;
    *n_out = 0;
    std::vector<int64_t> keep;
    keep.reserve((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        const uint8_t *r = codes + (size_t)i * k;
        int gaps = 0;
        for (int j = 0; j < k; j++) gaps += r[j] == 0;
        if (gaps <= v) keep.push_back(i);
    }
    const int64_t m = (int64_t)keep.size();
    std::vector<uint8_t> some;
    const uint8_t *sel = codes;
    if (m != n) {
        some.resize((size_t)m * k);
        for (int64_t i = 0; i < m; i++) memcpy(some.data() + (size_t)i * k, codes + (size_t)keep[(size_t)i] * k, (size_t)k);
        sel = some.data();
    }
    std::vector<int64_t> src((size_t)std::max<int64_t>(cap, 1));
    const int rc = mp_expand_kmer_words(k, m, sel, cap, out_words, src.data(), n_out);
    if (rc) return rc;                                           // (MP_ERR_CAPACITY: *n_out holds the expansions there are)
    for (int64_t o = 0; o < *n_out; o++) out_window[o] = x_window[keep[(size_t)src[(size_t)o]]];
    return MP_OK;
}

// (H5b) the same verdicts as assignments for mp_masks_set_bits, the selection included: of the n exception rows those whose window is an output
// window (slot_of[window] >= 0) and whose row lies in [row0, row0 + n_rows) give two assignments each, in exception order — (output row, local
// row, 0, forward verdict), (…, 1, reverse verdict).  Two passes over contiguous shares of the list (count, then write behind the shares before).
int mp_exception_assignments(int32_t k, int32_t v, int64_t n, const int32_t *x_window, const int64_t *x_row, const uint8_t *xc, int32_t n_windows,
                             const int32_t *slot_of, int64_t row0, int64_t n_rows, int64_t n_primers, const uint8_t *primers, uint64_t strictF,
                             uint64_t strictR, int32_t *cand, int32_t *row, uint8_t *which, uint8_t *value, int64_t *n_out) {
    if (k < 1 || k > kMaxLen || v < 0 || n < 0 || n_windows < 0 || n_rows < 0 || n_rows > 0x7fffffffLL || !n_out ||
        (n && (!x_window || !x_row || !xc || !slot_of || !primers || !cand || !row || !which || !value)))
        return MP_ERR_ARG;
    *n_out = 0;
    for (int32_t w = 0; w < n_windows; w++)
        if (slot_of[w] >= n_primers) return MP_ERR_ARG;
    for (int64_t i = 0; i < n; i++)
        if (x_window[i] < 0 || x_window[i] >= n_windows) return MP_ERR_ARG;
    auto taken = [&](int64_t i) { return slot_of[x_window[i]] >= 0 && x_row[i] >= row0 && x_row[i] - row0 < n_rows; };
    int n_thr = n >= 16384 ? (int)std::min<int64_t>(16, std::min<int64_t>((int64_t)std::max(1u, std::thread::hardware_concurrency()), n / 8192)) : 1;
    if (const char *e = getenv("MP_HOST_THREADS")) n_thr = std::max(1, std::min(n_thr, atoi(e)));
    std::vector<int64_t> first((size_t)n_thr + 1, 0);
    auto share = [&](int t, bool write) {
        int64_t o = first[(size_t)t], found = 0;
        for (int64_t i = n * t / n_thr, i1 = n * (t + 1) / n_thr; i < i1; i++) {
            if (!taken(i)) continue;
            found++;
            if (!write) continue;
            const int32_t slot = slot_of[x_window[i]];
            const uint8_t *r = xc + (size_t)i * k, *pr = primers + (size_t)slot * k;
            int gaps = 0, miss = 0;
            uint64_t can = 0;
            for (int j = 0; j < k; j++) {
                const bool gap = r[j] == 0, m = gap || (r[j] & ~pr[j] & 15u) != 0;
                gaps += gap;
                miss += m;
                can |= (uint64_t)m << j;
            }
            const bool both = gaps > v || miss > v;
            cand[2 * o] = slot; cand[2 * o + 1] = slot;
            row[2 * o] = row[2 * o + 1] = (int32_t)(x_row[i] - row0);
            which[2 * o] = 0; which[2 * o + 1] = 1;
            value[2 * o] = both || (can & strictF) != 0;
            value[2 * o + 1] = both || (can & strictR) != 0;
            o++;
        }
        if (!write) first[(size_t)t + 1] = found;
    };
    if (n_thr <= 1) share(0, false);
    else mp::run_on_threads(n_thr, [&](int t) { share(t, false); });
    for (int t = 0; t < n_thr; t++) first[(size_t)t + 1] += first[(size_t)t];
    if (n_thr <= 1) share(0, true);
    else mp::run_on_threads(n_thr, [&](int t) { share(t, true); });
    *n_out = 2 * first[(size_t)n_thr];
    return MP_OK;
}

}  // extern "C"
