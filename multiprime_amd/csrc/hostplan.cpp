// hostplan.cpp — part of libmprime_hip.so: the native HOST stage of the core step behind include/mprime_host.h (H2).
// Plain C++17 on the host cores, threads over windows; no device calls.  "V20" = scripts/multiPrime-core_V20.py.
//
// What is restated here (each function cites its lines): the insertion-ordered cover / gap_sequence dictionaries of
// get_primers (V20:689-711) rebuilt from the device histograms and the IUPAC exception list, the gates (V20:713-740),
// entropy (V20:602-614), get_optimal_primer_by_viterbi / _by_MM (V20:579-600), refine_by_NN_array (V20:922-1089), the
// structural part of coverage_stast (V20:860-906) and the replay of its stopping rules on the batched evaluations.
// All floating-point expressions keep the reference's operand order and libm calls (log(x)/log(2), Python's round()).
//
// Compiled twice: as it stands for k <= 32 (keys of 32 symbol nibbles in two 64-bit words, the pipeline's usual -l 18..28), and through
// hostplan_wide.cpp with MP_PLAN_WIDE = 1 for 33 <= k <= 63 (four words per key), every exported name suffixed _w64.  The entry points
// of THIS unit look at k (or at the plan's tag) and hand wide work over; a plan is only ever touched by the unit that made it.
#ifndef MP_PLAN_WIDE
#define MP_PLAN_WIDE 0
#endif
#if MP_PLAN_WIDE
#define mp_plan mp_plan_w64
#define mp_plan_error mp_plan_error_w64
#define mp_plan_destroy mp_plan_destroy_w64
#define mp_plan_create mp_plan_create_w64
#define mp_plan_create_segments mp_plan_create_segments_w64
#define mp_plan_create_segments_ready mp_plan_create_segments_ready_w64
#define mp_plan_windows mp_plan_windows_w64
#define mp_plan_sizes mp_plan_sizes_w64
#define mp_plan_candidates mp_plan_candidates_w64
#define mp_plan_seeds mp_plan_seeds_w64
#define mp_plan_chain mp_plan_chain_w64
#define mp_plan_finish mp_plan_finish_w64
#define mp_plan_results mp_plan_results_w64
#define mp_plan_window_table mp_plan_window_table_w64
#define mp_expand_kmers mp_expand_kmers_w64
#define mp_expand_kmer_words mp_expand_kmer_words_w64
#define mp_plan_write_side_files mp_plan_write_side_files_w64
#define mp_plan_write_side_files_part mp_plan_write_side_files_part_w64
#endif
#include "../../include/mprime.h"
#include "workers.hpp"
#include "../../include/mprime_host.h"
#include "planstream.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------------------------
// symbols
// ---------------------------------------------------------------------------------------------------------------
constexpr int kKeyWords = MP_PLAN_WIDE ? 4 : 2;        // 16 symbol nibbles per word
constexpr int kMaxK = MP_PLAN_WIDE ? 63 : 32;          // positions a key (and every per-position array below) holds
typedef uint64_t word_t;                               // a window word as this unit computes with it (read by width, rd_word)

struct Key {                    // k symbol codes, one nibble each (position j = nibble j & 15 of word j >> 4)
    uint64_t q[kKeyWords] = {};
    bool operator==(const Key &o) const {
        bool same = true;
        for (int i = 0; i < kKeyWords; i++) same &= q[i] == o.q[i];
        return same;
    }
    uint32_t get(int j) const { return (uint32_t)((q[j >> 4] >> (4 * (j & 15))) & 15u); }
    void set(int j, uint32_t c) { q[j >> 4] = (q[j >> 4] & ~(15ull << (4 * (j & 15)))) | ((uint64_t)c << (4 * (j & 15))); }
};

// window words at the ABI are uint32 while k <= MP_NARROW_K and uint64 above (mprime.h)
inline word_t rd_word(const void *words, size_t i, int k) {
    return k > MP_NARROW_K ? ((const uint64_t *)words)[i] : (word_t)((const uint32_t *)words)[i];
}

// Key of a window k-mer given as window words (b0, b1 = base index bits, g = gap flags; mprime.h): the bits of 16 positions are
// spread to nibble lanes through a byte table and the one-hot symbol codes are formed for all of them at once.
struct SpreadTable {
    uint32_t t[256];
    SpreadTable() {
        for (int x = 0; x < 256; x++) {
            uint32_t r = 0;
            for (int j = 0; j < 8; j++) if ((x >> j) & 1) r |= 1u << (4 * j);
            t[x] = r;
        }
    }
    uint64_t operator()(uint32_t x16) const { return (uint64_t)t[x16 & 255u] | ((uint64_t)t[(x16 >> 8) & 255u] << 32); }
};
inline Key key_from_words(word_t b0, word_t b1, word_t g, word_t kmask) {
    static const SpreadTable spread;
    const word_t valid = kmask & ~g;
    Key key;
    for (int part = 0; part < kKeyWords; part++) {
        const int sh = 16 * part;
        const uint64_t v = spread((uint32_t)(valid >> sh) & 0xFFFFu), s0 = spread((uint32_t)(b0 >> sh) & 0xFFFFu),
                       s1 = spread((uint32_t)(b1 >> sh) & 0xFFFFu);
        key.q[part] = (v & ~s0 & ~s1) | ((v & s0 & ~s1) << 1) | ((v & ~s0 & s1) << 2) | ((v & s0 & s1) << 3);
    }
    return key;
}

inline uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
inline uint64_t hash_key(const Key &k) {
    uint64_t h = mix(k.q[kKeyWords - 1] + 0x9E3779B97F4A7C15ULL);
    for (int i = kKeyWords - 2; i >= 0; i--) h = mix(k.q[i] ^ h);
    return h;
}

// members of a symbol in the reference's enumeration order (degenerate_base, V20:105-107), as base codes
struct Members { uint8_t n; uint8_t m[4]; };
constexpr uint8_t A = 1, C = 2, G = 4, T = 8;
const Members kMembers[16] = {
    {1, {0, 0, 0, 0}},        // '-'
    {1, {A, 0, 0, 0}},        // A
    {1, {C, 0, 0, 0}},        // C
    {2, {A, C, 0, 0}},        // M = AC
    {1, {G, 0, 0, 0}},        // G
    {2, {A, G, 0, 0}},        // R = AG
    {2, {G, C, 0, 0}},        // S = GC
    {3, {G, A, C, 0}},        // V = GAC
    {1, {T, 0, 0, 0}},        // T
    {2, {A, T, 0, 0}},        // W = AT
    {2, {C, T, 0, 0}},        // Y = CT
    {3, {A, T, C, 0}},        // H = ATC
    {2, {G, T, 0, 0}},        // K = GT
    {3, {G, A, T, 0}},        // D = GAT
    {3, {G, T, C, 0}},        // B = GTC
    {4, {A, T, G, C}},        // N = ATGC
};
inline int set_size(uint32_t code) { return code == 0 ? 1 : __builtin_popcount(code); }     // floor(score), V20:211

// itertools.product over the member lists, last position fastest (degenerate_seq, V20:368-380): calls f(Key) for
// every expansion in that order.  Returns the number of expansions.
template <typename F>
int64_t for_each_expansion(const uint8_t *codes, int k, F &&f) {
    int idx[kMaxK + 1] = {0};
    Key cur;
    for (int j = 0; j < k; j++) cur.set(j, kMembers[codes[j]].m[0]);
    int64_t n = 0;
    for (;;) {
        f(cur);
        n++;
        int j = k - 1;
        for (; j >= 0; j--) {
            const Members &mb = kMembers[codes[j]];
            if (++idx[j] < mb.n) { cur.set(j, mb.m[idx[j]]); break; }
            idx[j] = 0;
            cur.set(j, mb.m[0]);
        }
        if (j < 0) break;
    }
    return n;
}

inline double expansions_of(const uint8_t *codes, int k) {
    double d = 1;
    for (int j = 0; j < k; j++) d *= set_size(codes[j]);
    return d;
}

// Python's round(x, 2) on a float: correctly rounded decimal with two places (round-half-even on the exact binary
// value), then back to the nearest double — what glibc's printf/strtod pair does.
double py_round2(double x) {
    if (!std::isfinite(x)) return x;
    char buf[64];
    snprintf(buf, sizeof buf, "%.2f", x);
    return strtod(buf, nullptr);
}

// math.log(x, 2) of CPython: log(x) / log(2.0), two libm calls and a division
inline double py_log2(double x) { return std::log(x) / std::log(2.0); }

// ---------------------------------------------------------------------------------------------------------------
// per-window tables
// ---------------------------------------------------------------------------------------------------------------
struct Entry {
    Key key;
    int64_t count;
    int64_t first_row;
    int32_t first_sub;        // expansion index inside its exception row (0 for plain rows)
    int32_t ngap;
};

// open-addressing map Key -> index into a vector<Entry>
struct KeyMap {
    std::vector<int32_t> slot;
    uint32_t mask = 0;
    void reset(size_t n_expected) {
        size_t cap = 16;
        while (cap < 2 * n_expected + 8) cap <<= 1;
        slot.assign(cap, -1);
        mask = (uint32_t)(cap - 1);
    }
    // returns the index stored for `key`, or -1 after inserting `idx_new`
    int32_t find_or_insert(const Key &key, const std::vector<Entry> &ents, int32_t idx_new) {
        uint32_t h = (uint32_t)hash_key(key) & mask;
        for (;;) {
            int32_t s = slot[h];
            if (s < 0) { slot[h] = idx_new; return -1; }
            if (ents[(size_t)s].key == key) return s;
            h = (h + 1) & mask;
        }
    }
    int32_t find(const Key &key, const std::vector<Entry> &ents) const {
        if (slot.empty()) return -1;
        uint32_t h = (uint32_t)hash_key(key) & mask;
        for (;;) {
            int32_t s = slot[h];
            if (s < 0) return -1;
            if (ents[(size_t)s].key == key) return s;
            h = (h + 1) & mask;
        }
    }
};

struct Seed {
    uint8_t index[kMaxK + 1];                       // base index per position
    std::vector<Key> chain;                  // chain[0] = the seed
    std::vector<int64_t> cov;                // running perfect coverage (optimal_coverage_init)
    std::vector<uint8_t> stops;              // a structural break rule ends the loop after this member
    int64_t first_cand = -1;
    // after replay
    int final_i = 0;
    int64_t F = 0, R = 0;
};

struct Window {
    int32_t status = MP_WIN_PLANNED;
    int64_t cover_number = 0, gap_number = 0;
    double cbit = std::numeric_limits<double>::quiet_NaN(), tbit = std::numeric_limits<double>::quiet_NaN();
    std::vector<Entry> cover, gap;           // insertion order
    KeyMap cover_map;                        // over `cover`
    int n_seeds = 0;
    Seed seeds[2];
    Key present;                             // phantom key the reference inserts for the NM seed (V20:787/800/835)
    // results
    Key primer;
    int64_t cov = 0, f_mis = 0, r_mis = 0;
    int32_t nonsense = 0, n_dege = 0;
};

}  // namespace

struct mp_plan {
    int32_t wide = MP_PLAN_WIDE;             // first member in both builds: which unit owns the plan
    char err[512] = {0};
    mp_plan_params P{};
    std::vector<Window> win;
    std::vector<int32_t> planned;            // window indices, ascending
    int64_t n_cand = 0;
    bool finished = false;
};

#if !MP_PLAN_WIDE
// the wide unit's entry points (hostplan_wide.cpp): same signatures, its own plan type
struct mp_plan_w64;
extern "C" {
const char *mp_plan_error_w64(const mp_plan_w64 *p);
void mp_plan_destroy_w64(mp_plan_w64 *p);
int mp_plan_create_w64(const mp_plan_params *params, int64_t n_entries, const int32_t *e_window, const void *e_words, const int64_t *e_count,
                       const int64_t *e_first, int64_t n_exc, const int32_t *x_window, const int64_t *x_row, const uint8_t *x_codes,
                       const int64_t *freq, const int64_t *nn, mp_plan_w64 **out);
int mp_plan_create_segments_w64(const mp_plan_params *params, const int64_t *e_off, const void *e_words, const int32_t *e_count,
                                const int32_t *e_first, int64_t row_base, int64_t n_exc, const int32_t *x_window, const int64_t *x_row,
                                const uint8_t *x_codes, const int64_t *freq, const int64_t *nn, mp_plan_w64 **out);
int mp_plan_create_segments_ready_w64(const mp_plan_params *params, const int64_t *e_off, const void *e_words, const int32_t *e_count,
                                      const int32_t *e_first, int64_t row_base, int64_t n_exc, const int32_t *x_window, const int64_t *x_row,
                                      const uint8_t *x_codes, const int64_t *freq, const int64_t *nn, const void *ready, const uint8_t *skip,
                                      mp_plan_w64 **out);
int mp_plan_windows_w64(const mp_plan_w64 *p, int32_t *status, int64_t *cover_number, int64_t *gap_number, double *cbit, double *tbit);
int mp_plan_sizes_w64(const mp_plan_w64 *p, int32_t *n_planned, int64_t *n_candidates);
int mp_plan_candidates_w64(const mp_plan_w64 *p, int32_t *cand_window, uint8_t *cand_codes);
int mp_plan_seeds_w64(const mp_plan_w64 *p, int32_t w, uint8_t *nm, uint8_t *mm, int32_t *has_mm, int32_t *n_chain_nm, int32_t *n_chain_mm);
int mp_plan_chain_w64(const mp_plan_w64 *p, int32_t w, int32_t seed, int32_t cap, uint8_t *codes, int64_t *cov, uint8_t *stops, int32_t *n);
int mp_plan_finish_w64(mp_plan_w64 *p, const int64_t *ev);
int mp_plan_results_w64(const mp_plan_w64 *p, int32_t *window, double *cbit, double *tbit, uint8_t *primer_codes, int64_t *cov, int64_t *f_mis,
                        int64_t *r_mis, int32_t *nonsense, int32_t *n_dege, int64_t *cover_number);
int mp_plan_window_table_w64(const mp_plan_w64 *p, int32_t w, int32_t which, int64_t cap, uint8_t *codes, int64_t *counts, int64_t *first_row,
                             int64_t *n);
int mp_expand_kmers_w64(int32_t k, int64_t n, const uint8_t *codes, int64_t cap, uint8_t *out_codes, int64_t *out_src, int64_t *n_out);
int mp_expand_kmer_words_w64(int32_t k, int64_t n, const uint8_t *codes, int64_t cap, void *out_words, int64_t *out_src, int64_t *n_out);
int mp_plan_write_side_files_part_w64(const mp_plan_w64 *p, int32_t n_out, const int32_t *out_window, const int64_t *out_pos,
                                      const uint8_t *primer_codes, uint64_t strictF, uint64_t strictR, const int64_t *dev_off,
                                      const void *dev_words, int64_t n_dev, const int32_t *labels, int32_t n_rows, int64_t n_exc,
                                      const int32_t *x_window, const int64_t *x_row, const uint8_t *x_codes, const uint8_t *ids,
                                      const int64_t *id_off, const char *noncov_path, const char *gap_path, int32_t part);
}
// a plan made by the wide unit goes back to it
#define MP_PLAN_FORWARD(fn, ...) if (p && p->wide) return fn##_w64(reinterpret_cast<const mp_plan_w64 *>(p), ##__VA_ARGS__)
#define MP_PLAN_FORWARD_MUT(fn, ...) if (p && p->wide) return fn##_w64(reinterpret_cast<mp_plan_w64 *>(p), ##__VA_ARGS__)
#else
#define MP_PLAN_FORWARD(fn, ...) (void)0
#define MP_PLAN_FORWARD_MUT(fn, ...) (void)0
#endif

namespace {

int pfail(mp_plan *p, int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(p->err, sizeof p->err, fmt, ap);
    va_end(ap);
    return code;
}

struct NNArr {                                // nn[i][a][b], i < k-1
    int64_t v[kMaxK][4][4];
};

// sum of cover[] over the expansions of a primer (V20:954-956)
int64_t perfect(const Window &w, const uint8_t *P, int k) {
    bool concrete = true;
    for (int j = 0; j < k; j++) concrete &= (P[j] & (P[j] - 1)) == 0;
    if (concrete) {
        Key key;
        for (int j = 0; j < k; j++) key.set(j, P[j]);
        int32_t s = w.cover_map.find(key, w.cover);
        return s < 0 ? 0 : w.cover[(size_t)s].count;
    }
    int64_t tot = 0;
    for_each_expansion(P, k, [&](const Key &e) {
        int32_t s = w.cover_map.find(e, w.cover);
        if (s >= 0) tot += w.cover[(size_t)s].count;
    });
    return tot;
}

// np.argsort(x)[::-1] with the stable tie order of the author's numpy (SURVEY A-14): ascending stable, reversed
inline void desc_stable(const int64_t v[4], int order[4]) {
    int idx[4] = {0, 1, 2, 3};
    std::stable_sort(idx, idx + 4, [&](int a, int b) { return v[a] < v[b]; });
    for (int i = 0; i < 4; i++) order[i] = idx[3 - i];
}
inline int npos(const int64_t v[4]) { return (v[0] > 0) + (v[1] > 0) + (v[2] > 0) + (v[3] > 0); }

struct Refined {
    uint8_t P[kMaxK + 1];
    int64_t c2;
    int64_t cv[kMaxK];
    NNArr nn;
};

// refine_by_NN_array (V20:922-1089): add one base next to the weakest nearest-neighbour link(s); among the weakest
// links keep the first one with the largest perfect coverage.
void refine(const Window &w, int k, const uint8_t *primer, int64_t cov, const uint8_t *index, const int64_t *nn_cov,
            const NNArr &NN, Refined &best) {
    const int last = k - 2;
    int64_t lo = nn_cov[0];
    for (int i = 1; i <= last; i++) lo = std::min(lo, nn_cov[i]);
    bool have = false;
    Refined cur;
    for (int i = 0; i <= last; i++) {
        if (nn_cov[i] != lo) continue;
        cur.nn = NN;
        memcpy(cur.cv, nn_cov, sizeof(int64_t) * (size_t)(k - 1));
        memcpy(cur.P, primer, (size_t)k);
        cur.c2 = cov;
        const int row = index[i], col = index[i + 1];
        auto widen = [&](int pos, const int64_t ranking[4], int skip) -> int {
            int order[4];
            desc_stable(ranking, order);
            for (int t = 0; t < 4; t++) {
                int x = order[t];
                if (x == skip) continue;
                cur.P[pos] = (uint8_t)(1u << x);
                cur.c2 += perfect(w, cur.P, k);                                   // coverage gained (V20:954-956)
                cur.P[pos] = (uint8_t)(primer[pos] | (1u << x));
                return x;
            }
            return -1;
        };
        auto &nn = cur.nn.v;
        int64_t colv[4] = {nn[0][0][col], nn[0][1][col], nn[0][2][col], nn[0][3][col]};
        if (i == 0 && npos(colv) > 1) {
            // position 0 (V20:941-965): fold predecessor x of `col` into the seed's row
            int x = widen(0, colv, row);
            if (x >= 0) {
                for (int b = 0; b < 4; b++) { nn[0][row][b] += nn[0][x][b]; nn[0][x][b] = 0; }
                cur.cv[0] = nn[0][row][col];
            }
        } else if (i == 0 && !(npos(nn[0][row]) > 1)) {
            // V20:1002-1003
        } else if (i == last && i != 0) {
            // last position (V20:1004-1031)
            int64_t line[4] = {nn[i][row][0], nn[i][row][1], nn[i][row][2], nn[i][row][3]};
            if (npos(line) > 1) {
                int x = widen(i + 1, line, col);
                if (x >= 0) {
                    for (int a = 0; a < 4; a++) { nn[i][a][col] += nn[i][a][x]; nn[i][a][x] = 0; }
                    cur.cv[i] = nn[i][row][col];
                }
            }
        } else if (i + 2 <= k - 1) {
            // inner position i+1, shared by link i and link i+1 (V20:967-1001, 1032-1072)
            const int nrow = index[i + 1], ncol = index[i + 2];
            int64_t both[4];
            for (int b = 0; b < 4; b++) both[b] = std::min(nn[i][row][b], nn[i + 1][b][ncol]);
            if (npos(both) > 1) {
                int x = widen(i + 1, both, col);
                if (x >= 0) {
                    for (int a = 0; a < 4; a++) { nn[i][a][col] += nn[i][a][x]; nn[i][a][x] = 0; }
                    for (int b = 0; b < 4; b++) { nn[i + 1][nrow][b] += nn[i + 1][x][b]; nn[i + 1][x][b] = 0; }
                    cur.cv[i] = nn[i][row][col];
                    cur.cv[i + 1] = nn[i + 1][nrow][ncol];
                }
            }
        }
        if (!have || cur.c2 > best.c2) { best = cur; have = true; }
    }
}

// everything coverage_stast (V20:860-920) does that does not depend on an evaluation
int build_chain(const Window &w, Seed &s, int k, const NNArr &NN, double d, int n_max) {
    uint8_t P[kMaxK + 1];
    Key key;
    for (int j = 0; j < k; j++) { P[j] = (uint8_t)(1u << s.index[j]); key.set(j, P[j]); }
    int32_t slot = w.cover_map.find(key, w.cover);
    int64_t cov = slot < 0 ? 0 : w.cover[(size_t)slot].count;
    NNArr nn = NN;
    int64_t nn_cov[kMaxK];
    for (int i = 0; i < k - 1; i++) nn_cov[i] = NN.v[i][s.index[i]][s.index[i + 1]];
    s.chain.push_back(key);
    s.cov.push_back(cov);
    s.stops.push_back(0);
    Refined r;
    for (int guard = 0; guard < 4 * kMaxK + 8; guard++) {
        refine(w, k, P, cov, s.index, nn_cov, nn, r);
        memcpy(P, r.P, (size_t)k);
        cov = r.c2;
        nn = r.nn;
        double deg = expansions_of(P, k);
        int ndeg = 0;
        for (int j = 0; j < k; j++) ndeg += set_size(P[j]) > 1;
        bool same = memcmp(r.cv, nn_cov, sizeof(int64_t) * (size_t)(k - 1)) == 0;
        bool stop = same || 2 * deg > d || 3 * deg / 2 > d || ndeg == n_max;              // V20:899-904
        Key ck;
        for (int j = 0; j < k; j++) ck.set(j, P[j]);
        s.chain.push_back(ck);
        s.cov.push_back(cov);
        s.stops.push_back(stop ? 1 : 0);
        if (stop) return MP_OK;
        memcpy(nn_cov, r.cv, sizeof(int64_t) * (size_t)(k - 1));
    }
    return MP_ERR_ARG;       // a chain that neither saturates nor stops: cannot happen (every step adds a base)
}

// get_optimal_primer_by_viterbi (V20:579-593): max-sum path over base frequencies and nearest-neighbour counts;
// ties go to the lowest base index (numpy argmax takes the first)
void viterbi(const int64_t *freq /*[4][k]*/, const int64_t *nn /*[k-1][4][4]*/, int k, uint8_t *path) {
    int64_t score[4];
    uint8_t back[kMaxK + 1][4];
    for (int a = 0; a < 4; a++) score[a] = freq[a * k + 0];
    for (int t = 1; t < k; t++) {
        int64_t nxt[4];
        for (int b = 0; b < 4; b++) {
            int besta = 0;
            int64_t bestv = 0;
            for (int a = 0; a < 4; a++) {
                int64_t m = score[a] + nn[((t - 1) * 4 + a) * 4 + b] + freq[b * k + t];
                if (a == 0 || m > bestv) { bestv = m; besta = a; }
            }
            nxt[b] = bestv;
            back[t][b] = (uint8_t)besta;
        }
        memcpy(score, nxt, sizeof score);
    }
    int cur = 0;
    for (int a = 1; a < 4; a++) if (score[a] > score[cur]) cur = a;
    path[k - 1] = (uint8_t)cur;
    for (int t = k - 1; t >= 1; t--) { cur = back[t][cur]; path[t - 1] = (uint8_t)cur; }
}

struct Sight {                 // one sighting of a k-mer inside a window, before merging
    Key key;
    int64_t count, row;
    int32_t sub, ngap;
};

// what a planning thread keeps from window to window: the sightings, the merged table and its map (a deep window's are hundreds of
// kilobytes — allocated afresh they went through mmap / munmap once per window and thread, and the page faults of 32 threads in one
// address space were most of the stage's time at 131072 x 1000)
struct Scratch {
    double t_sights = 0, t_merge = 0, t_sort = 0, t_gates = 0, t_chain = 0;      // MP_TRACE: seconds this thread spent per phase
    std::vector<Sight> sights;
    std::vector<Entry> all;               // merged, in the order of the first sighting met; `order` lists it in insertion order
    std::vector<uint32_t> order, order2;
    std::vector<uint64_t> keys, keys2;
    KeyMap map;
};

// S.order = the indices of S.all by (first row, expansion index inside that row): a byte-wise radix sort over the bytes in which the
// keys differ at all (std::sort on the 40-byte entries was most of the stage's time: ~3000 entries per window at 131072 x 1000).
// Keys are unique (a row holds one k-mer per window; expansions of one exception row differ in the index), so stability is not an issue.
void sort_insertion_order(Scratch &S) {
    const size_t n = S.all.size();
    S.order.resize(n); S.keys.resize(n);
    uint64_t diff = 0;
    bool wide_sub = false;
    for (size_t i = 0; i < n; i++) {
        const Entry &e = S.all[i];
        wide_sub |= (uint64_t)e.first_sub >= ((uint64_t)1 << 24) || (uint64_t)e.first_row >= ((uint64_t)1 << 40);
        S.keys[i] = ((uint64_t)e.first_row << 24) | ((uint64_t)e.first_sub & 0xFFFFFFu);
        S.order[i] = (uint32_t)i;
        diff |= S.keys[i] ^ S.keys[0];
    }
    if (wide_sub) {                        // beyond the packed key's range (2^40 rows, 2^24 expansions of one k-mer): compare the fields
        std::sort(S.order.begin(), S.order.end(), [&](uint32_t a, uint32_t b) {
            const Entry &x = S.all[a], &y = S.all[b];
            return x.first_row != y.first_row ? x.first_row < y.first_row : x.first_sub < y.first_sub;
        });
        return;
    }
    if (n < 64) {
        std::sort(S.order.begin(), S.order.end(), [&](uint32_t a, uint32_t b) { return S.keys[a] < S.keys[b]; });
        return;
    }
    S.order2.resize(n); S.keys2.resize(n);
    for (int byte = 0; byte < 8; byte++) {
        if (!((diff >> (8 * byte)) & 0xFFu)) continue;
        size_t cnt[257] = {0};
        for (size_t i = 0; i < n; i++) cnt[((S.keys[i] >> (8 * byte)) & 0xFFu) + 1]++;
        for (int b = 0; b < 256; b++) cnt[b + 1] += cnt[b];
        for (size_t i = 0; i < n; i++) {
            const size_t at = cnt[(S.keys[i] >> (8 * byte)) & 0xFFu]++;
            S.keys2[at] = S.keys[i]; S.order2[at] = S.order[i];
        }
        S.keys.swap(S.keys2); S.order.swap(S.order2);
    }
}

// cover / gap_sequence of one window in the reference's dict insertion order (V20:689-711): a key takes the place of
// its earliest sighting (row, then expansion index inside that row); counts add up.  The merged table stays in the scratch area
// (cover entries: ngap <= v, gap entries: the others, both in insertion order); materialize() copies it into the window.
void build_tables(Window &w, Scratch &S, int v, int64_t n_exc_cover, int64_t n_exp, size_t &n_cover) {
    std::vector<Sight> &sights = S.sights;
    std::vector<Entry> &all = S.all;
    all.clear();
    all.reserve(sights.size());
    KeyMap &map = S.map;
    map.reset(sights.size());
    for (const Sight &s : sights) {
        int32_t at = map.find_or_insert(s.key, all, (int32_t)all.size());
        if (at < 0) all.push_back(Entry{s.key, s.count, s.row, s.sub, s.ngap});
        else {
            Entry &e = all[(size_t)at];
            e.count += s.count;
            if (s.row < e.first_row || (s.row == e.first_row && s.sub < e.first_sub)) { e.first_row = s.row; e.first_sub = s.sub; }
        }
    }
    const auto t_a = std::chrono::steady_clock::now();
    sort_insertion_order(S);
    S.t_sort += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count();
    int64_t csum = 0, gsum = 0;
    n_cover = 0;
    for (const Entry &e : all) {
        if (e.ngap > v) gsum += e.count;
        else { csum += e.count; n_cover++; }
    }
    w.gap_number = gsum;
    w.cover_number = csum - n_exp + n_exc_cover;      // cover_number counts sequences (V20:702), cover counts expansions
}

// the window's own copies of the tables: `cover` for every window that is planned, both when the caller keeps the tables
void materialize(Window &w, const Scratch &S, int v, size_t n_cover, bool want_gap) {
    w.cover.reserve(n_cover);
    if (want_gap) w.gap.reserve(S.all.size() - n_cover);
    for (uint32_t i : S.order) {
        const Entry &e = S.all[i];
        if (e.ngap > v) { if (want_gap) w.gap.push_back(e); }
        else w.cover.push_back(e);
    }
}

// look-ups into `cover` (perfect coverage of a candidate's expansions, nonsense count): only windows past the gates need them
void build_cover_map(Window &w) {
    w.cover_map.reset(w.cover.size());
    for (size_t i = 0; i < w.cover.size(); i++) w.cover_map.find_or_insert(w.cover[i].key, w.cover, (int32_t)i);
}

// entropy (V20:602-614), same summation order: the cover entries in insertion order, then the gap entries
void entropy(Window &w, const Scratch &S, int v) {
    const std::vector<Entry> &all = S.all;
    const int64_t cn = w.cover_number, gn = w.gap_number, tot = cn + gn;
    double cbit = 0, tbit = 0;
    for (uint32_t i : S.order) {
        const Entry &e = all[i];
        if (e.ngap > v) continue;
        double pc = (double)e.count / (double)cn, pt = (double)e.count / (double)tot;
        cbit += pc * py_log2(pc);
        tbit += pt * py_log2(pt);
    }
    for (uint32_t i : S.order) {
        const Entry &e = all[i];
        if (e.ngap <= v) continue;
        double pt = (double)e.count / (double)tot;
        tbit += pt * py_log2(pt);
    }
    w.cbit = py_round2(-cbit);
    w.tbit = py_round2(-tbit);
}

int plan_window(mp_plan *p, int wi, Scratch &S, int64_t n_exc_cover, int64_t n_exp, const int64_t *freq,
                const int64_t *nn) {
    const mp_plan_params &P = p->P;
    const int k = P.k;
    Window &w = p->win[(size_t)wi];
    size_t n_cover = 0;
    const auto t_0 = std::chrono::steady_clock::now();
    build_tables(w, S, P.v, n_exc_cover, n_exp, n_cover);
    const auto t_1 = std::chrono::steady_clock::now();
    S.t_merge += std::chrono::duration<double>(t_1 - t_0).count();
    struct Tail { Scratch &S; std::chrono::steady_clock::time_point t; ~Tail() { S.t_gates += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } tail{S, t_1};
    // a window that stops at a gate keeps its tables only when the caller asked for them (JSON side files, mp_plan_window_table)
    auto stop = [&](int32_t status) {
        w.status = status;
        if (P.keep_tables) materialize(w, S, P.v, n_cover, true);
        return MP_OK;
    };
    // gates (V20:713-740)
    if (py_round2((double)w.gap_number / (double)P.total_sequences) >= (1 - P.coverage)) return stop(MP_WIN_GAP_GATE);
    if (n_cover == 0) return stop(MP_WIN_NO_COVER);
    entropy(w, S, P.v);
    if (w.tbit > P.entropy_threshold) return stop(MP_WIN_ENTROPY);
    int bases = 0;
    for (int a = 0; a < 4; a++) {
        int64_t s = 0;
        for (int j = 0; j < k; j++) s += freq[a * k + j];
        bases += s > 0;
    }
    if (bases < 4) return stop(MP_WIN_FEW_BASES);                                       // V20:736
    for (int j = 0; j < k; j++)
        if (freq[0 * k + j] + freq[1 * k + j] + freq[2 * k + j] + freq[3 * k + j] == 0) return stop(MP_WIN_GAP_COLUMN);
    materialize(w, S, P.v, n_cover, P.keep_tables != 0);
    build_cover_map(w);
    NNArr NN;
    memset(&NN, 0, sizeof NN);
    memcpy(NN.v, nn, sizeof(int64_t) * (size_t)(k - 1) * 16);
    // seeds
    Seed &nm = w.seeds[0];
    viterbi(freq, nn, k, nm.index);
    w.n_seeds = 1;
    // get_optimal_primer_by_MM (V20:595-600): first of the most frequent gap-free k-mers
    int64_t best = 0;
    int best_i = -1;
    for (size_t i = 0; i < w.cover.size(); i++)
        if (w.cover[i].ngap == 0 && w.cover[i].count > best) { best = w.cover[i].count; best_i = (int)i; }
    if (best_i >= 0) {
        Seed &mm = w.seeds[1];
        bool same = true;
        for (int j = 0; j < k; j++) {
            uint32_t c = w.cover[(size_t)best_i].key.get(j);
            mm.index[j] = (uint8_t)__builtin_ctz(c);
            same &= mm.index[j] == nm.index[j];
        }
        if (!same) w.n_seeds = 2;
    }
    for (int j = 0; j < k; j++) w.present.set(j, 1u << nm.index[j]);
    for (int s = 0; s < w.n_seeds; s++) {
        int rc = build_chain(w, w.seeds[s], k, NN, P.max_degeneracy, P.max_dege_positions);
        if (rc != MP_OK) return pfail(p, rc, "refinement chain of window %d does not terminate", wi);
    }
    return MP_OK;
}

int resolve_threads(int asked, int64_t work_items) {
    int n = asked;
    if (const char *e = getenv("MP_HOST_THREADS")) n = atoi(e);
    if (n <= 0) {
        n = (int)std::thread::hardware_concurrency();
        if (n <= 0) n = 1;
        n = std::min(n, 32);
    }
    if ((int64_t)n > work_items) n = (int)std::max<int64_t>(1, work_items);
    return n;
}

}  // namespace

extern "C" {

const char *mp_plan_error(const mp_plan *p) {
    MP_PLAN_FORWARD(mp_plan_error);
    return p ? p->err : "mp_plan_create failed";
}

void mp_plan_destroy(mp_plan *p) {
#if !MP_PLAN_WIDE
    if (p && p->wide) { mp_plan_destroy_w64(reinterpret_cast<mp_plan_w64 *>(p)); return; }
#endif
    delete p;
}

// the body of mp_plan_create; allocation failures (here and in the worker threads) leave as MP_ERR_NOMEM, never as exceptions
// entries either as (e_window, 64-bit counts and global first rows) in any order, or — one rank's read-back as it stands — as window
// segments e_off [W+1] with 32-bit counts and local first rows (+ row_base)
struct EntryInput {
    int64_t n;
    const int32_t *window;
    const int64_t *off;
    const void *words;
    const int64_t *count64, *first64;
    const int32_t *count32, *first32;
    int64_t row_base;
    mp_ready_gate *ready;                     // null, or: the entries of windows below the gate have arrived (mp_plan_create_streamed)
    const uint8_t *skip;                      // null, or: windows the entropy gate rejected on the device (no entries, not planned)
    int64_t count(int64_t i) const { return count64 ? count64[i] : (int64_t)count32[i]; }
    int64_t first(int64_t i) const { return first64 ? first64[i] : (int64_t)first32[i] + row_base; }
};
static int plan_create_body(mp_plan *p, const EntryInput &E, int64_t n_exc, const int32_t *x_window, const int64_t *x_row, const uint8_t *x_codes,
                            const int64_t *freq, const int64_t *nn);
static int plan_create_guarded(const mp_plan_params *params, const EntryInput &E, int64_t n_exc, const int32_t *x_window, const int64_t *x_row,
                               const uint8_t *x_codes, const int64_t *freq, const int64_t *nn, mp_plan **out) {
    if (!out) return MP_ERR_ARG;
    *out = nullptr;
    if (!params || !freq || !nn) return MP_ERR_ARG;
    mp_plan *p = new (std::nothrow) mp_plan();
    if (!p) return MP_ERR_NOMEM;
    *out = p;                                  // returned even on failure so that the caller can read the message
    p->P = *params;
    try {
        return plan_create_body(p, E, n_exc, x_window, x_row, x_codes, freq, nn);
    } catch (const std::bad_alloc &) {
        return pfail(p, MP_ERR_NOMEM, "mp_plan_create: out of memory");
    } catch (const std::exception &e) {
        return pfail(p, MP_ERR_ARG, "mp_plan_create: %s", e.what());
    }
}

int mp_plan_create(const mp_plan_params *params, int64_t n_entries, const int32_t *e_window, const void *e_words,
                   const int64_t *e_count, const int64_t *e_first, int64_t n_exc, const int32_t *x_window,
                   const int64_t *x_row, const uint8_t *x_codes, const int64_t *freq, const int64_t *nn, mp_plan **out) {
#if !MP_PLAN_WIDE
    if (params && params->k > kMaxK)
        return mp_plan_create_w64(params, n_entries, e_window, e_words, e_count, e_first, n_exc, x_window, x_row, x_codes, freq, nn, (mp_plan_w64 **)out);
#endif
    if (n_entries < 0 || (n_entries && (!e_window || !e_words || !e_count || !e_first))) return MP_ERR_ARG;
    const EntryInput E{n_entries, e_window, nullptr, e_words, e_count, e_first, nullptr, nullptr, 0, nullptr, nullptr};
    return plan_create_guarded(params, E, n_exc, x_window, x_row, x_codes, freq, nn, out);
}

int mp_plan_create_segments(const mp_plan_params *params, const int64_t *e_off, const void *e_words, const int32_t *e_count,
                            const int32_t *e_first, int64_t row_base, int64_t n_exc, const int32_t *x_window, const int64_t *x_row,
                            const uint8_t *x_codes, const int64_t *freq, const int64_t *nn, mp_plan **out) {
    return mp_plan_create_segments_ready(params, e_off, e_words, e_count, e_first, row_base, n_exc, x_window, x_row, x_codes, freq, nn, nullptr, nullptr, out);
}

// planstream.hpp: the same with the entries still arriving — windows below the gate `ready` (an mp_ready_gate) are complete
int mp_plan_create_segments_ready(const mp_plan_params *params, const int64_t *e_off, const void *e_words, const int32_t *e_count,
                                  const int32_t *e_first, int64_t row_base, int64_t n_exc, const int32_t *x_window, const int64_t *x_row,
                                  const uint8_t *x_codes, const int64_t *freq, const int64_t *nn, const void *ready, const uint8_t *skip,
                                  mp_plan **out) {
#if !MP_PLAN_WIDE
    if (params && params->k > kMaxK)
        return mp_plan_create_segments_ready_w64(params, e_off, e_words, e_count, e_first, row_base, n_exc, x_window, x_row, x_codes, freq, nn, ready,
                                                 skip, (mp_plan_w64 **)out);
#endif
    if (!params || !e_off || params->n_windows < 0) return MP_ERR_ARG;
    const int64_t n = e_off[params->n_windows];
    if (n < 0 || (n && (!e_words || !e_count || !e_first))) return MP_ERR_ARG;
    const EntryInput E{n, nullptr, e_off, e_words, nullptr, nullptr, e_count, e_first, row_base, static_cast<mp_ready_gate *>(const_cast<void *>(ready)), skip};
    return plan_create_guarded(params, E, n_exc, x_window, x_row, x_codes, freq, nn, out);
}

static int plan_create_body(mp_plan *p, const EntryInput &E, int64_t n_exc, const int32_t *x_window, const int64_t *x_row, const uint8_t *x_codes,
                            const int64_t *freq, const int64_t *nn) {
    const mp_plan_params &P = p->P;
    const int k = P.k, W = P.n_windows;
    const int64_t n_entries = E.n;
    const int32_t *e_window = E.window;
    const void *e_words = E.words;
    if (k < 2 || k > kMaxK || W < 0 || P.v < 0 || P.total_sequences <= 0) return pfail(p, MP_ERR_ARG, "mp_plan_create: bad parameters");
    if (n_exc < 0 || (n_exc && (!x_window || !x_row || !x_codes))) return pfail(p, MP_ERR_ARG, "mp_plan_create: bad arguments");
    // counting sort of entries and exceptions by window
    std::vector<int64_t> eoff((size_t)W + 1, 0), xoff((size_t)W + 1, 0);
    bool e_sorted = true;                      // one rank's histogram read-back arrives window by window: no scatter needed then
    if (E.off) {
        for (int w = 0; w <= W; w++) {
            if (E.off[w] < 0 || (w && E.off[w] < E.off[w - 1])) return pfail(p, MP_ERR_ARG, "window offsets must not decrease");
            eoff[(size_t)w] = E.off[w];
        }
        if (E.off[0] != 0) return pfail(p, MP_ERR_ARG, "window offsets must start at 0");
    } else {
        for (int64_t i = 0; i < n_entries; i++) {
            if (e_window[i] < 0 || e_window[i] >= W) return pfail(p, MP_ERR_ARG, "entry %lld: window %d out of range", (long long)i, e_window[i]);
            eoff[(size_t)e_window[i] + 1]++;
            e_sorted &= i == 0 || e_window[i - 1] <= e_window[i];
        }
    }
    for (int64_t i = 0; i < n_exc; i++) {
        if (x_window[i] < 0 || x_window[i] >= W) return pfail(p, MP_ERR_ARG, "exception %lld: window %d out of range", (long long)i, x_window[i]);
        xoff[(size_t)x_window[i] + 1]++;
    }
    for (int w = 0; w < W; w++) { if (!E.off) eoff[(size_t)w + 1] += eoff[(size_t)w]; xoff[(size_t)w + 1] += xoff[(size_t)w]; }
    std::vector<int64_t> eidx(e_sorted ? 0 : (size_t)n_entries), xidx((size_t)n_exc);
    {
        std::vector<int64_t> cur(eoff.begin(), eoff.end() - 1);
        if (!e_sorted)
            for (int64_t i = 0; i < n_entries; i++) eidx[(size_t)cur[(size_t)e_window[i]]++] = i;
        std::vector<int64_t> cux(xoff.begin(), xoff.end() - 1);
        for (int64_t i = 0; i < n_exc; i++) xidx[(size_t)cux[(size_t)x_window[i]]++] = i;
    }
    p->win.resize((size_t)W);
    const word_t kmask = k == 64 ? ~0ull : ((1ull << k) - 1ull);
    const double max_exp = 1 << 22;            // expansions of one exception k-mer the host is willing to enumerate
    std::atomic<int> next{0}, failed{0};       // failed: 0 or the MP_ERR_* code of the first failure
    // 32 threads by default; 96 from 2 M entries up — the entries that reach the host once the device's entropy gate has kept the
    // heaviest windows (10^6 rows: planning 4.3 ms with 32 threads, 3.0-3.7 with 64, 2.85 with 96; at 131072 rows — under 1 M entries —
    // more than 32 gain nothing: profiles/r05_exp_threads2.txt, tools/r05_threads2.sh; the threads are kept, workers.hpp)
    int n_thr = resolve_threads(P.n_threads, W);
    if (P.n_threads <= 0 && !getenv("MP_HOST_THREADS") && n_entries >= ((int64_t)2 << 20))
        n_thr = (int)std::min<int64_t>(W, std::max(n_thr, std::min(96, (int)std::thread::hardware_concurrency())));
    std::atomic<long long> us_sights{0}, us_merge{0}, us_sort{0}, us_rest{0};      // MP_TRACE: thread time per phase
    auto work = [&]() {
        Scratch scratch;
        std::vector<Sight> &sights = scratch.sights;
        for (;;) {
            int w = next.fetch_add(1);
            if (w >= W || failed.load()) break;
            if (E.skip && E.skip[w]) { p->win[(size_t)w].status = MP_WIN_ENTROPY_DEVICE; continue; }      // rejected on the device: nothing to plan
            if (E.ready) E.ready->wait_for(w); // streamed read-back: this window's entries may still be on their way
            const auto t_w0 = std::chrono::steady_clock::now();
            sights.clear();
            for (int64_t t = eoff[(size_t)w]; t < eoff[(size_t)w + 1]; t++) {
                const int64_t i = e_sorted ? t : eidx[(size_t)t];
                const word_t b0 = rd_word(e_words, (size_t)i, k), b1 = rd_word(e_words, (size_t)n_entries + (size_t)i, k),
                             g = rd_word(e_words, 2 * (size_t)n_entries + (size_t)i, k) & kmask;
                Sight s;
                s.key = key_from_words(b0, b1, g, kmask);
                s.count = E.count(i);
                s.row = E.first(i);
                s.sub = 0;
                s.ngap = __builtin_popcountll(g);
                sights.push_back(s);
            }
            int64_t n_exc_cover = 0, n_exp = 0;
            for (int64_t t = xoff[(size_t)w]; t < xoff[(size_t)w + 1]; t++) {
                const int64_t i = xidx[(size_t)t];
                const uint8_t *codes = x_codes + (size_t)i * k;
                int ngap = 0;
                for (int j = 0; j < k; j++) ngap += codes[j] == 0;
                if (ngap > P.v) {                      // gap_sequence is keyed by the raw string (V20:691)
                    Sight s;
                    for (int j = 0; j < k; j++) s.key.set(j, codes[j]);
                    s.count = 1; s.row = x_row[i]; s.sub = 0; s.ngap = ngap;
                    sights.push_back(s);
                } else {
                    if (expansions_of(codes, k) > max_exp) {
                        int zero = 0;
                        if (failed.compare_exchange_strong(zero, MP_ERR_CAPACITY))
                            pfail(p, MP_ERR_CAPACITY, "window %d, sequence %lld: an IUPAC k-mer with %.0f expansions (limit %.0f)", w,
                                  (long long)x_row[i], expansions_of(codes, k), max_exp);
                        break;
                    }
                    int32_t sub = 0;
                    n_exp += for_each_expansion(codes, k, [&](const Key &e) {
                        sights.push_back(Sight{e, 1, x_row[i], sub++, ngap});
                    });
                    n_exc_cover++;
                }
            }
            if (failed.load()) break;
            scratch.t_sights += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_w0).count();
            int rc = plan_window(p, w, scratch, n_exc_cover, n_exp, freq + (size_t)w * 4 * k, nn + (size_t)w * (k - 1) * 16);
            if (rc != MP_OK) { int zero = 0; failed.compare_exchange_strong(zero, rc); break; }
            if (!P.keep_tables && p->win[(size_t)w].status != MP_WIN_PLANNED) {
                Window &ww = p->win[(size_t)w];
                std::vector<Entry>().swap(ww.cover);
                std::vector<Entry>().swap(ww.gap);
                std::vector<int32_t>().swap(ww.cover_map.slot);
            }
        }
        us_sights += (long long)(scratch.t_sights * 1e6); us_merge += (long long)((scratch.t_merge - scratch.t_sort) * 1e6);
        us_sort += (long long)(scratch.t_sort * 1e6); us_rest += (long long)(scratch.t_gates * 1e6);
    };
    auto worker = [&]() {                      // an exception must not leave a thread (std::terminate) nor cross the C boundary
        try {
            work();
        } catch (const std::bad_alloc &) {
            int zero = 0;
            if (failed.compare_exchange_strong(zero, MP_ERR_NOMEM)) pfail(p, MP_ERR_NOMEM, "mp_plan_create: out of memory while planning the windows");
        } catch (const std::exception &e) {
            int zero = 0;
            if (failed.compare_exchange_strong(zero, MP_ERR_ARG)) pfail(p, MP_ERR_ARG, "mp_plan_create: %s", e.what());
        }
    };
    if (n_thr <= 1) worker();
    else {
        mp::run_on_threads(n_thr, [&](int) { worker(); });
    }
    if (failed.load()) return failed.load();
    if (getenv("MP_TRACE"))
        fprintf(stderr, "[mprime] plan: %d threads; thread time: sightings %.2f ms, merge %.2f ms, order %.2f ms, gates + chains %.2f ms\n", n_thr,
                us_sights.load() / 1e3, us_merge.load() / 1e3, us_sort.load() / 1e3, us_rest.load() / 1e3);
    for (int w = 0; w < W; w++) {
        Window &ww = p->win[(size_t)w];
        if (ww.status != MP_WIN_PLANNED) continue;
        p->planned.push_back(w);
        for (int s = 0; s < ww.n_seeds; s++) {
            ww.seeds[s].first_cand = p->n_cand;
            p->n_cand += (int64_t)ww.seeds[s].chain.size();
        }
    }
    return MP_OK;
}

int mp_plan_windows(const mp_plan *p, int32_t *status, int64_t *cover_number, int64_t *gap_number, double *cbit, double *tbit) {
    MP_PLAN_FORWARD(mp_plan_windows, status, cover_number, gap_number, cbit, tbit);
    if (!p) return MP_ERR_ARG;
    for (size_t w = 0; w < p->win.size(); w++) {
        const Window &x = p->win[w];
        if (status) status[w] = x.status;
        if (cover_number) cover_number[w] = x.cover_number;
        if (gap_number) gap_number[w] = x.gap_number;
        if (cbit) cbit[w] = x.cbit;
        if (tbit) tbit[w] = x.tbit;
    }
    return MP_OK;
}

int mp_plan_sizes(const mp_plan *p, int32_t *n_planned, int64_t *n_candidates) {
    MP_PLAN_FORWARD(mp_plan_sizes, n_planned, n_candidates);
    if (!p) return MP_ERR_ARG;
    if (n_planned) *n_planned = (int32_t)p->planned.size();
    if (n_candidates) *n_candidates = p->n_cand;
    return MP_OK;
}

int mp_plan_candidates(const mp_plan *p, int32_t *cand_window, uint8_t *cand_codes) {
    MP_PLAN_FORWARD(mp_plan_candidates, cand_window, cand_codes);
    if (!p || !cand_window || !cand_codes) return MP_ERR_ARG;
    const int k = p->P.k;
    int64_t c = 0;
    for (int32_t w : p->planned) {
        const Window &x = p->win[(size_t)w];
        for (int s = 0; s < x.n_seeds; s++)
            for (const Key &key : x.seeds[s].chain) {
                cand_window[c] = w;
                for (int j = 0; j < k; j++) cand_codes[(size_t)c * k + j] = (uint8_t)key.get(j);
                c++;
            }
    }
    return MP_OK;
}

int mp_plan_seeds(const mp_plan *p, int32_t w, uint8_t *nm, uint8_t *mm, int32_t *has_mm, int32_t *n_chain_nm, int32_t *n_chain_mm) {
    MP_PLAN_FORWARD(mp_plan_seeds, w, nm, mm, has_mm, n_chain_nm, n_chain_mm);
    if (!p || w < 0 || (size_t)w >= p->win.size()) return MP_ERR_ARG;
    const Window &x = p->win[(size_t)w];
    if (x.status != MP_WIN_PLANNED) return MP_ERR_ARG;
    const int k = p->P.k;
    if (nm) memcpy(nm, x.seeds[0].index, (size_t)k);
    if (mm && x.n_seeds == 2) memcpy(mm, x.seeds[1].index, (size_t)k);
    if (has_mm) *has_mm = x.n_seeds == 2;
    if (n_chain_nm) *n_chain_nm = (int32_t)x.seeds[0].chain.size();
    if (n_chain_mm) *n_chain_mm = x.n_seeds == 2 ? (int32_t)x.seeds[1].chain.size() : 0;
    return MP_OK;
}

int mp_plan_chain(const mp_plan *p, int32_t w, int32_t seed, int32_t cap, uint8_t *codes, int64_t *cov, uint8_t *stops, int32_t *n) {
    MP_PLAN_FORWARD(mp_plan_chain, w, seed, cap, codes, cov, stops, n);
    if (!p || w < 0 || (size_t)w >= p->win.size() || !n) return MP_ERR_ARG;
    const Window &x = p->win[(size_t)w];
    if (x.status != MP_WIN_PLANNED || seed < 0 || seed >= x.n_seeds) return MP_ERR_ARG;
    const Seed &s = x.seeds[seed];
    *n = (int32_t)s.chain.size();
    if (*n > cap) return MP_ERR_CAPACITY;
    const int k = p->P.k;
    for (size_t i = 0; i < s.chain.size(); i++) {
        if (codes) for (int j = 0; j < k; j++) codes[i * k + j] = (uint8_t)s.chain[i].get(j);
        if (cov) cov[i] = s.cov[i];
        if (stops) stops[i] = s.stops[i];
    }
    return MP_OK;
}

int mp_plan_finish(mp_plan *p, const int64_t *ev) {
    MP_PLAN_FORWARD_MUT(mp_plan_finish, ev);
    if (!p || !ev) return MP_ERR_ARG;
    const int k = p->P.k;
    // windows are independent: the replay and the nonsense counts (expansions of every final primer looked up in the window's table)
    // run on the host's cores; the first perfect-coverage mismatch (in window order) is the one reported
    std::atomic<size_t> next{0};
    std::atomic<int64_t> bad_at{-1};
    auto finish_window = [&](size_t pi) -> bool {
        const int32_t w = p->planned[pi];
        Window &x = p->win[(size_t)w];
        const int64_t cn = x.cover_number;
        for (int si = 0; si < x.n_seeds; si++) {
            Seed &s = x.seeds[si];
            const int64_t base = s.first_cand;
            // the host's running perfect coverage and the device's count are the same quantity
            for (size_t j = 0; j < s.chain.size(); j++)
                if (ev[3 * (base + (int64_t)j)] != s.cov[j]) return false;
            // the stopping rules of coverage_stast (V20:881-906)
            int i = 0;
            int64_t F = ev[3 * base + 1], R = ev[3 * base + 2];
            if (s.cov[0] + F < cn || s.cov[0] + R < cn) {
                while (s.cov[(size_t)i] + F < cn || s.cov[(size_t)i] + R < cn) {
                    i++;
                    F = ev[3 * (base + i) + 1];
                    R = ev[3 * (base + i) + 2];
                    if (std::max(F, R) == cn || s.stops[(size_t)i]) break;
                }
            }
            s.final_i = i;
            s.F = s.cov[(size_t)i] + F;
            s.R = s.cov[(size_t)i] + R;
        }
        const Seed *ch = &x.seeds[0];
        if (x.n_seeds == 2) {
            const Seed &nm = x.seeds[0], &mm = x.seeds[1];
            ch = (nm.F + nm.R) > (mm.F + mm.R) ? &nm : &mm;                              // V20:816: ties go to MM
        }
        x.primer = ch->chain[(size_t)ch->final_i];
        x.cov = ch->cov[(size_t)ch->final_i];
        x.f_mis = ch->F;
        x.r_mis = ch->R;
        uint8_t codes[kMaxK + 1];
        x.n_dege = 0;
        for (int j = 0; j < k; j++) { codes[j] = (uint8_t)x.primer.get(j); x.n_dege += set_size(codes[j]) > 1; }
        // nonsense_primer_number (V20:846): expansions that are neither observed k-mers nor the NM seed's phantom key
        int32_t nonsense = 0;
        for_each_expansion(codes, k, [&](const Key &e) {
            if (x.cover_map.find(e, x.cover) < 0 && !(e == x.present)) nonsense++;
        });
        x.nonsense = nonsense;
        return true;
    };
    auto work = [&]() {
        for (;;) {
            const size_t pi = next.fetch_add(1);
            if (pi >= p->planned.size()) break;
            if (!finish_window(pi)) {
                int64_t cur = bad_at.load();
                while ((cur < 0 || (int64_t)pi < cur) && !bad_at.compare_exchange_weak(cur, (int64_t)pi)) {}
            }
        }
    };
    const int n_thr = std::min(8, resolve_threads(p->P.n_threads, (int64_t)p->planned.size() / 64));      // (a thread costs ~30 us to start)
    if (n_thr <= 1) work();
    else {
        mp::run_on_threads(n_thr, [&](int) { work(); });
    }
    if (bad_at.load() >= 0) {
        const int32_t w = p->planned[(size_t)bad_at.load()];
        const Window &x = p->win[(size_t)w];
        for (int si = 0; si < x.n_seeds; si++) {
            const Seed &s = x.seeds[si];
            for (size_t j = 0; j < s.chain.size(); j++)
                if (ev[3 * (s.first_cand + (int64_t)j)] != s.cov[j])
                    return pfail(p, MP_ERR_ARG, "perfect-coverage mismatch between the host chain and the evaluation at window %d "
                                 "(seed %d, member %zu: host %lld, evaluation %lld)", w, si, j, (long long)s.cov[j],
                                 (long long)ev[3 * (s.first_cand + (int64_t)j)]);
        }
    }
    p->finished = true;
    return MP_OK;
}

int mp_plan_results(const mp_plan *p, int32_t *window, double *cbit, double *tbit, uint8_t *primer_codes, int64_t *cov,
                    int64_t *f_mis, int64_t *r_mis, int32_t *nonsense, int32_t *n_dege, int64_t *cover_number) {
    MP_PLAN_FORWARD(mp_plan_results, window, cbit, tbit, primer_codes, cov, f_mis, r_mis, nonsense, n_dege, cover_number);
    if (!p) return MP_ERR_ARG;
    if (!p->finished) return MP_ERR_ARG;
    const int k = p->P.k;
    size_t i = 0;
    for (int32_t w : p->planned) {
        const Window &x = p->win[(size_t)w];
        if (window) window[i] = w;
        if (cbit) cbit[i] = x.cbit;
        if (tbit) tbit[i] = x.tbit;
        if (primer_codes) for (int j = 0; j < k; j++) primer_codes[i * k + j] = (uint8_t)x.primer.get(j);
        if (cov) cov[i] = x.cov;
        if (f_mis) f_mis[i] = x.f_mis;
        if (r_mis) r_mis[i] = x.r_mis;
        if (nonsense) nonsense[i] = x.nonsense;
        if (n_dege) n_dege[i] = x.n_dege;
        if (cover_number) cover_number[i] = x.cover_number;
        i++;
    }
    return MP_OK;
}

int mp_plan_window_table(const mp_plan *p, int32_t w, int32_t which, int64_t cap, uint8_t *codes, int64_t *counts,
                         int64_t *first_row, int64_t *n) {
    MP_PLAN_FORWARD(mp_plan_window_table, w, which, cap, codes, counts, first_row, n);
    if (!p || w < 0 || (size_t)w >= p->win.size() || (which != 0 && which != 1) || !n) return MP_ERR_ARG;
    const Window &x = p->win[(size_t)w];
    const std::vector<Entry> &t = which == 0 ? x.cover : x.gap;
    *n = (int64_t)t.size();
    if ((int64_t)t.size() > cap) return MP_ERR_CAPACITY;
    const int k = p->P.k;
    for (size_t i = 0; i < t.size(); i++) {
        if (codes) for (int j = 0; j < k; j++) codes[i * k + j] = (uint8_t)t[i].key.get(j);
        if (counts) counts[i] = t[i].count;
        if (first_row) first_row[i] = t[i].first_row;
    }
    return MP_OK;
}

extern "C++" {
// The expansions of many short IUPAC k-mers (the exception list: one or two degenerate positions each): the member lists are walked as a
// mixed-radix counter over the DEGENERATE positions only (last position fastest — itertools.product order, V20:368-380), every step
// rewrites the positions that changed and nothing else.  `emit(i, first, changed positions...)` sees each expansion once.
template <typename Start, typename Change, typename Emit>
static void walk_expansions(const uint8_t *codes, int k, Start &&start, Change &&change, Emit &&emit) {
    int dpos[kMaxK + 1], didx[kMaxK + 1], nd = 0;
    for (int j = 0; j < k; j++)
        if (kMembers[codes[j]].n > 1) { dpos[nd] = j; didx[nd] = 0; nd++; }
    start();
    for (;;) {
        emit();
        int t = nd - 1;
        for (; t >= 0; t--) {
            const Members &mb = kMembers[codes[dpos[t]]];
            if (++didx[t] < mb.n) { change(dpos[t], mb.m[didx[t]]); break; }
            didx[t] = 0;
            change(dpos[t], mb.m[0]);
        }
        if (t < 0) break;
    }
}

// first output slot of every k-mer's expansions (first[n] = their number); MP_ERR_ARG for a bad symbol code, MP_ERR_CAPACITY beyond 9e15
// run body(i0, i1) over [0, n) on a few threads when there is enough of it (the exception list of a deep alignment: 10^4 .. 10^6 k-mers)
template <typename Body>
static void over_kmers(int64_t n, Body &&body) {
    const int n_thr = n >= 8192 ? std::min(16, resolve_threads(0, n / 4096)) : 1;
    if (n_thr <= 1) { body((int64_t)0, n); return; }
    mp::run_on_threads(n_thr, [&](int t) { body(n * t / n_thr, n * (t + 1) / n_thr); });
}

static int expansion_offsets(int32_t k, int64_t n, const uint8_t *codes, std::vector<int64_t> &first) {
    first.assign((size_t)n + 1, 0);
    // the degeneracy of every k-mer on the threads (first[i + 1] for now), the running sum behind them
    std::atomic<int> bad{0};
    std::vector<uint8_t> big((size_t)n, 0);
    over_kmers(n, [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; i++) {
            int64_t d = 1;
            for (int j = 0; j < k; j++) {
                const uint8_t c = codes[(size_t)i * k + j];
                if (c > 15) { bad.store(1); return; }
                d *= kMembers[c].n;
                if (d > ((int64_t)1 << 40)) big[(size_t)i] = 1;                   // counted as a double below ...
                if (d > ((int64_t)1 << 60)) d = (int64_t)1 << 60;                 // ... and beyond 2^60 the call fails there anyway (> 9e15)
            }
            first[(size_t)i + 1] = d;
        }
    });
    if (bad.load()) return MP_ERR_ARG;
    double need = 0;
    for (int64_t i = 0; i < n; i++) {
        need += big[(size_t)i] ? expansions_of(codes + (size_t)i * k, k) : (double)first[(size_t)i + 1];
        if (need > 9e15) return MP_ERR_CAPACITY;
        first[(size_t)i + 1] += first[(size_t)i];
    }
    return MP_OK;
}
}  // extern "C++"

int mp_expand_kmers(int32_t k, int64_t n, const uint8_t *codes, int64_t cap, uint8_t *out_codes, int64_t *out_src, int64_t *n_out) {
#if !MP_PLAN_WIDE
    if (k > kMaxK) return mp_expand_kmers_w64(k, n, codes, cap, out_codes, out_src, n_out);
#endif
    if (k < 1 || k > kMaxK || n < 0 || !n_out || (n && !codes)) return MP_ERR_ARG;
    std::vector<int64_t> first;
    int rc = expansion_offsets(k, n, codes, first);
    if (rc) return rc;
    *n_out = first[(size_t)n];
    if (*n_out > cap) return MP_ERR_CAPACITY;
    over_kmers(n, [&](int64_t i0, int64_t i1) {
        uint8_t cur[kMaxK + 1];
        for (int64_t i = i0; i < i1; i++) {
            const uint8_t *c = codes + (size_t)i * k;
            int64_t o = first[(size_t)i];
            walk_expansions(c, k, [&] { for (int j = 0; j < k; j++) cur[j] = kMembers[c[j]].m[0]; },
                            [&](int j, uint8_t member) { cur[j] = member; },
                            [&] {
                                if (out_codes) memcpy(out_codes + (size_t)o * k, cur, (size_t)k);
                                if (out_src) out_src[o] = i;
                                o++;
                            });
        }
    });
    return MP_OK;
}

int mp_expand_kmer_words(int32_t k, int64_t n, const uint8_t *codes, int64_t cap, void *out_words, int64_t *out_src, int64_t *n_out) {
#if !MP_PLAN_WIDE
    if (k > kMaxK) return mp_expand_kmer_words_w64(k, n, codes, cap, out_words, out_src, n_out);
#endif
    if (k < 1 || k > kMaxK || n < 0 || !n_out || (n && !codes)) return MP_ERR_ARG;
    std::vector<int64_t> first;
    int rc = expansion_offsets(k, n, codes, first);
    if (rc) return rc;
    *n_out = first[(size_t)n];
    if (*n_out > cap) return MP_ERR_CAPACITY;
    over_kmers(n, [&](int64_t i0, int64_t i1) {
    word_t b0 = 0, b1 = 0, g = 0;                                 // window words of mprime.h: base index bits and the gap flag
    auto put = [&](int j, uint8_t member) {                       // member: a one-hot base code, or 0 = '-'
        const word_t bit = (word_t)1 << j;
        b0 &= ~bit; b1 &= ~bit; g &= ~bit;
        if (member == 0) g |= bit;
        else {
            if (member & 10u) b0 |= bit;                          // C or T
            if (member & 12u) b1 |= bit;                          // G or T
        }
    };
    for (int64_t i = i0; i < i1; i++) {
        const uint8_t *c = codes + (size_t)i * k;
        int64_t o = first[(size_t)i];
        auto base_words = [&] {                                   // every position at its first member, branch-free
            b0 = b1 = g = 0;
            for (int j = 0; j < k; j++) {
                const uint32_t m0 = kMembers[c[j]].m[0];
                b0 |= (word_t)(((m0 >> 1) | (m0 >> 3)) & 1u) << j;
                b1 |= (word_t)(((m0 >> 2) | (m0 >> 3)) & 1u) << j;
                g |= (word_t)(m0 == 0) << j;
            }
        };
        walk_expansions(c, k, base_words, put,
                        [&] {
                            if (out_words) {
                                if (k > MP_NARROW_K) {
                                    uint64_t *ow = (uint64_t *)out_words;
                                    ow[(size_t)o * 3] = b0; ow[(size_t)o * 3 + 1] = b1; ow[(size_t)o * 3 + 2] = g;
                                } else {
                                    uint32_t *ow = (uint32_t *)out_words;
                                    ow[(size_t)o * 3] = (uint32_t)b0; ow[(size_t)o * 3 + 1] = (uint32_t)b1; ow[(size_t)o * 3 + 2] = (uint32_t)g;
                                }
                            }
                            if (out_src) out_src[o] = i;
                            o++;
                        });
    }
    });
    return MP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// JSON side files (V20:1172-1177): {out}.non_coverage_seq_id_json and {out}.gap_seq_id_json, byte for byte what
// json.dump(obj, fh, indent=4) writes for the reference's dictionaries
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct Out {                             // the text of a run of output windows
    std::string &buf;
    explicit Out(std::string &b) : buf(b) { buf.reserve(1 << 20); }
    void put(const char *s) { buf += s; }
    void put(const std::string &s) { buf += s; }
    void pad(int n) { buf.append((size_t)n, ' '); }
};

// json.encoder.encode_basestring_ascii of a str that was decoded from `n` UTF-8 bytes with errors="surrogateescape"
std::string json_quote(const uint8_t *s, size_t n) {
    std::string o = "\"";
    char tmp[16];
    auto esc = [&](uint32_t cp) {
        if (cp >= 0x10000) {                                  // UTF-16 surrogate pair
            cp -= 0x10000;
            snprintf(tmp, sizeof tmp, "\\u%04x\\u%04x", 0xd800 | ((cp >> 10) & 0x3ff), 0xdc00 | (cp & 0x3ff));
        } else snprintf(tmp, sizeof tmp, "\\u%04x", cp);
        o += tmp;
    };
    size_t i = 0;
    while (i < n) {
        const uint8_t c = s[i];
        if (c < 0x80) {
            switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            default:
                if (c < 0x20) esc(c);
                else o += (char)c;
            }
            i++;
            continue;
        }
        // decode one UTF-8 sequence; an invalid byte becomes the lone surrogate U+DC00 + byte (surrogateescape)
        int len = (c >= 0xf0 && c <= 0xf4) ? 4 : (c >= 0xe0 && c < 0xf0) ? 3 : (c >= 0xc2 && c < 0xe0) ? 2 : 0;
        uint32_t cp = 0;
        bool ok = len > 0 && i + (size_t)len <= n;
        if (ok) {
            cp = len == 2 ? (c & 0x1f) : len == 3 ? (c & 0x0f) : (c & 0x07);
            for (int t = 1; t < len; t++) {
                if ((s[i + t] & 0xc0) != 0x80) { ok = false; break; }
                cp = (cp << 6) | (s[i + t] & 0x3f);
            }
            if (ok && ((len == 3 && (cp < 0x800 || (cp >= 0xd800 && cp <= 0xdfff))) || (len == 4 && (cp < 0x10000 || cp > 0x10ffff)))) ok = false;
        }
        if (ok) { esc(cp); i += (size_t)len; }
        else { esc(0xdc00 + c); i++; }
    }
    o += '"';
    return o;
}

std::string key_string(const Key &k, int n) {
    static const char *sym = "-ACMGRSVTWYHKDBN";
    std::string s((size_t)n, '-');
    for (int j = 0; j < n; j++) s[(size_t)j] = sym[k.get(j)];
    return s;
}

struct KeyHash { size_t operator()(const Key &k) const { return (size_t)hash_key(k); } };

}  // namespace

extern "C" int mp_plan_write_side_files(const mp_plan *p, int32_t n_out, const int32_t *out_window, const int64_t *out_pos,
                                        const uint8_t *primer_codes, uint64_t strictF, uint64_t strictR, const int64_t *dev_off,
                                        const void *dev_words, int64_t n_dev, const int32_t *labels, int32_t n_rows, int64_t n_exc,
                                        const int32_t *x_window, const int64_t *x_row, const uint8_t *x_codes, const uint8_t *ids,
                                        const int64_t *id_off, const char *noncov_path, const char *gap_path) {
    return mp_plan_write_side_files_part(p, n_out, out_window, out_pos, primer_codes, strictF, strictR, dev_off, dev_words, n_dev, labels, n_rows,
                                         n_exc, x_window, x_row, x_codes, ids, id_off, noncov_path, gap_path, 3);
}

extern "C" int mp_plan_write_side_files_part(const mp_plan *p, int32_t n_out, const int32_t *out_window, const int64_t *out_pos,
                                             const uint8_t *primer_codes, uint64_t strictF, uint64_t strictR, const int64_t *dev_off,
                                             const void *dev_words, int64_t n_dev, const int32_t *labels, int32_t n_rows, int64_t n_exc,
                                             const int32_t *x_window, const int64_t *x_row, const uint8_t *x_codes, const uint8_t *ids,
                                             const int64_t *id_off, const char *noncov_path, const char *gap_path, int32_t part) {
    MP_PLAN_FORWARD(mp_plan_write_side_files_part, n_out, out_window, out_pos, primer_codes, strictF, strictR, dev_off, dev_words, n_dev, labels,
                    n_rows, n_exc, x_window, x_row, x_codes, ids, id_off, noncov_path, gap_path, part);
    const bool first_part = part & 1, last_part = part & 2;
    if (!first_part && n_out <= 0) return MP_ERR_ARG;                    // a continuation holds at least one window
    if (first_part && !last_part && n_out <= 0) return MP_ERR_ARG;
    if (!p || n_out < 0 || n_rows < 0 || !noncov_path || !gap_path || (n_out && (!out_window || !out_pos || !primer_codes || !dev_off || !labels || !ids || !id_off)))
        return MP_ERR_ARG;
    if (!p->P.keep_tables) return MP_ERR_ARG;
    const int k = p->P.k, v = p->P.v;
    const word_t kmask = k == 64 ? ~0ull : ((1ull << k) - 1ull);
    mp_plan *pm = const_cast<mp_plan *>(p);                              // the message buffer only
    FILE *fn = fopen(noncov_path, first_part ? "wb" : "ab");
    if (!fn) return pfail(pm, MP_ERR_ARG, "%s: %s", noncov_path, strerror(errno));
    FILE *fg = fopen(gap_path, first_part ? "wb" : "ab");
    if (!fg) { const int e = errno; fclose(fn); return pfail(pm, MP_ERR_ARG, "%s: %s", gap_path, strerror(e)); }
    for (int32_t oi = 0; oi < n_out; oi++)
        if (out_window[oi] < 0 || (size_t)out_window[oi] >= p->win.size()) { fclose(fn); fclose(fg); return MP_ERR_ARG; }
    // The output windows are formatted in contiguous runs on the host's cores — one pair of text buffers per run, written in order:
    // the files are O(windows x sequences) text and a single thread spent 60 % of a 500-sequence cluster's whole run() here.
    const int n_threads = resolve_threads(0, std::max<int64_t>(1, n_out / 8));
    std::vector<std::string> text_n((size_t)n_threads), text_g((size_t)n_threads);
    std::atomic<bool> oom{false};
    // exceptions grouped by window, ascending rows
    std::vector<int64_t> xorder((size_t)n_exc);
    for (int64_t i = 0; i < n_exc; i++) xorder[(size_t)i] = i;
    std::sort(xorder.begin(), xorder.end(), [&](int64_t a, int64_t b) {
        return x_window[a] != x_window[b] ? x_window[a] < x_window[b] : x_row[a] < x_row[b];
    });
    auto format_run = [&](int t) {
    try {
    const int32_t o0 = (int32_t)((int64_t)n_out * t / n_threads), o1 = (int32_t)((int64_t)n_out * (t + 1) / n_threads);
    Out on(text_n[(size_t)t]), og(text_g[(size_t)t]);
    std::unordered_map<int64_t, std::string> idq;                        // quoted ids, made on first use (per run)
    auto quoted_id = [&](int64_t r) -> const std::string & {
        auto it = idq.find(r);
        if (it == idq.end()) it = idq.emplace(r, json_quote(ids + id_off[r], (size_t)(id_off[r + 1] - id_off[r]))).first;
        return it->second;
    };
    auto ids_block = [&](Out &o, const std::vector<int64_t> &rows, int ind) {
        if (rows.empty()) { o.put("[]"); return; }
        o.put("[\n");
        for (size_t i = 0; i < rows.size(); i++) {
            o.pad(ind + 4);
            o.put(quoted_id(rows[i]));
            if (i + 1 < rows.size()) o.put(",\n");
        }
        o.put("\n");
        o.pad(ind);
        o.put("]");
    };
    for (int32_t oi = o0; oi < o1; oi++) {
        const int32_t w = out_window[oi];
        const Window &W = p->win[(size_t)w];
        const uint8_t *pc = primer_codes + (size_t)oi * k;
        // rows of every device entry of the window (labels index the entries in device order)
        const int64_t a = dev_off[w], b = dev_off[w + 1];
        std::vector<std::vector<int64_t>> rows_of_entry((size_t)(b - a));
        const int32_t *lab = labels + (size_t)oi * n_rows;
        for (int32_t r = 0; r < n_rows; r++)
            if (lab[r] >= 0 && lab[r] < b - a) rows_of_entry[(size_t)lab[r]].push_back(r);
        std::unordered_map<Key, int32_t, KeyHash> entry_of;             // k-mer -> device entry
        for (int64_t e = a; e < b; e++) {
            const word_t b0 = rd_word(dev_words, (size_t)e, k), b1 = rd_word(dev_words, (size_t)n_dev + (size_t)e, k),
                         g = rd_word(dev_words, 2 * (size_t)n_dev + (size_t)e, k) & kmask;
            Key key;
            for (int j = 0; j < k; j++) key.set(j, (g >> j) & 1u ? 0u : 1u << (uint32_t)(((b0 >> j) & 1u) | (((b1 >> j) & 1u) << 1)));
            entry_of.emplace(key, (int32_t)(e - a));
        }
        // exceptions of this window
        auto lo = std::lower_bound(xorder.begin(), xorder.end(), w, [&](int64_t i, int32_t ww) { return x_window[i] < ww; });
        auto hi = std::upper_bound(xorder.begin(), xorder.end(), w, [&](int32_t ww, int64_t i) { return ww < x_window[i]; });
        // ids of a k-mer: rows of its device entry + the exception rows (of the wanted kind) one of whose expansions it is
        auto rows_of_key = [&](const Key &key, bool gap_rows) {
            std::vector<int64_t> rows;
            auto it = entry_of.find(key);
            if (it != entry_of.end()) rows = rows_of_entry[(size_t)it->second];
            bool touched = false;
            for (auto x = lo; x != hi; ++x) {
                const uint8_t *codes = x_codes + (size_t)*x * k;
                int ngap = 0;
                bool member = true;
                for (int j = 0; j < k; j++) {
                    ngap += codes[j] == 0;
                    const uint32_t kc = key.get(j);
                    member &= codes[j] == 0 ? kc == 0 : (kc != 0 && (codes[j] & kc) == kc);
                }
                if ((ngap > v) != gap_rows || !member) continue;
                rows.push_back(x_row[*x]);
                touched = true;
            }
            if (touched) std::sort(rows.begin(), rows.end());
            return rows;
        };
        // ---- non_coverage: [ {k-mer: ids} for F, {k-mer: ids} for R ] over the cover dict in insertion order (V20:1107-1127)
        char num[32];
        snprintf(num, sizeof num, "%lld", (long long)out_pos[oi]);
        on.pad(4); on.put("\""); on.put(num); on.put("\": [\n");
        for (int side = 0; side < 2; side++) {
            const uint64_t strict = side == 0 ? strictF : strictR;
            on.pad(8);
            bool any = false;
            for (const Entry &e : W.cover) {
                uint64_t D = 0;
                int nd = 0;
                for (int j = 0; j < k; j++) {
                    const uint32_t c = e.key.get(j);
                    if (c == 0 || !(pc[j] & c)) { D |= 1ull << j; nd++; }
                }
                if (nd == 0 || !(nd > v || (D & strict))) continue;
                on.put(any ? ",\n" : "{\n");
                any = true;
                on.pad(12); on.put("\""); on.put(key_string(e.key, k)); on.put("\": ");
                ids_block(on, rows_of_key(e.key, false), 12);
            }
            if (any) { on.put("\n"); on.pad(8); on.put("}"); }
            else on.put("{}");
            on.put(side == 0 ? ",\n" : "\n");
        }
        on.pad(4); on.put("]");
        on.put(oi + 1 < n_out ? ",\n" : (last_part ? "\n}" : ""));
        // ---- gap_seq_id: expansions of the gap_sequence keys in insertion order (V20:698)
        og.pad(4); og.put("\""); og.put(num); og.put("\": ");
        std::vector<Key> gkeys;
        std::unordered_map<Key, int32_t, KeyHash> seen;
        for (const Entry &e : W.gap) {
            uint8_t codes[kMaxK + 1];
            for (int j = 0; j < k; j++) codes[j] = (uint8_t)e.key.get(j);
            for_each_expansion(codes, k, [&](const Key &x) { if (seen.emplace(x, 1).second) gkeys.push_back(x); });
        }
        if (gkeys.empty()) og.put("{}");
        else {
            og.put("{\n");
            for (size_t i = 0; i < gkeys.size(); i++) {
                og.pad(8); og.put("\""); og.put(key_string(gkeys[i], k)); og.put("\": ");
                ids_block(og, rows_of_key(gkeys[i], true), 8);
                if (i + 1 < gkeys.size()) og.put(",\n");
            }
            og.put("\n"); og.pad(4); og.put("}");
        }
        og.put(oi + 1 < n_out ? ",\n" : (last_part ? "\n}" : ""));
    }
    } catch (const std::exception &) { oom = true; }                     // bad_alloc of the id / row tables: no exception leaves the C ABI
    };
    if (n_threads == 1) format_run(0);
    else {
        mp::run_on_threads(n_threads, format_run);
    }
    if (oom) { fclose(fn); fclose(fg); return pfail(pm, MP_ERR_NOMEM, "mp_plan_write_side_files: out of memory"); }
    // a continuation starts with the separator the previous part left out
    const char *head = first_part ? (n_out ? "{\n" : "{}") : ",\n";
    fputs(head, fn); fputs(head, fg);
    for (int t = 0; t < n_threads; t++) {
        fwrite(text_n[(size_t)t].data(), 1, text_n[(size_t)t].size(), fn);
        fwrite(text_g[(size_t)t].data(), 1, text_g[(size_t)t].size(), fg);
    }
    const bool bad = ferror(fn) || ferror(fg);
    fclose(fn); fclose(fg);
    return bad ? pfail(pm, MP_ERR_ARG, "write error on %s / %s", noncov_path, gap_path) : MP_OK;
}
