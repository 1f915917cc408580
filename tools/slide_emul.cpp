// slide_emul.cpp — CPU check of the sliding evaluation's plan builder and band routine (csrc/slideplan.hpp, csrc/slidecore.hpp), the
// code evalslide.hip runs per lane, against brute force.  No GPU, no HIP: g++ -O2 -std=c++17 tools/slide_emul.cpp -o slide_emul.
// Random alignments (A, C, G, T, gap), random exclusion masks, random NESTED chains per window — with several events in one step,
// steps without events, members that drop the column's reference base, most-degenerate members that do not accept it, strict
// positions anywhere, several chains per window, windows without chains and gaps between them.  Exit status 0 = every count equal.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../multiprime_amd/csrc/slidecore.hpp"

using namespace mp;

struct Case {
    int n_rows, n_cols, k, v, p0, W, nw32;
    std::vector<uint8_t> rows;                 // [n_rows][n_cols] 0..3 base, 4 gap
    std::vector<uint32_t> cols;                // [n_cols][4][nw32]
    std::vector<uint32_t> valid;               // [W][nw32]
    std::vector<SlideChainIn> chains;
    std::vector<uint32_t> events;
    std::vector<int32_t> cand_out;
    std::vector<std::vector<uint8_t>> members; // per candidate: k symbols (bit sets)
    std::vector<int> cand_win;
    uint32_t sF, sR;
};

template <int GW>
struct HostEnv {
    const Case &C;
    const SlidePlan &P;
    int word0;
    std::vector<uint32_t> ring;
    std::vector<long long> &out;               // [n_cand][3] perfect, forward raw, reverse raw
    HostEnv(const Case &c, const SlidePlan &p, int w0, std::vector<long long> &o) : C(c), P(p), word0(w0), ring((size_t)c.k * GW, 0u), out(o) {}
    SlideBand band{};
    SlideBand uband(int b) { band = P.bands[(size_t)b]; return band; }
    int it_base = 0;
    void load_iters(int idx) { it_base = idx; }
    uint32_t iter_word(int j) const { return P.iters[(size_t)(it_base + j)]; }
    typedef int Rec;
    Rec load_rec(int item) const { return item; }
    uint32_t rec_word(Rec item, int q) const { return P.recs[(size_t)item * kSlideRec + (size_t)q]; }
    uint32_t rec_word_dyn(Rec item, int q) const { return rec_word(item, q); }
    void fetch(uint32_t row, uint32_t (&d)[GW]) const {
        for (int i = 0; i < GW; i++) d[i] = word0 + i < C.nw32 ? C.cols[(size_t)row * C.nw32 + (size_t)(word0 + i)] : 0u;
    }
    void fetch_event(uint32_t row, uint32_t (&d)[GW]) const { fetch(row, d); }
    void valid_of(uint32_t win, uint32_t (&v)[GW]) const {
        for (int i = 0; i < GW; i++) v[i] = word0 + i < C.nw32 ? C.valid[(size_t)win * C.nw32 + (size_t)(word0 + i)] : 0u;
    }
    void stamp(int) const {}
    void progress(int) const {}
    void ring_zero(int k) { for (int i = 0; i < k * GW; i++) ring[(size_t)i] = 0u; }
    void ring_write(int slot, const uint32_t (&in)[GW]) {
        for (int i = 0; i < GW; i++) ring[(size_t)slot * GW + i] = in[i];
    }
    void ring_read(int slot, uint32_t (&o)[GW]) const {
        for (int i = 0; i < GW; i++) o[i] = ring[(size_t)slot * GW + i];
    }
    void commit(int done, const uint32_t (&accPF)[8], const uint32_t (&accR)[4]) {
        const int item = band.item0 + done;
        const long long rows = 32 * GW;                      // what this "lane" covers; the counts are of the rows that are OUT
        for (int t = 0; t < 8; t++) {
            const int32_t oc = (int32_t)rec_word(item, 8 + t);
            if (oc < 0) continue;
            const int s = (int)((rec_word(item, 26) >> (4 * t)) & 15u);
            const long long out1 = accPF[s] & 0xFFFFu, outF = accPF[s] >> 16, outR = (accR[s >> 1] >> (16 * (s & 1))) & 0xFFFFu;
            out[(size_t)oc * 3] += rows - out1;
            out[(size_t)oc * 3 + 1] += rows - outF;          // raw (includes the perfect rows), as brute() counts
            out[(size_t)oc * 3 + 2] += rows - outR;
        }
    }
};

static void make_case(Case &C, std::mt19937 &rng, int trial) {
    auto U = [&](int n) { return (int)(rng() % (unsigned)n); };
    C.k = 2 + U(30);
    if (trial % 3 == 0) C.k = 18;
    C.v = U(4);
    C.n_rows = 40 + U(900);
    C.nw32 = (C.n_rows + 31) / 32;
    C.n_cols = C.k + 20 + U(120);
    C.p0 = U(5);
    C.W = C.n_cols - C.p0 - C.k + 1 - U(3);
    if (C.W < 1) C.W = 1;
    std::vector<uint8_t> root((size_t)C.n_cols);
    for (auto &x : root) x = (uint8_t)U(4);
    C.rows.assign((size_t)C.n_rows * C.n_cols, 0);
    const int p_sub = 2 + U(25), p_gap = U(6);
    for (int r = 0; r < C.n_rows; r++)
        for (int c = 0; c < C.n_cols; c++) {
            uint8_t b = root[(size_t)c];
            if (U(100) < p_sub) b = (uint8_t)U(4);
            if (U(100) < p_gap) b = 4;
            C.rows[(size_t)r * C.n_cols + c] = b;
        }
    C.cols.assign(((size_t)C.n_cols * 4 + 1) * C.nw32, 0u);               // + the all-zero row
    for (int r = 0; r < C.n_rows; r++)
        for (int c = 0; c < C.n_cols; c++) {
            const uint8_t b = C.rows[(size_t)r * C.n_cols + c];
            if (b < 4) C.cols[((size_t)c * 4 + b) * C.nw32 + (size_t)(r >> 5)] |= 1u << (r & 31);
        }
    C.valid.assign((size_t)C.W * C.nw32, 0u);
    for (int w = 0; w < C.W; w++)
        for (int r = 0; r < C.n_rows; r++)
            if (U(100) >= 7) C.valid[(size_t)w * C.nw32 + (size_t)(r >> 5)] |= 1u << (r & 31);
    const uint32_t kmask = (1u << C.k) - 1u;
    C.sF = C.sR = 0;
    const int n_strict = U(5);
    for (int i = 0; i < n_strict; i++) { const int j = U(C.k); if (U(2)) C.sF |= 1u << j; if (U(2)) C.sR |= 1u << j; }
    if (trial % 3 == 0) { C.sF = (1u << 2) | (1u << 3); C.sR = (1u << 2) | (1u << (C.k - 3)) | (1u << (C.k - 2)); }
    C.sF &= kmask; C.sR &= kmask;
    C.chains.clear(); C.events.clear(); C.cand_out.clear(); C.members.clear(); C.cand_win.clear();
    const int skip_pct = U(40);
    for (int w = 0; w < C.W; w++) {
        if (U(100) < skip_pct) continue;
        if (U(25) == 0) { w += C.k + U(6); if (w >= C.W) break; }                   // a gap: the next band warms up afresh
        const int n_chains = 1 + (U(5) == 0);
        for (int ci = 0; ci < n_chains; ci++) {
            const int n = 1 + U(8);
            // the LAST member is the seed; walking back, every step adds bases (so the first member is the most degenerate)
            std::vector<std::vector<uint8_t>> mem((size_t)n, std::vector<uint8_t>((size_t)C.k));
            for (int j = 0; j < C.k; j++) {
                uint8_t s = (uint8_t)(1u << root[(size_t)(C.p0 + w + j)]);
                if (trial % 2 && U(12) == 0) s = (uint8_t)(1u << U(4));               // a seed that is not the consensus
                if (U(20) == 0) s |= (uint8_t)(1u << U(4));
                mem[(size_t)n - 1][(size_t)j] = s;
            }
            int budget = 7;
            for (int t = n - 2; t >= 0; t--) {
                mem[(size_t)t] = mem[(size_t)t + 1];
                int adds = U(6) == 0 ? 0 : (U(5) == 0 ? 2 : 1);
                for (int a = 0; a < adds && budget > 0; a++) {
                    const int j = U(C.k);
                    const uint8_t bit = (uint8_t)(1u << U(4));
                    if (mem[(size_t)t][(size_t)j] & bit) continue;
                    mem[(size_t)t][(size_t)j] |= bit;
                    budget--;
                }
            }
            SlideChainIn ch{w, (int32_t)C.cand_out.size(), n, (int32_t)C.events.size(), 0, {0u, 0u, 0u, 0u}};
            for (int j = 0; j < C.k; j++) ch.sym[j >> 3] |= (uint32_t)mem[0][(size_t)j] << (4 * (j & 7));
            for (int t = 1; t < n; t++)
                for (int j = 0; j < C.k; j++) {
                    const uint32_t lost = mem[(size_t)t - 1][(size_t)j] & ~mem[(size_t)t][(size_t)j];
                    for (uint32_t bit = 1; bit < 16; bit <<= 1)
                        if (lost & bit) C.events.push_back((uint32_t)j | (bit << 8) | ((uint32_t)t << 16));
                }
            ch.n_ev = (int32_t)C.events.size() - ch.ev0;
            C.chains.push_back(ch);
            for (int t = 0; t < 8; t++) {                                             // 8 padded slots per item, as mp_eval_upload lays them out
                if (t < n) { C.cand_out.push_back((int32_t)C.members.size()); C.members.push_back(mem[(size_t)t]); C.cand_win.push_back(w); }
                else C.cand_out.push_back(-1);
            }
        }
    }
}

static void brute(const Case &C, std::vector<long long> &out) {
    out.assign(C.members.size() * 3, 0);
    for (size_t c = 0; c < C.members.size(); c++) {
        const int w = C.cand_win[c];
        for (int r = 0; r < C.n_rows; r++) {
            if (!(C.valid[(size_t)w * C.nw32 + (size_t)(r >> 5)] >> (r & 31) & 1u)) continue;
            int mm = 0;
            bool hf = false, hr = false;
            for (int j = 0; j < C.k; j++) {
                const uint8_t b = C.rows[(size_t)r * C.n_cols + (size_t)(C.p0 + w + j)];
                const bool miss = b == 4 || !(C.members[c][(size_t)j] >> b & 1u);
                if (miss) { mm++; if (C.sF >> j & 1u) hf = true; if (C.sR >> j & 1u) hr = true; }
            }
            if (mm == 0) out[c * 3]++;
            if (mm <= C.v && !hf) out[c * 3 + 1]++;                                   // raw: includes the perfect rows
            if (mm <= C.v && !hr) out[c * 3 + 2]++;
        }
    }
}

template <int LV, int GW>
static void run_plan(const Case &C, const SlidePlan &P, std::vector<long long> &out, bool only_simple, bool use_valid, bool fast) {
    out.assign(C.members.size() * 3, 0);
    SlideArgs A{P.bands.data(), P.iters.data(), P.recs.data(), P.k, C.p0, P.ns, P.spos, P.fmask, P.rmask, 1u, 0u, 0u};
    if (fast && (!only_simple || !slide_strict_lists(C.k, C.sF, C.sR, A.fpos, A.rpos))) { fprintf(stderr, "fast form asked for a plan it does not serve\n"); exit(3); }
    for (size_t b = 0; b < P.bands.size(); b++)
        for (int w0 = 0; w0 < C.nw32; w0 += GW) {
            HostEnv<GW> env(C, P, w0, out);
            if (fast && !use_valid) slide_band<LV, GW, true, false, true>(env, A, (int)b);          // the GPU kernel's form
            else if (fast) slide_band<LV, GW, true, true, true>(env, A, (int)b);
            else if (!use_valid) slide_band<LV, GW, true, false>(env, A, (int)b);
            else if (only_simple) slide_band<LV, GW, true, true>(env, A, (int)b);
            else slide_band<LV, GW, false, true>(env, A, (int)b);
        }
}

int main(int argc, char **argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937 rng(argc > 2 ? (unsigned)atoi(argv[2]) : 12345u);
    int slid = 0, refused = 0, n_fast = 0;
    long long items_slid = 0, items_rest = 0;
    for (int trial = 0; trial < trials; trial++) {
        Case C;
        make_case(C, rng, trial);
        if (C.chains.empty()) continue;
        SlidePlan P;
        const int B = 1 + (int)(rng() % 40);
        const bool only_simple = trial & 1;
        const bool use_valid = !only_simple || (trial & 2) || C.k <= C.v;             // without: the GPU kernel's form, every row of the alignment counts
        if (!use_valid)
            for (auto &w : C.valid) w = 0xFFFFFFFFu;                       // (rows past n_rows are all gaps: they never reach a count)
        if (!build_slide_plan(C.chains, C.events, C.cand_out, C.k, C.sF, C.sR, C.p0, C.n_cols, B, 1u, only_simple, P)) { refused++; continue; }
        P.iters.resize(P.iters.size() + 64, 0u);                          // as upload_eval_slide pads it
        slid++;
        std::vector<long long> want, got, got_fast;
        brute(C, want);
        const int gw = 1 << (int)(rng() % 3);
        // the strict positions as two-bit counts per side (slidecore.hpp FAST): wherever that form applies it runs too, beside the per-position form
        uint32_t fp, rp;
        const bool fast = only_simple && slide_strict_lists(C.k, C.sF, C.sR, fp, rp);
#define RUN(LV, G, F) (gw == 1 ? run_plan<LV, 1>(C, P, G, only_simple, use_valid, F) : (gw == 2 ? run_plan<LV, 2>(C, P, G, only_simple, use_valid, F) : run_plan<LV, 4>(C, P, G, only_simple, use_valid, F)))
        for (int f = 0; f <= (fast ? 1 : 0); f++) {
            std::vector<long long> &g = f ? got_fast : got;
            switch (C.v) {
                case 0: RUN(1, g, f); break;
                case 1: RUN(2, g, f); break;
                case 2: RUN(3, g, f); break;
                default: RUN(4, g, f); break;
            }
        }
#undef RUN
        if (fast) { n_fast++; if (got_fast != got) { fprintf(stderr, "trial %d: k=%d v=%d sF=%x sR=%x: the two strict forms differ\n", trial, C.k, C.v, C.sF, C.sR); return 1; } }
        // candidates of the items the builder left to the first-pass kernels are not the plan's to count
        for (size_t ci = 0; ci < C.chains.size(); ci++) {
            if (P.slides[ci]) { items_slid++; continue; }
            items_rest++;
            for (int t = 0; t < C.chains[ci].n_steps; t++) {
                const int32_t oc = C.cand_out[(size_t)C.chains[ci].cand0 + (size_t)t];
                for (int r = 0; r < 3; r++) want[(size_t)oc * 3 + r] = 0;
            }
        }
        if (want != got) {
            size_t bad = 0;
            for (size_t i = 0; i < want.size(); i++)
                if (want[i] != got[i]) { bad = i; break; }
            fprintf(stderr, "trial %d: k=%d v=%d rows=%d bands=%zu B=%d: candidate %zu counter %zu: brute %lld, plan %lld\n", trial, C.k, C.v,
                    C.n_rows, P.bands.size(), B, bad / 3, bad % 3, want[bad], got[bad]);
            return 1;
        }
    }
    printf("slide_emul: %d cases equal to brute force (%lld items slid, %lld left to the first-pass kernels; %d cases also in the two-bit strict form), %d cases without a plan\n",
           slid, items_slid, items_rest, n_fast, refused);
    return slid > 0 ? 0 : 2;
}
