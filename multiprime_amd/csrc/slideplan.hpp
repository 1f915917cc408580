// slideplan.hpp — host side of the sliding evaluation (evalslide.hip): the per-launch PLAN the host writes at upload time.
// Plain C++ (no HIP headers): tools/slide_emul.cpp compiles it with g++ together with slidecore.hpp and runs the very same band
// routine on the CPU against brute force.
//
// The idea (mis_primer_check + Y_distance, V20:1103-1130, 229-233, for nested refinement chains).  A window's candidates differ
// from their neighbours' in a few positions only: per alignment column c the host picks a REFERENCE base R_c (the base most chain
// items accept there).  With b_c = "row does not carry R_c at column c" (one plane fetch), the mismatch count of the all-reference
// k-mer of window p is the sliding sum cnt_p = b_p + ... + b_{p+k-1}: moving on by one window is ONE plane fetch, one add and one
// subtract on 5-bit bit-sliced counters — instead of the k (or more) plane fetches of a first pass per window.  A chain's most
// degenerate member S0 then is cnt_p minus the planes of the bases S0 accepts beyond R (rows that carry one of them mismatch R
// but match S0), plus the planes of R where S0 does not accept it; those planes are the chain's event planes anyway (every base a
// later member loses is one S0 accepts), fetched ONCE per item and kept in registers for the walk down the chain.  Per window:
// 1 + (events) plane fetches instead of ~25 + events.
//
// A band = a run of consecutive windows one wave slides through (k - 1 warm-up columns, then one column per window); the values
// b_c of the last k columns wait in an LDS ring (slide-out, strict positions).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace mp {

constexpr int kSlideRec = 32;            // dwords per item record
constexpr int kSlideKept = 7;            // event planes held in registers: entries 1..7, entry i is applied before member slot i is counted
constexpr int kSlideExtra = 8;           // corrections that are not events (fetched, applied, dropped)
constexpr int kSlideStrict = 6;          // strict positions with a slot
// record: [0] header: n_slots | n_extra << 8;  [1..7] kept entries;  [8..15] output slot (candidate index, -1: none) of member slot m;
//         [16..23] extra corrections;  [24], [25] per strict slot q one byte: bit i = kept entry i is a SUB plane at that position
// entry:  bits 0-6 plane row inside the window (position * 4 + base), then flags
constexpr uint32_t kSlPresent = 1u << 7, kSlSub = 1u << 8, kSlStrictF = 1u << 9, kSlStrictR = 1u << 10;

struct SlideChainIn {                    // what the builder needs of a chain item (common.hpp ChainItem)
    int32_t win, cand0, n_steps, ev0, n_ev;
    uint32_t sym[4];                     // nibble j = symbol of the most degenerate member at position j
};

struct SlideBand { int32_t w0, n_win, item0, n_items, iter0, pad; };     // windows [w0, w0 + n_win); items [item0, item0 + n_items)

struct SlidePlan {
    std::vector<SlideBand> bands;
    // per band, per iteration t = -(k-1) .. n_win-1 (iter0 + 2 (t + k - 1)): {plane row of the column sliding in (col * 4 + R_col),
    // first item of window w0 + t | number of items << 24 (0 while warming up)}
    std::vector<uint32_t> iters;
    std::vector<uint32_t> recs;          // kSlideRec dwords per item, band after band, ascending windows
    std::vector<uint8_t> ref;            // R_c per alignment column
    int k = 0, ns = 0, max_items_band = 0;
    uint32_t spos = 0, fmask = 0, rmask = 0;     // strict slot q: position (spos >> 5 q) & 31; bit q of fmask / rmask: forward / reverse strict
};

// Returns false when some item cannot slide (more than 8 member slots, too many corrections, a correction that is not an event at a
// strict position, more than kSlideStrict strict positions): the caller keeps the first-pass kernels for this upload.
inline bool build_slide_plan(const std::vector<SlideChainIn> &chains, const std::vector<uint32_t> &events, const std::vector<int32_t> &cand_out,
                             int k, uint32_t sF, uint32_t sR, int p0, int n_cols, int band_windows, SlidePlan &P) {
    P = SlidePlan();
    P.k = k;
    if (k < 2 || k > 31 || chains.empty()) return false;
    const uint32_t kmask = (1u << k) - 1u;
    int slot_of_pos[32];
    for (int j = 0; j < 32; j++) slot_of_pos[j] = -1;
    for (int j = 0; j < k; j++)
        if (((sF | sR) & kmask) >> j & 1u) {
            if (P.ns == kSlideStrict) return false;
            slot_of_pos[j] = P.ns;
            P.spos |= (uint32_t)j << (5 * P.ns);
            if ((sF >> j) & 1u) P.fmask |= 1u << P.ns;
            if ((sR >> j) & 1u) P.rmask |= 1u << P.ns;
            P.ns++;
        }
    auto sym = [](const SlideChainIn &ch, int j) { return (ch.sym[j >> 3] >> (4 * (j & 7))) & 15u; };
    // reference base per column: the base most items accept there (ties: the lowest)
    std::vector<int32_t> votes((size_t)n_cols * 4, 0);
    for (const SlideChainIn &ch : chains) {
        if (ch.win < 0 || p0 + ch.win + k > n_cols) return false;
        for (int j = 0; j < k; j++)
            for (int b = 0; b < 4; b++)
                if (sym(ch, j) >> b & 1u) votes[(size_t)(p0 + ch.win + j) * 4 + b]++;
    }
    P.ref.assign((size_t)n_cols, 0);
    for (int c = 0; c < n_cols; c++) {
        int best = 0;
        for (int b = 1; b < 4; b++)
            if (votes[(size_t)c * 4 + b] > votes[(size_t)c * 4 + best]) best = b;
        P.ref[(size_t)c] = (uint8_t)best;
    }
    // records
    P.recs.assign(chains.size() * (size_t)kSlideRec, 0u);
    for (size_t i = 0; i < chains.size(); i++) {
        const SlideChainIn &ch = chains[i];
        if (i && ch.win < chains[i - 1].win) return false;               // items come in ascending windows
        if (ch.n_steps < 1 || ch.n_steps > 8) return false;
        uint32_t *rec = P.recs.data() + i * (size_t)kSlideRec;
        for (int m = 0; m < 8; m++) rec[8 + m] = 0xFFFFFFFFu;
        // member slots: slot 0 = the most degenerate member; every event of step t takes a slot, the step's member is counted after its
        // last event; a step without events still takes a (plane-less) slot
        int cur = 0, e = 0;
        rec[8] = (uint32_t)cand_out[(size_t)ch.cand0];
        bool lost_seen[32 * 4] = {false};
        for (int t = 1; t < ch.n_steps; t++) {
            int n_here = 0;
            while (e < ch.n_ev && (int)(events[(size_t)ch.ev0 + (size_t)e] >> 16) == t) {
                const uint32_t ev = events[(size_t)ch.ev0 + (size_t)e];
                const int j = (int)(ev & 255u), base = __builtin_ctz((ev >> 8) & 15u);
                if (++cur > kSlideKept) return false;
                uint32_t en = (uint32_t)(j * 4 + base) | kSlPresent;
                if ((sF >> j) & 1u) en |= kSlStrictF;
                if ((sR >> j) & 1u) en |= kSlStrictR;
                if (base != P.ref[(size_t)(p0 + ch.win + j)]) {           // a base beyond the reference: its plane is a correction too
                    en |= kSlSub;
                    if (slot_of_pos[j] >= 0) rec[24 + slot_of_pos[j] / 4] |= (1u << cur) << (8 * (slot_of_pos[j] & 3));
                }
                lost_seen[j * 4 + base] = true;
                rec[cur] = en;
                e++; n_here++;
            }
            if (!n_here && ++cur > kSlideKept) return false;
            rec[8 + cur] = (uint32_t)cand_out[(size_t)ch.cand0 + (size_t)t];
        }
        if (e != ch.n_ev) return false;                                  // events beyond the last step: not a chain this builder knows
        // corrections that no event covers: bases S0 accepts beyond R that are never lost (SUB), R itself where S0 does not accept it (ADD)
        int nx = 0;
        for (int j = 0; j < k; j++) {
            const uint32_t sy = sym(ch, j);
            const int r = P.ref[(size_t)(p0 + ch.win + j)];
            for (int b = 0; b < 4; b++) {
                const bool in_s0 = sy >> b & 1u;
                const bool sub = in_s0 && b != r && !lost_seen[j * 4 + b], add = !in_s0 && b == r;
                if (!sub && !add) continue;
                if (slot_of_pos[j] >= 0 || nx == kSlideExtra) return false;
                rec[16 + nx++] = (uint32_t)(j * 4 + b) | kSlPresent | (sub ? kSlSub : 0u);
            }
        }
        rec[0] = (uint32_t)(cur + 1) | ((uint32_t)nx << 8);
    }
    // bands over the windows that hold items
    const int B = std::max(1, band_windows);
    size_t i = 0;
    while (i < chains.size()) {
        SlideBand bd{chains[i].win, 1, (int32_t)i, 0, (int32_t)P.iters.size(), 0};
        size_t j = i;
        int last = chains[i].win;
        while (j < chains.size()) {
            const int w = chains[j].win;
            if (w - bd.w0 + 1 > B || w - last > k - 1) break;           // a gap of k windows or more: a fresh warm-up is cheaper
            last = w;
            j++;
        }
        bd.n_win = last - bd.w0 + 1;
        bd.n_items = (int32_t)(j - i);
        P.max_items_band = std::max(P.max_items_band, (int)bd.n_items);
        size_t it = i;
        for (int t = -(k - 1); t < bd.n_win; t++) {
            const int col = p0 + bd.w0 + t + k - 1;
            uint32_t first = (uint32_t)(it - i), n = 0;
            if (t >= 0)
                while (it < j && chains[it].win == bd.w0 + t) { it++; n++; }
            if (n > 255) return false;
            P.iters.push_back((uint32_t)col * 4u + P.ref[(size_t)col]);
            P.iters.push_back(first | (n << 24));
        }
        P.bands.push_back(bd);
        i = j;
    }
    return true;
}

}  // namespace mp
