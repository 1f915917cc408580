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
// record: [0] header: n_slots | n_extra << 8 | kSlSimple;  [1..7] the event planes in order (n_slots = events + 1): plane row x
//         row_scale — what a fetch takes as it stands; the all-zero row for a slot the item does not have;
//         [8..15] candidate index member t reports to (-1: none);  [16..23] extra corrections (entry form below);  [24], [25] per strict
//         slot q one byte: bit i = event plane i is a SUB plane at that position;  [26] nibble t = the member slot whose counts member t
//         reports;  [27] the item's window;  [28] flags of the event planes: bit i strict forward, bit 8 + i strict reverse, bit 16 + i
//         SUB (the plane of a base beyond the reference);  [29] the window's row of the exclusion words x row_scale;
//         [30] first plane row of the window x row_scale (extras)
// extra correction entry: bits 0-6 plane row inside the window (position * 4 + base), then flags
constexpr uint32_t kSlPresent = 1u << 7, kSlSub = 1u << 8, kSlStrictF = 1u << 9, kSlStrictR = 1u << 10;
constexpr uint32_t kSlSimple = 1u << 16;      // header: every event plane is the plane of a base beyond the reference

struct SlideChainIn {                    // what the builder needs of a chain item (common.hpp ChainItem)
    int32_t win, cand0, n_steps, ev0, n_ev;
    uint32_t sym[4];                     // nibble j = symbol of the most degenerate member at position j
};

struct SlideBand { int32_t w0, n_win, item0, n_items, iter0, pad; };     // windows [w0, w0 + n_win); items [item0, item0 + n_items)

struct SlidePlan {
    std::vector<uint8_t> slides;         // per chain item of the upload: 1 = in this plan, 0 = left to the first-pass kernels
    std::vector<int32_t> rest;           // the items that are not (indices into the upload's chain items, ascending)
    std::vector<int32_t> item_of;        // plan item -> chain item
    std::vector<SlideBand> bands;
    // per band, per iteration t = -(k-1) .. n_win-1 (iter0 + 2 (t + k - 1)): {plane row of the column sliding in (col * 4 + R_col) x row_scale,
    // first item of window w0 + t | number of items << 24 (0 while warming up)}
    std::vector<uint32_t> iters;
    std::vector<uint32_t> recs;          // kSlideRec dwords per item, band after band, ascending windows
    std::vector<uint8_t> ref;            // R_c per alignment column
    int k = 0, ns = 0, max_items_band = 0;
    uint32_t spos = 0, fmask = 0, rmask = 0;     // strict slot q: position (spos >> 5 q) & 31; bit q of fmask / rmask: forward / reverse strict
};

// One record, or false when the item cannot slide: more than 8 member slots, too many corrections, a correction that is not an event
// at a strict position.  Such items stay with the first-pass kernels (SlidePlan::rest).
inline bool slide_record(const SlideChainIn &ch, const std::vector<uint32_t> &events, const std::vector<int32_t> &cand_out, int k, uint32_t sF,
                         uint32_t sR, int p0, const std::vector<uint8_t> &ref, const int (&slot_of_pos)[32], uint32_t row_scale,
                         uint32_t zero_row, uint32_t *rec) {
    auto sym = [&](int j) { return (ch.sym[j >> 3] >> (4 * (j & 7))) & 15u; };
    for (int q = 0; q < kSlideRec; q++) rec[q] = 0u;
    if (ch.n_steps < 1 || ch.n_steps > 8) return false;
    for (int m = 0; m < 8; m++) rec[8 + m] = 0xFFFFFFFFu;
    for (int q = 1; q <= kSlideKept; q++) rec[q] = zero_row * row_scale;
    const uint32_t row0 = (uint32_t)(p0 + ch.win) * 4u;
    // member slots: slot 0 = the most degenerate member; every event takes a slot (its plane is applied, then the slot is counted);
    // member t reports the counts of the slot of its step's last event — a step without events those of the member before it
    int cur = 0, e = 0;
    rec[8] = (uint32_t)cand_out[(size_t)ch.cand0];
    bool lost_seen[32 * 4] = {false};
    bool simple = true;
    for (int t = 1; t < ch.n_steps; t++) {
        while (e < ch.n_ev && (int)(events[(size_t)ch.ev0 + (size_t)e] >> 16) == t) {
            const uint32_t ev = events[(size_t)ch.ev0 + (size_t)e];
            const int j = (int)(ev & 255u), base = __builtin_ctz((ev >> 8) & 15u);
            if (++cur > kSlideKept) return false;
            rec[cur] = (row0 + (uint32_t)(j * 4 + base)) * row_scale;
            if ((sF >> j) & 1u) rec[28] |= 1u << cur;
            if ((sR >> j) & 1u) rec[28] |= 1u << (8 + cur);
            if (base != ref[(size_t)(p0 + ch.win + j)]) {                 // a base beyond the reference: its plane is a correction too
                rec[28] |= 1u << (16 + cur);
                if (slot_of_pos[j] >= 0) rec[24 + slot_of_pos[j] / 4] |= (1u << cur) << (8 * (slot_of_pos[j] & 3));
            } else {
                simple = false;
            }
            lost_seen[j * 4 + base] = true;
            e++;
        }
        rec[8 + t] = (uint32_t)cand_out[(size_t)ch.cand0 + (size_t)t];
        rec[26] |= (uint32_t)cur << (4 * t);
    }
    if (e != ch.n_ev) return false;                                       // events beyond the last step: not a chain this builder knows
    // corrections that no event covers: bases S0 accepts beyond R that are never lost (SUB), R itself where S0 does not accept it (ADD)
    int nx = 0;
    for (int j = 0; j < k; j++) {
        const uint32_t sy = sym(j);
        const int r = ref[(size_t)(p0 + ch.win + j)];
        for (int b = 0; b < 4; b++) {
            const bool in_s0 = sy >> b & 1u;
            const bool sub = in_s0 && b != r && !lost_seen[j * 4 + b], add = !in_s0 && b == r;
            if (!sub && !add) continue;
            if (slot_of_pos[j] >= 0 || nx == kSlideExtra) return false;
            rec[16 + nx++] = (uint32_t)(j * 4 + b) | kSlPresent | (sub ? kSlSub : 0u);
        }
    }
    rec[0] = (uint32_t)(cur + 1) | ((uint32_t)nx << 8) | (simple ? kSlSimple : 0u);
    rec[27] = (uint32_t)ch.win;
    rec[29] = (uint32_t)ch.win * row_scale;
    rec[30] = row0 * row_scale;
    return true;
}

// The plan of an upload.  Returns false when NOTHING slides (more than kSlideStrict strict positions, no item qualifies).
inline bool build_slide_plan(const std::vector<SlideChainIn> &chains, const std::vector<uint32_t> &events, const std::vector<int32_t> &cand_out,
                             int k, uint32_t sF, uint32_t sR, int p0, int n_cols, int band_windows, uint32_t row_scale, bool simple_only,
                             SlidePlan &P) {
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
    // records of the items that can slide, in upload order (ascending windows)
    P.slides.assign(chains.size(), 0);
    std::vector<uint32_t> rec((size_t)kSlideRec);
    for (size_t i = 0; i < chains.size(); i++) {
        if (i && chains[i].win < chains[i - 1].win) return false;        // items come in ascending windows
        if (slide_record(chains[i], events, cand_out, k, sF, sR, p0, P.ref, slot_of_pos, row_scale, (uint32_t)n_cols * 4u, rec.data()) &&
            (!simple_only || ((rec[0] & kSlSimple) && ((rec[0] >> 8) & 15u) == 0))) {
            P.slides[i] = 1;
            P.item_of.push_back((int32_t)i);
            P.recs.insert(P.recs.end(), rec.begin(), rec.end());
        } else {
            P.rest.push_back((int32_t)i);
        }
    }
    if (P.item_of.empty()) return false;
    // bands over the windows that hold sliding items
    const int B = std::max(1, band_windows);
    const size_t n = P.item_of.size();
    auto win_of = [&](size_t pi) { return chains[(size_t)P.item_of[pi]].win; };
    size_t i = 0;
    while (i < n) {
        SlideBand bd{win_of(i), 1, (int32_t)i, 0, (int32_t)P.iters.size(), 0};
        size_t j = i;
        int last = win_of(i);
        while (j < n) {
            const int w = win_of(j);
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
            uint32_t first = (uint32_t)(it - i), cnt = 0;
            if (t >= 0)
                while (it < j && win_of(it) == bd.w0 + t) { it++; cnt++; }
            if (cnt > 255) return false;
            P.iters.push_back(((uint32_t)col * 4u + P.ref[(size_t)col]) * row_scale);
            P.iters.push_back(first | (cnt << 24));
        }
        P.bands.push_back(bd);
        i = j;
    }
    return true;
}

}  // namespace mp
