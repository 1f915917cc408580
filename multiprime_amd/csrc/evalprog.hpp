// evalprog.hpp — fetch programs of the chain items (evalprog.hip): layout constants and entry points.
//
// One block of kProgRegs x 64 32-bit entries per chain item, register q = entries [64 q, 64 q + 64), one entry per lane:
//   register 0: lanes 0-23 the output slot (candidate index or -1) of counter lane / 3 — where the commit's lanes look for it;
//               lanes 32-40 the header {win, n_steps, n_ev, nA, nB, nC, nD, nE, wide}
//   registers 1, 2: the fetches of the first pass (the most degenerate member over all k positions), one per base of a position's
//               symbol, grouped by what their consumption needs: nA single-base positions without a strict position, nB with
//               one; nC two-base positions (two entries each) without, nD with; then nE entries of the positions with three or
//               four bases (kMore = further bases of this position follow).  wide = 1: more than 64 entries or events (two registers)
//   registers 3, 4: the events (one lost base each), ascending by step
// entry = plane row (window position * 4 + base) | strict-position flags | chain step << 24 (events)
#pragma once

#include "common.hpp"
#include "bitslice.hpp"

namespace mp {

constexpr int kProgRegs = 5;
constexpr uint32_t kRowMask = 0x7Fu, kStrictF = 1u << 20, kStrictR = 1u << 21, kMore = 1u << 28;
// bits 8-11: a slot of the wave's LDS scratch; kKeepNext (first entry of a two-base position): park the NEXT entry's plane there;
// kKeepThis: park this entry's plane (positions with three or four bases) / an event: the plane waits in that slot
constexpr uint32_t kKeepNext = 1u << 12, kKeepThis = 1u << 13;
constexpr int kProgShapes = 14;
constexpr int kProgWords[kProgShapes] = {2, 2, 2, 4, 4, 1, 8, 8, 8, 16, 4, 4, 8, 4};      // row words per thread of a shape
constexpr int kProgKeep[kProgShapes] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 4, 4, 4};       // LDS slots per wave for parked event planes

void build_eval_programs(const std::vector<ChainItem> &chains, const std::vector<uint32_t> &events, const std::vector<int32_t> &cand_out,
                         int k, uint32_t sF, uint32_t sR, int keep_slots, std::vector<uint32_t> &prog);
int launch_eval_prog(mp_ctx *c, int shape, const BlockMap &bm, const PatchArgs &pa, unsigned grid, unsigned long long *device_out);

}  // namespace mp
