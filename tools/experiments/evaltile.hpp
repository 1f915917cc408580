// evaltile.hpp — the LDS-tiled evaluation of nested refinement chains (evaltile.hip): plan structures and entry points.
#pragma once

#include "common.hpp"

namespace mp {

// One round of a band: items [item0, item0 + n_items) of the chain list, one per wave.  The alignment columns
// [new_c0, new_c1) join the ring before the round, the first of them in ring slot new_slot0.
struct TileRound { int32_t item0, n_items, new_c0, new_c1, new_slot0, pad; };
// One band of consecutive chain items: rounds [round0, round0 + n_rounds); ring slot of column c = (c - cbase) mod ring columns.
struct TileBand { int32_t round0, n_rounds, cbase, pad; };

struct TilePlan {
    std::vector<TileRound> rounds;
    std::vector<TileBand> bands;
    std::vector<uint32_t> prog;          // one fixed block per chain item (layout: evaltile.hip, tile_item)
};

int tile_words(int gw);
int tile_ring_cols(int gw, int per_cu);
void plan_tiles(const std::vector<ChainItem> &chains, const std::vector<uint32_t> &events, const std::vector<int32_t> &cand_out, int p0, int k, uint32_t sF, uint32_t sR, int gw,
                int ring_cols, int n_slices, int target_groups, TilePlan &P);
// launches eval_tile_kernel over the plan staged in the context (c->tile_*); `gw` = 32-bit row words per lane (2 or 4)
int launch_eval_tile(mp_ctx *c, int gw, unsigned long long *device_out);

}  // namespace mp
