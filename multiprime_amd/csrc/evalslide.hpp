// evalslide.hpp — the sliding evaluation of nested refinement chains (evalslide.hip; plan: slideplan.hpp, band routine: slidecore.hpp).
#pragma once

#include "common.hpp"
#include "slideplan.hpp"

namespace mp {

// Builds the plan for the staged chain items and uploads it; c->slide_items = 0 when nothing slides (the first-pass kernels keep
// every item).  Items the plan leaves out are collected in c->chain_rest for the first-pass kernel.
int upload_eval_slide(mp_ctx *c, const std::vector<ChainItem> &chains, const std::vector<uint32_t> &events, const std::vector<int32_t> &cand_out);
struct EvalChainArgs;
// `patch` (may be null) = the patch units of the same step (eval.hip: positive and subtracting run), `patch_blocks` workgroups of
// eval_chain_block<LV, 8, 4>: they run as the tail of the sliding kernel's own grid instead of a launch of their own.
// `clear` (may be null): n_clear counters the launch sets to zero beside its work (mp_eval_launch_rotating).
// `patch_words`: the shape the patch units were laid out for (patch_args): 1 or 8 words per lane.
int launch_eval_slide(mp_ctx *c, unsigned long long *device_out, const EvalChainArgs *patch, int patch_blocks, int patch_words,
                      unsigned long long *clear, uint32_t n_clear);
void free_slide(mp_ctx *c);

}  // namespace mp
