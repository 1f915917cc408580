// planstream.hpp — internal to libmprime_hip.so: the planning stage started while the histogram entries are still coming off the device.
#pragma once

#include <condition_variable>
#include <mutex>

#include "../../include/mprime_host.h"

// windows below `ready` have arrived; the planning threads sleep on it (a spinning pool would take the cores the copies need)
struct mp_ready_gate {
    std::mutex m;
    std::condition_variable cv;
    int ready = 0;
    void raise(int w) { { std::lock_guard<std::mutex> g(m); ready = w; } cv.notify_all(); }
    void wait_for(int w) { std::unique_lock<std::mutex> g(m); cv.wait(g, [&] { return ready > w; }); }
};

extern "C" {
// mp_plan_create_segments with the entries of window w valid only once the gate stands above w: `ready` points to an mp_ready_gate that
// the copying thread of mp_plan_create_streamed (unique.hip) raises band by band (null: everything is there).  `skip` (null: none):
// skip[w] != 0 marks a window the entropy gate already rejected on the device (mp_set_entropy_gate): it has no entries, is not planned
// and comes out as MP_WIN_ENTROPY_DEVICE.  hostplan.cpp
int mp_plan_create_segments_ready(const mp_plan_params *params, const int64_t *e_off, const void *e_words, const int32_t *e_count,
                                  const int32_t *e_first, int64_t row_base, int64_t n_exc, const int32_t *x_window, const int64_t *x_row,
                                  const uint8_t *x_codes, const int64_t *freq, const int64_t *nn, const void *ready, const uint8_t *skip,
                                  mp_plan **out);
}
