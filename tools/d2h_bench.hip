// Latency of small device-to-host / host-to-device copies into pageable and page-locked memory (what the library's result
// read-backs cost).  hipcc --offload-arch=gfx950 -O2 tools/d2h_bench.hip -o tools/_build/d2h_bench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    const size_t sizes[] = {4, 4096, 131072, 1 << 20, 12 << 20, 128 << 20};
    void *dev;
    hipMalloc(&dev, 128 << 20);
    hipMemset(dev, 1, 128 << 20);
    void *pinned;
    hipHostMalloc(&pinned, 128 << 20, hipHostMallocDefault);
    for (size_t n : sizes) {
        for (int mode = 0; mode < 3; mode++) {       // 0 fresh pageable, 1 touched pageable, 2 pinned
            double best[2] = {1e9, 1e9};
            for (int rep = 0; rep < 5; rep++) {
                void *host = mode == 2 ? pinned : malloc(n);
                if (mode == 1) for (size_t i = 0; i < n; i += 4096) ((volatile char *)host)[i] = 0;
                double t0 = now_ms();
                hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, st);
                hipStreamSynchronize(st);
                double t1 = now_ms();
                hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, st);
                hipStreamSynchronize(st);
                double t2 = now_ms();
                if (t1 - t0 < best[0]) best[0] = t1 - t0;
                if (t2 - t1 < best[1]) best[1] = t2 - t1;
                if (mode != 2) free(host);
            }
            printf("{\"bytes\": %zu, \"host\": \"%s\", \"d2h_ms\": %.4f, \"h2d_ms\": %.4f}\n", n,
                   mode == 0 ? "pageable fresh" : mode == 1 ? "pageable touched" : "pinned", best[0], best[1]);
        }
    }
    return 0;
}
