// d2h_bench.hip — how fast do 58 MB come off the device into host memory of different kinds?  (tools/ubench: measurements behind
// mp_plan_create_streamed's choice of buffer; hipcc --offload-arch=gfx950 -O2 -pthread tools/ubench/d2h_bench.hip -o tools/_build/d2h_big)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 58u << 20;
    void *d = nullptr;
    hipMalloc(&d, n);
    hipMemset(d, 1, n);
    hipDeviceSynchronize();
    { void *w = malloc(1 << 20); hipMemcpy(w, d, 1 << 20, hipMemcpyDeviceToHost); free(w); }      // runtime warm-up
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        void *p = malloc(n);
        hipMemcpy(p, d, n, hipMemcpyDeviceToHost);
        printf("fresh malloc:            %.2f ms\n", now() - t0);
        t0 = now();
        hipMemcpy(p, d, n, hipMemcpyDeviceToHost);
        printf("touched malloc:          %.2f ms\n", now() - t0);
        free(p);
        t0 = now();
        void *m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(m, n, MADV_HUGEPAGE);
        hipMemcpy(m, d, n, hipMemcpyDeviceToHost);
        printf("fresh mmap + hugepage:   %.2f ms\n", now() - t0);
        munmap(m, n);
        t0 = now();
        m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
        double t1 = now();
        hipMemcpy(m, d, n, hipMemcpyDeviceToHost);
        printf("mmap populate:           %.2f ms populate + %.2f ms copy\n", t1 - t0, now() - t1);
        munmap(m, n);
        {   // what mp_plan_create_streamed could do: huge-page mapping, pages faulted in on 16 threads, then registered with the runtime
            t0 = now();
            uint8_t *m2 = (uint8_t *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            madvise(m2, n, MADV_HUGEPAGE);
            std::vector<std::thread> th;
            for (int t = 0; t < 16; t++)
                th.emplace_back([=] { for (size_t o = n * t / 16 / 4096 * 4096; o < n * (t + 1) / 16; o += 4096) ((volatile uint8_t *)m2)[o] = 0; });
            for (auto &x : th) x.join();
            double t1 = now();
            hipError_t e = hipHostRegister(m2, n, hipHostRegisterDefault);
            double t2 = now();
            hipMemcpy(m2, d, n, hipMemcpyDeviceToHost);
            double t3 = now();
            hipMemcpy(m2, d, n, hipMemcpyDeviceToHost);
            double t4 = now();
            hipHostUnregister(m2);
            double t5 = now();
            printf("hugepage+prefault %.2f ms, hipHostRegister %.2f ms (%d), copy %.2f ms (again %.2f ms), unregister %.2f ms\n", t1 - t0, t2 - t1, (int)e, t3 - t2, t4 - t3, t5 - t4);
            // unregistered, the same touched area: 5 copies of a fifth each (what mp_get_unique issues)
            t0 = now();
            for (int q = 0; q < 5; q++) hipMemcpyAsync(m2 + n / 5 * q, (uint8_t *)d + n / 5 * q, n / 5, hipMemcpyDeviceToHost, 0);
            hipStreamSynchronize(0);
            printf("touched hugepage area, 5 async copies: %.2f ms\n", now() - t0);
            munmap(m2, n);
        }
        t0 = now();
        void *h = nullptr;
        hipHostMalloc(&h, n, hipHostMallocDefault);
        t1 = now();
        hipMemcpy(h, d, n, hipMemcpyDeviceToHost);
        double t2 = now();
        hipMemcpy(h, d, n, hipMemcpyDeviceToHost);
        printf("hipHostMalloc:           %.2f ms alloc + %.2f ms copy (again: %.2f ms)\n", t1 - t0, t2 - t1, now() - t2);
        t0 = now();
        hipHostFree(h);
        printf("hipHostFree:             %.2f ms\n", now() - t0);
    }
    return 0;
}
