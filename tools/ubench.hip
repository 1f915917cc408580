// ubench.hip — ceilings the evaluation kernel is priced against (bench.py `roofline`), measured on the box itself:
//   valu   issue rate of the integer VALU instructions the bit-sliced kernels are made of (v_bitop3_b32, v_bcnt_u32_b32,
//          v_and_or_b32, v_add_u32, v_alignbit_b32) in wave-instructions per cycle per SIMD, at 1 .. 8 waves per SIMD;
//          decides "2 or 4 cycles per wave64 instruction" 
//   l2     read bandwidth out of the eight L2s (per-XCD regions that fit 4 MiB), dword and dwordx4 loads
//   mall   read bandwidth out of the Infinity Cache (a 128 MiB buffer re-read by every workgroup)
//   hbm    read bandwidth of a 4 GiB buffer read once
// Standalone: hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench tools/ubench.hip ; prints one JSON object.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int kUnroll = 8;      // independent dependency chains per lane

// OP: 0 v_bitop3_b32  1 v_bcnt_u32_b32 (accumulating form)  2 v_and_or_b32  3 v_add_u32  4 v_alignbit_b32  5 v_xor_b32
template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(uint32_t *out, int iters, uint32_t a, uint32_t b, long long *clk) {
    uint32_t x[kUnroll];
    for (int i = 0; i < kUnroll; i++) x[i] = threadIdx.x * 2654435761u + i * 40503u + a;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < kUnroll; i++) {
            if (OP == 0) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xf8" : "+v"(x[i]) : "v"(a), "v"(b));
            if (OP == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
            if (OP == 2) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (OP == 4) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (OP == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    uint32_t s = 0;
    for (int i = 0; i < kUnroll; i++) s ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// every workgroup walks `region_words` u32 words starting at its XCD's region (workgroup b runs on XCD b % 8)
template <int VEC>
__global__ __launch_bounds__(256) void read_kernel(const uint32_t *__restrict__ buf, size_t region_words, int n_regions,
                                                   int passes, size_t start_stride, uint32_t *out) {
    const uint32_t *reg = buf + (size_t)(blockIdx.x % n_regions) * region_words;
    const size_t n = region_words / VEC;
    size_t pos = ((size_t)(blockIdx.x / n_regions) * start_stride) % n;
    uint32_t acc = 0;
    for (int p = 0; p < passes; p++) {
        for (size_t i = threadIdx.x; i < n; i += 256 * 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                size_t j = pos + i + (size_t)u * 256;
                if (j >= n) j -= n;
                if (j >= n) j -= n;
                if (VEC == 1) acc ^= reg[j];
                else {
                    uint4 q = reinterpret_cast<const uint4 *>(reg)[j];
                    acc ^= q.x ^ q.y ^ q.z ^ q.w;
                }
            }
        }
    }
    if (acc == 0x12345679u) out[0] = acc;     // never true in practice; keeps the loads
}

static double time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount, simds = cus * 4;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t *out; long long *clk;
    CK(hipMalloc(&out, sizeof(uint32_t) * 256 * (size_t)cus * 8));
    CK(hipMalloc(&clk, 16));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz_prop\": %.0f,\n", pr.name, cus, pr.clockRate / 1000.0);

    // ---- VALU issue ----
    const char *names[] = {"v_bitop3_b32", "v_bcnt_u32_b32", "v_and_or_b32", "v_add_u32", "v_alignbit_b32", "v_xor_b32"};
    printf(" \"valu\": [\n");
    bool first = true;
    for (int op = 0; op < 6; op++) {
        for (int wps : {1, 2, 4, 8}) {             // waves per SIMD: a 256-thread block puts one wave on each SIMD
            const int blocks = cus * wps, iters = 20000;
            auto launch = [&](int it) {
                switch (op) {
                case 0: hipLaunchKernelGGL(valu_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, it, 0x55aa55aau, 0x0f0f1234u, clk); break;
                case 1: hipLaunchKernelGGL(valu_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, it, 0x55aa55aau, 0x0f0f1234u, clk); break;
                case 2: hipLaunchKernelGGL(valu_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, it, 0x55aa55aau, 0x0f0f1234u, clk); break;
                case 3: hipLaunchKernelGGL(valu_kernel<3>, dim3(blocks), dim3(256), 0, 0, out, it, 0x55aa55aau, 0x0f0f1234u, clk); break;
                case 4: hipLaunchKernelGGL(valu_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, it, 0x55aa55aau, 7u, clk); break;
                default: hipLaunchKernelGGL(valu_kernel<5>, dim3(blocks), dim3(256), 0, 0, out, it, 0x55aa55aau, 0x0f0f1234u, clk); break;
                }
            };
            launch(2000);
            CK(hipDeviceSynchronize());
            double best = 1e30;
            long long hc[2] = {0, 0};
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                launch(iters);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                double ms = time_ms(e0, e1);
                if (ms < best) { best = ms; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost)); }
            }
            const double winstr = (double)blocks * 4 * iters * kUnroll;            // wave-instructions
            // s_memtime ticks of block 0 / its wall time (s_memrealtime, 100 MHz) = the counter's rate; if that is the
            // shader clock the per-SIMD rate follows directly, else use the property clock
            const double wall_s = hc[1] / 100e6, tick_mhz = hc[0] / wall_s / 1e6;
            const double per_simd_per_s = winstr / simds / (best * 1e-3);
            printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_instr_per_s_per_simd\": %.4e, "
                   "\"memtime_mhz\": %.1f, \"cycles_per_wave_instr_at_prop_clock\": %.3f, \"cycles_per_wave_instr_block0\": %.3f}",
                   first ? "" : ",\n", names[op], wps, best, per_simd_per_s, tick_mhz,
                   pr.clockRate * 1e3 / per_simd_per_s, (double)hc[0] / ((double)iters * kUnroll * wps));
            first = false;
        }
    }
    printf("\n ],\n");

    // ---- memory hierarchy reads ----
    struct Case { const char *name; size_t region_bytes; int n_regions; int passes; int blocks_per_cu; int vec; };
    const Case cases[] = {
        {"l2_dword", 2u << 20, 8, 64, 8, 1},        {"l2_dwordx4", 2u << 20, 8, 64, 8, 4},
        {"l2_1MiB_dword", 1u << 20, 8, 128, 8, 1},  {"mall_dword", 128u << 20, 1, 1, 8, 1},
        {"mall_dwordx4", 128u << 20, 1, 1, 8, 4},   {"mall_64MiB_dwordx4", 64u << 20, 1, 2, 8, 4},
    };
    uint32_t *buf;
    const size_t big = (size_t)4 << 30;
    CK(hipMalloc(&buf, big));
    CK(hipMemset(buf, 1, big));
    printf(" \"reads\": [\n");
    first = true;
    for (const Case &cs : cases) {
        const int blocks = cus * cs.blocks_per_cu;
        const size_t rw = cs.region_bytes / 4;
        const size_t stride = (rw / cs.vec) / (size_t)(blocks / cs.n_regions) + 256;
        auto launch = [&]() {
            if (cs.vec == 1) hipLaunchKernelGGL(read_kernel<1>, dim3(blocks), dim3(256), 0, 0, buf, rw, cs.n_regions, cs.passes, stride, out);
            else hipLaunchKernelGGL(read_kernel<4>, dim3(blocks), dim3(256), 0, 0, buf, rw, cs.n_regions, cs.passes, stride, out);
        };
        launch(); launch();
        CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            best = std::min(best, time_ms(e0, e1));
        }
        const double bytes = (double)blocks * cs.passes * (double)cs.region_bytes;
        printf("%s  {\"case\": \"%s\", \"region_bytes\": %zu, \"regions\": %d, \"bytes_per_launch\": %.0f, \"ms\": %.4f, \"GBs\": %.1f}",
               first ? "" : ",\n", cs.name, cs.region_bytes, cs.n_regions, bytes, best, bytes / (best * 1e-3) / 1e9);
        first = false;
    }
    {   // HBM: 4 GiB read once, every workgroup its own contiguous slice
        const int blocks = cus * 16;
        const size_t rw = big / 4 / blocks;
        auto launch = [&]() { hipLaunchKernelGGL(read_kernel<4>, dim3(blocks), dim3(256), 0, 0, buf, rw, blocks, 1, 0, out); };
        launch();
        CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            best = std::min(best, time_ms(e0, e1));
        }
        printf(",\n  {\"case\": \"hbm_dwordx4\", \"region_bytes\": %zu, \"regions\": %d, \"bytes_per_launch\": %.0f, \"ms\": %.4f, \"GBs\": %.1f}",
               rw * 4, blocks, (double)blocks * rw * 4, best, (double)blocks * rw * 4 / (best * 1e-3) / 1e9);
    }
    printf("\n ]\n}\n");
    return 0;
}
