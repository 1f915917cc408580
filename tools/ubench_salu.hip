// ubench_salu.hip — scalar-unit throughput of a CU (gfx950): the bit-sliced kernels spend 0.6-1.2 scalar instructions per vector
// instruction on address and symbol decoding, and a CU has ONE scalar unit for its four SIMDs.  Measures SALU wave-instructions per
// second per CU for s_add_u32 / s_and_b32 / s_lshl_b32 chains (8 independent chains per wave) at 1..8 waves per SIMD, alone and
// interleaved 1:1 with v_add_u32.   hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench_salu tools/ubench_salu.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MIX>
__global__ __launch_bounds__(256) void salu_kernel(uint32_t *out, int iters, uint32_t a) {
    uint32_t s0 = a, s1 = a + 1, s2 = a + 2, s3 = a + 3, s4 = a + 4, s5 = a + 5, s6 = a + 6, s7 = a + 7;
    uint32_t v = threadIdx.x;
    for (int it = 0; it < iters; it++) {
        asm volatile("s_add_u32 %0, %0, %8\n s_add_u32 %1, %1, %8\n s_add_u32 %2, %2, %8\n s_add_u32 %3, %3, %8\n"
                     "s_add_u32 %4, %4, %8\n s_add_u32 %5, %5, %8\n s_add_u32 %6, %6, %8\n s_add_u32 %7, %7, %8\n"
                     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(a) : "scc");
        if (MIX) {
            asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                         "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n" : "+v"(v) : "v"(a));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7 ^ v;
}

int main() {
    uint32_t *out;
    CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    const int iters = 20000;
    printf("{\"salu\": [");
    bool first = true;
    for (int mix = 0; mix < 2; mix++)
        for (int wps = 1; wps <= 8; wps *= 2) {
            const int blocks = 256 * wps;                       // 4 waves per block: wps waves per SIMD when every CU holds wps blocks
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                if (mix) hipLaunchKernelGGL(salu_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
                else hipLaunchKernelGGL(salu_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
            }
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double salu = (double)blocks * 4 * iters * 8;
            printf("%s{\"mix_valu\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"salu_wave_instr_per_s_per_cu\": %.4g, \"valu_wave_instr_per_s_per_simd\": %.4g}",
                   first ? "" : ", ", mix, wps, ms, salu / (ms * 1e-3) / 256, mix ? salu / (ms * 1e-3) / 1024 : 0.0);
            first = false;
        }
    printf("]}\n");
    return 0;
}
