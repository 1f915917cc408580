// ubench_overlap.hip — do the register-return path of a CU (buffer_load_dwordx4, L2 hits: 64 B/clk) and the vector ALUs overlap, or do
// their times add?  Every wave keeps D x 16-byte loads per lane in flight (the evaluation kernel's fetch shape: `buffer_load_dwordx4
// v, v_off, s[rsrc], s_off offen`) and spends NV vector instructions on each batch of D loads while the next batch is under way.
// Reported per NV: time, bytes/clk/CU, VALU wave-instructions/s/SIMD, and the two single-resource times the run is made of.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench_overlap tools/ubench_overlap.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int D = 4;

template <int NV>
__device__ __forceinline__ void valu(uint32_t (&acc)[4], const u32x4 (&x)[D]) {
#pragma unroll
    for (int i = 0; i < NV / 16; i++)          // 16 instructions per round: every loaded register used once
#pragma unroll
        for (int u = 0; u < D; u++)
            asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %6\n v_xor_b32 %3, %3, %7\n"
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(x[u].x), "v"(x[u].y), "v"(x[u].z), "v"(x[u].w));
}

// LOADS = 0: the vector instructions alone (operands from the first batch)
template <int NV, int LOADS>
__global__ __launch_bounds__(256) void overlap_kernel(const uint32_t *__restrict__ src, uint32_t *out, int iters, uint32_t row_bytes, uint32_t rows_mask) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(src), 0, 0x7FFFFFFF, 0x00020000);
    const int voff = (int)(threadIdx.x & 63) * 16 + (int)(threadIdx.x >> 6) * 1024;
    uint32_t row = blockIdx.x * 7u;
    uint32_t acc[4] = {0, 0, 0, 0};
    u32x4 cur[D], nxt[D];
#pragma unroll
    for (int u = 0; u < D; u++) cur[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (int)(((row + u) & rows_mask) * row_bytes), 0);
    for (int it = 0; it < iters; it++) {
        row += D;
        if (LOADS) {
#pragma unroll
            for (int u = 0; u < D; u++) nxt[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (int)(((row + u) & rows_mask) * row_bytes), 0);
        }
        valu<NV>(acc, cur);
        if (LOADS) {
#pragma unroll
            for (int u = 0; u < D; u++) cur[u] = nxt[u];
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

template <int NV, int LOADS>
static double run(const uint32_t *src, uint32_t *out, int blocks, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((overlap_kernel<NV, LOADS>), dim3(blocks), dim3(256), 0, 0, src, out, iters, 4096u, 255u);   // 256 rows x 4 KB = 1 MB: L2 hits
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    return ms;
}

template <int NV>
static void point(const uint32_t *src, uint32_t *out, int wps, bool &first) {
    const int blocks = 256 * wps, iters = 4000;
    // "loads only" = 16 vector instructions per batch: one per loaded register, or the compiler drops the loads
    const double both = run<NV, 1>(src, out, blocks, iters), mem = run<16, 1>(src, out, blocks, iters), alu = run<NV, 0>(src, out, blocks, iters);
    const double bytes = (double)blocks * 256 * iters * D * 16, instr = (double)blocks * 4 * iters * NV;
    printf("%s{\"valu_per_4_loads\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"ms_loads_only\": %.3f, \"ms_valu_only\": %.3f, \"sum_over_measured\": %.2f, "
           "\"max_over_measured\": %.2f, \"TBps\": %.2f, \"valu_wave_instr_per_s_per_simd\": %.3g}",
           first ? "" : ",\n ", NV, wps, both, mem, alu, (mem + alu) / both, (mem > alu ? mem : alu) / both, bytes / (both * 1e-3) / 1e12, instr / (both * 1e-3) / 1024);
    first = false;
}

int main() {
    uint32_t *src, *out;
    CK(hipMalloc(&src, 2 << 20));
    CK(hipMemset(src, 1, 2 << 20));
    CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    printf("{\"overlap\": [");
    bool first = true;
    for (int wps = 4; wps <= 8; wps *= 2) {
        point<16>(src, out, wps, first);
        point<32>(src, out, wps, first);
        point<64>(src, out, wps, first);
        point<96>(src, out, wps, first);
        point<128>(src, out, wps, first);
        point<192>(src, out, wps, first);
        point<256>(src, out, wps, first);
    }
    printf("]}\n");
    return 0;
}
