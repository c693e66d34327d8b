// Dev micro-benchmark: do vector instructions of the SAME SIMD run while v_mfma_f32_32x32x2_f32 executes?
// 1024-thread workgroups (four waves per SIMD, one workgroup per CU), per wave and iteration: 8 dependent MFMAs (one chain)
// and V independent v_fma_f32 (chains in other registers); accumulators in VGPRs (builtin) or in AGPRs (inline asm).
// (hipcc --offload-arch=gfx950 -O3 tools/dev/mfma_valu_overlap.hip -o tools/dev/mfma_valu_overlap)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int V>      // MODE 0: MFMA (VGPR acc) + VALU, 1: MFMA (AGPR acc) + VALU, 2: VALU only, 3: MFMA only (VGPR)
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc;
    for (int r = 0; r < 16; ++r)
        acc[r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    float v[8];
    for (int i = 0; i < 8; ++i)
        v[i] = a + i;
    if (MODE == 1)
        asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, %1, 0" ::"v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7",
                     "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0 || MODE == 3)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            if (MODE == 1)
                asm volatile("v_mfma_f32_32x32x2_f32 a[0:15], %0, %1, a[0:15]" ::"v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5",
                             "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
            if (MODE != 3) {
#pragma unroll
                for (int j = 0; j < V; ++j)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b), "v"(a));
            }
        }
    }
    float s = 0.f;
    if (MODE == 1) {
        asm volatile("s_nop 15\n s_nop 3\n v_accvgpr_read_b32 %0, a0" : "=v"(s));
    } else {
        for (int r = 0; r < 16; ++r)
            s += acc[r];
    }
    for (int i = 0; i < 8; ++i)
        s += v[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int MODE, int V>
void run(const char *tag)
{
    const int blocks = 256, iters = 20000;
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, V>), dim3(blocks), dim3(1024), 0, 0, out, 2000, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, V>), dim3(blocks), dim3(1024), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD and iteration: 4 waves x 8 MFMAs x 64 cycles = 2048 cycles of matrix work, 4 waves x 8 V x 4 cycles of vector work
    printf("%-28s V=%2d: %.3f ms = %.0f ns per iteration (matrix alone 2048 cycles, vector alone %d cycles)\n", tag, V, ms,
           ms * 1e6 / iters, 4 * 8 * V * 4);
    hipFree(out);
}
int main()
{
    run<3, 0>("MFMA only (VGPR acc)");
    run<2, 4>("VALU only");
    run<2, 8>("VALU only");
    run<2, 12>("VALU only");
    run<0, 4>("MFMA (VGPR acc) + VALU");
    run<0, 8>("MFMA (VGPR acc) + VALU");
    run<0, 12>("MFMA (VGPR acc) + VALU");
    run<1, 0>("MFMA (AGPR acc)");
    run<1, 4>("MFMA (AGPR acc) + VALU");
    run<1, 8>("MFMA (AGPR acc) + VALU");
    run<1, 12>("MFMA (AGPR acc) + VALU");
    return 0;
}
