// Dev micro-benchmark: do vector instructions run while the matrix pipe of the SAME SIMD executes an MFMA?
// 1024-thread workgroups (sixteen waves: four per SIMD, one workgroup per CU).  Two matrix instructions:
//   F32  v_mfma_f32_32x32x2_f32   (64 cycles: the FP32 vector rate),   BF16  v_mfma_f32_32x32x16_bf16 (32 cycles).
// Three arrangements of the same work per SIMD and iteration (32 MFMAs, 32 V independent v_fma_f32):
//   SAME   every wave issues 8 MFMAs (one dependent chain) and 8 V vector instructions, interleaved;
//   SPLIT  role-split waves: two waves of every SIMD issue 16 MFMAs each and nothing else, the other two 16 V vector
//          instructions each and nothing else (wave w runs on SIMD w % 4: roles by (w / 4) % 2);
//   and the two halves alone (MFMA only, VALU only).
// (hipcc --offload-arch=gfx950 -O3 tools/dev/mfma_valu_overlap.hip -o tools/dev/mfma_valu_overlap)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { SAME = 0, SPLIT = 1, MFMA_ONLY = 2, VALU_ONLY = 3 };

template <int BF16>
__device__ __forceinline__ f32x16 mma(f32x16 acc, float a, float b, bf16x8 ab, bf16x8 bb)
{
    if (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}

template <int BF16, int MODE, int V>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc;
    for (int r = 0; r < 16; ++r)
        acc[r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) {
        ab[i] = (__bf16)(a0 + i);
        bb[i] = (__bf16)b0;
    }
    float v[8];
    for (int i = 0; i < 8; ++i)
        v[i] = a + i;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool matrix_wave = ((wave >> 2) & 1) == 0;
    if (MODE == SAME || MODE == MFMA_ONLY || MODE == VALU_ONLY) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE != VALU_ONLY)
                    acc = mma<BF16>(acc, a, b, ab, bb);
                if (MODE != MFMA_ONLY) {
#pragma unroll
                    for (int j = 0; j < V; ++j)
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b), "v"(a));
                }
            }
        }
    } else if (matrix_wave) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                acc = mma<BF16>(acc, a, b, ab, bb);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
#pragma unroll
                for (int j = 0; j < V; ++j)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(b), "v"(a));
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r)
        s += acc[r];
    for (int i = 0; i < 8; ++i)
        s += v[i];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int BF16, int MODE, int V>
void run(const char *tag)
{
    const int blocks = 256, iters = 20000;
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BF16, MODE, V>), dim3(blocks), dim3(1024), 0, 0, out, 2000, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BF16, MODE, V>), dim3(blocks), dim3(1024), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD and iteration: 32 MFMAs x (64 | 32) cycles of matrix work, 32 V x 4 cycles of vector work
    printf("%-5s %-10s V=%2d: %7.3f ms = %5.0f ns per iteration (matrix alone %d cycles, vector alone %d cycles)\n",
           BF16 ? "BF16" : "F32", tag, V, ms, ms * 1e6 / iters, MODE == VALU_ONLY ? 0 : 32 * (BF16 ? 32 : 64),
           MODE == MFMA_ONLY ? 0 : 32 * V * 4);
    hipFree(out);
}

template <int BF16>
void all()
{
    run<BF16, MFMA_ONLY, 0>("MFMA only");
    run<BF16, VALU_ONLY, 4>("VALU only");
    run<BF16, VALU_ONLY, 8>("VALU only");
    run<BF16, VALU_ONLY, 12>("VALU only");
    run<BF16, SAME, 4>("same waves");
    run<BF16, SAME, 8>("same waves");
    run<BF16, SAME, 12>("same waves");
    run<BF16, SPLIT, 4>("role split");
    run<BF16, SPLIT, 8>("role split");
    run<BF16, SPLIT, 12>("role split");
}

int main()
{
    all<0>();
    all<1>();
    return 0;
}
