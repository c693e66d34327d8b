// Dev micro-benchmark: what does v_mfma_f32_32x32x2_f32 sustain with nothing else in the way?
// (hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void peak(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r)
            acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r)
            s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters, const char *tag)
{
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(peak<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(peak<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
    printf("%s blocks=%d (waves/SIMD=%d) NACC=%d: %.3f ms  %.1f TFLOP/s\n", tag, blocks, blocks / 256, NACC, ms,
           flops / ms / 1e9);
    hipFree(out);
}
int main()
{
    run<4>(256, 20000, "short");
    run<4>(256, 200000, "long ");
    run<4>(512, 100000, "long ");
    run<4>(768, 60000, "long ");
    run<1>(256, 100000, "dep1w");
    run<2>(256, 100000, "dep1w");
    run<1>(512, 100000, "dep2w");
    run<1>(1024, 100000, "dep  ");
    run<2>(1024, 100000, "dep2 ");
    return 0;
}
