// Dev probe: how many workgroups of a given size and dynamic-LDS footprint run side by side on one CU.
// Each workgroup spins for a fixed number of clock ticks; grid = 256 CUs x `per_cu`; the elapsed time tells
// whether the `per_cu` workgroups of a CU ran concurrently (1 x spin) or one after the other (per_cu x spin).
//   hipcc --offload-arch=gfx950 -O2 tools/dev/lds_occupancy.hip -o /tmp/lds_occ && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long ticks, int *sink)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {
    }
    if (lds[(threadIdx.x + 1) % blockDim.x] == -1)
        *sink = 1;
}
int main()
{
    int *sink;
    hipMalloc(&sink, 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const long long ticks = 5000;   // 100 MHz wall clock: 50 us
    for (int threads : {256, 512, 1024})
        for (int kb : {16, 32, 48, 64, 72, 76, 78, 80, 96, 128, 160}) {
            for (int per_cu : {2}) {
                hipLaunchKernelGGL(spin, dim3(256 * per_cu), dim3(threads), kb * 1024, 0, ticks, sink);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(spin, dim3(256 * per_cu), dim3(threads), kb * 1024, 0, ticks, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                printf("threads %4d lds %3d KB x%d per CU: %.1f us (%s)\n", threads, kb, per_cu, ms * 1e3,
                       hipGetLastError() != hipSuccess ? "launch failed" : (ms * 1e3 > 80 ? "serial" : "side by side"));
            }
        }
    return 0;
}
