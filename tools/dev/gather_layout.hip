// Dev probe: does the row stride of the gathered array matter?  One wave per point, lanes = channels, k = 10 neighbour
// rows of 64 floats summed per point (the access pattern of ec_stats_kernel / ec_apply_kernel), neighbours random
// within the point's cloud of 1024.  Layouts: rows of 128 floats of which the second 64 are read (the [P' | Q]
// rows of the edge convolution), rows of 64 floats (Q stored on its own), rows of 320 floats (a slot of the concat buffer).
//   hipcc --offload-arch=gfx950 -O2 tools/dev/gather_layout.hip -o tools/dev/gather_layout && tools/dev/gather_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void gather(int P, int N, int k, const int *idx, const float *q, int ld, int off, float *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int pt = blockIdx.x * 4 + wave; pt < P; pt += gridDim.x * 4) {
        const int base = pt / N * N;
        const int mine = lane < k ? idx[(size_t)pt * k + lane] : 0;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 10; ++j)
            acc += q[(size_t)(base + __shfl(mine, j, 64)) * ld + off + lane];
        out[(size_t)pt * 64 + lane] = acc;
    }
}
int main()
{
    const int B = 128, N = 1024, k = 10, P = B * N;
    std::vector<int> h((size_t)P * k);
    srand(1);
    for (auto &v : h)
        v = rand() % N;
    int *idx;
    float *q, *out;
    hipMalloc(&idx, h.size() * 4);
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&q, (size_t)P * 320 * 4);
    hipMemset(q, 0, (size_t)P * 320 * 4);
    hipMalloc(&out, (size_t)P * 64 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct { const char *name; int ld, off; } cases[] = {{"rows of 128, second half", 128, 64}, {"rows of 128, first half", 128, 0},
                                                       {"rows of 64 (dense)", 64, 0}, {"rows of 320, slot 1", 320, 64}};
    for (int b : {32, 128})
        for (auto &c : cases) {
            const int p = b * N;
            for (int it = 0; it < 3; ++it)
                hipLaunchKernelGGL(gather, dim3(4096), dim3(256), 0, 0, p, N, k, idx, q, c.ld, c.off, out);
            hipEventRecord(e0);
            for (int it = 0; it < 20; ++it)
                hipLaunchKernelGGL(gather, dim3(4096), dim3(256), 0, 0, p, N, k, idx, q, c.ld, c.off, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("B=%3d %-28s %7.1f us  %.2f TB/s gathered\n", b, c.name, ms * 50, (double)p * k * 256 / (ms / 20 * 1e-3) / 1e12);
        }
    return 0;
}
