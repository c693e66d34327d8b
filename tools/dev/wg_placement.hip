// Dev probe: where the dispatcher puts the workgroups of a small grid.  Every workgroup records (XCC, SE, CU) of its
// first wave and spins for a while (so that none retires before the last one starts); the host counts distinct CUs.
//   hipcc --offload-arch=gfx950 -O2 tools/dev/wg_placement.hip -o tools/dev/wg_place && tools/dev/wg_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void where(long long ticks, unsigned *out)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = 1;
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    while (wall_clock64() - t0 < ticks) {
    }
}
int main()
{
    unsigned *out;
    hipMalloc(&out, 8 * 4096);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&where), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned> h(2 * 4096);
    for (int threads : {256, 512, 1024})
        for (int kb : {8, 40, 78, 150})
            for (int grid : {128, 256, 512}) {
                hipLaunchKernelGGL(where, dim3(grid), dim3(threads), kb * 1024, 0, 3000LL, out);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), out, 8 * grid, hipMemcpyDeviceToHost);
                std::map<unsigned, int> per_cu;
                for (int b = 0; b < grid; ++b) {
                    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
                    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                    per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
                }
                int most = 0;
                for (auto &kv : per_cu)
                    most = kv.second > most ? kv.second : most;
                printf("threads %4d lds %3d KB grid %3d: %3zu distinct CUs, at most %d workgroups on one\n", threads, kb, grid,
                       per_cu.size(), most);
            }
    return 0;
}
