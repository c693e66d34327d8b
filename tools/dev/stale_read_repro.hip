// Dev reproducer, outside the package (VERDICT r4 #7b): does a kernel that reads what the PREVIOUS kernel of the same stream wrote
// ever see stale lines when two processes dispatch small kernels at a high rate on ONE GPU?
//   writer:  buf[i] = f(step, i), rows staged in LDS and stored as 16-byte pieces (the shape of input_assemble_kernel's stores);
//   reader:  every workgroup brings its rows to LDS (plain 16-byte loads, or LDS DMA: global_load_lds_dwordx4) and sums them; the sum
//            of every workgroup is compared with the value the host computes for this step;  issued twice back to back ("twin").
// A step = FILL small kernels on other buffers (the rest of a train step: ~60 dispatches), then writer, reader, reader.  The buffer
// is reused every step with new contents and, every other step, another kernel overwrites it with garbage first.
//   stale_read_repro PROCS STEPS [DMA 0|1] [SHIFT]      (forks PROCS children before the first HIP call; AMD_OPT_FLUSH=0 to compare)
// (hipcc --offload-arch=gfx950 -O3 tools/dev/stale_read_repro.hip -o tools/dev/stale_read_repro)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>
typedef float float4v __attribute__((ext_vector_type(4)));
constexpr int ROWS = 4096, COLS = 24, WG_ROWS = 64;      // [B N, 24] fp32 rows: 96 bytes each, 64 rows per workgroup

__device__ __forceinline__ float value(unsigned step, unsigned i) { return (float)((step * 2654435761u + i * 40503u) >> 12 & 0xffff) * 0.25f; }

__global__ __launch_bounds__(256) void writer(float *buf, unsigned step)
{
    __shared__ float4v stage[WG_ROWS * COLS / 4];
    float *st = reinterpret_cast<float *>(stage);
    const unsigned base = blockIdx.x * WG_ROWS * COLS;
    for (int e = threadIdx.x; e < WG_ROWS * COLS; e += 256)
        st[e] = value(step, base + e);
    __syncthreads();
    for (int q = threadIdx.x; q < WG_ROWS * COLS / 4; q += 256)
        reinterpret_cast<float4v *>(buf + base)[q] = stage[q];
}
__global__ __launch_bounds__(256) void garbage(float *buf, unsigned step)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < ROWS * COLS)
        buf[i] = -1.0f - (float)(step & 7);
}
template <bool DMA>
__global__ __launch_bounds__(256) void reader(const float *buf, double *sums, int shift)
{
    __shared__ float4v stage[WG_ROWS * COLS / 4 + 256];
    __shared__ double part[4];
    // shift != 0: workgroup w reads the rows workgroup (w + shift) % gridDim.x wrote -- another XCD's workgroup when
    // workgroups go round the XCDs in order
    const unsigned src_wg = (blockIdx.x + shift) % gridDim.x;
    const unsigned base = src_wg * WG_ROWS * COLS;
    const float4v *src = reinterpret_cast<const float4v *>(buf + base);
    for (int q0 = 0; q0 < WG_ROWS * COLS / 4; q0 += 256) {
        const int q = q0 + threadIdx.x;
        const bool in = q < WG_ROWS * COLS / 4;
        if (DMA) {
            // wave-uniform LDS base + lane * 16: the DMA form of the same 16-byte load
            __builtin_amdgcn_global_load_lds(src + (in ? q : 0), stage + q0 + (threadIdx.x & ~63), 16, 0, 0);
        } else if (in) {
            stage[q] = src[q];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double s = 0.0;
    const float *st = reinterpret_cast<const float *>(stage);
    for (int e = threadIdx.x; e < WG_ROWS * COLS; e += 256)
        s += (double)st[e];
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        sums[src_wg] = part[0] + part[1] + part[2] + part[3];
}
__global__ void filler(float *p, int n, float a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        p[i] = p[i] * a + 1.0f;
}

static int child(int rank, int steps, bool dma, int shift)
{
    constexpr int WGS = ROWS / WG_ROWS, FILL = 56;
    float *buf, *other;
    double *s1, *s2;
    if (hipMalloc(&buf, sizeof(float) * ROWS * COLS) != hipSuccess) return 2;
    hipMalloc(&other, sizeof(float) * (1 << 20));
    hipMemset(other, 0, sizeof(float) * (1 << 20));
    hipMalloc(&s1, sizeof(double) * WGS);
    hipMalloc(&s2, sizeof(double) * WGS);
    std::vector<double> h1(WGS), h2(WGS), want(WGS);
    long bad_first = 0, bad_twin = 0;
    for (int step = 0; step < steps; ++step) {
        for (int f = 0; f < FILL; ++f)
            hipLaunchKernelGGL(filler, dim3(64 + 16 * (f % 7)), dim3(256), 0, 0, other + 4096 * (f % 16), 16384, 0.999f);
        if (step & 1)
            hipLaunchKernelGGL(garbage, dim3(ROWS * COLS / 256), dim3(256), 0, 0, buf, (unsigned)step);
        hipLaunchKernelGGL(writer, dim3(WGS), dim3(256), 0, 0, buf, (unsigned)step);
        if (dma) {
            hipLaunchKernelGGL(reader<true>, dim3(WGS), dim3(256), 0, 0, buf, s1, shift);
            hipLaunchKernelGGL(reader<true>, dim3(WGS), dim3(256), 0, 0, buf, s2, shift);
        } else {
            hipLaunchKernelGGL(reader<false>, dim3(WGS), dim3(256), 0, 0, buf, s1, shift);
            hipLaunchKernelGGL(reader<false>, dim3(WGS), dim3(256), 0, 0, buf, s2, shift);
        }
        hipMemcpy(h1.data(), s1, sizeof(double) * WGS, hipMemcpyDeviceToHost);     // (synchronises: the stream goes idle, as between steps)
        hipMemcpy(h2.data(), s2, sizeof(double) * WGS, hipMemcpyDeviceToHost);
        for (int w = 0; w < WGS; ++w) {
            double s = 0.0;
            for (int e = 0; e < WG_ROWS * COLS; ++e)
                s += (double)((float)(((unsigned)step * 2654435761u + (unsigned)(w * WG_ROWS * COLS + e) * 40503u) >> 12 & 0xffff) * 0.25f);
            want[w] = s;
        }
        bool b1 = false, b2 = false;
        for (int w = 0; w < WGS; ++w) {
            b1 |= h1[w] != want[w];
            b2 |= h2[w] != want[w];
        }
        bad_first += b1;
        bad_twin += b2;
    }
    printf("process %d (%s loads, reader shifted by %d workgroups): %d steps, first reader wrong in %ld, its twin in %ld\n", rank,
           dma ? "LDS-DMA" : "plain", shift, steps, bad_first, bad_twin);
    return 0;
}

int main(int argc, char **argv)
{
    const int procs = argc > 1 ? atoi(argv[1]) : 2, steps = argc > 2 ? atoi(argv[2]) : 3000;
    const bool dma = argc > 3 && atoi(argv[3]) != 0;
    const int shift = argc > 4 ? atoi(argv[4]) : 0;
    std::vector<pid_t> kids;
    for (int r = 0; r < procs; ++r) {
        pid_t p = fork();       // (before anything touches the GPU)
        if (p == 0)
            return child(r, steps, dma, shift);
        kids.push_back(p);
    }
    int rc = 0;
    for (pid_t p : kids) {
        int st = 0;
        waitpid(p, &st, 0);
        rc |= st;
    }
    return rc != 0;
}
