// pk_opsel_repro.hip -- stand-alone reproducer of the packed-fp32 fault behind profiles/notes_two_processes_one_gpu.md (round 6).
//   hipcc --offload-arch=gfx950 -O2 tools/dev/pk_opsel_repro.hip -o tools/dev/pk_opsel_repro
//   tools/dev/pk_opsel_repro 20000            # alone: expect 0 wrong
//   (python bench.py --step-only --steps 4000 &) ; tools/dev/pk_opsel_repro 20000    # next to a process running training steps
//   tools/dev/pk_opsel_repro 20000 K      # K = 1..4: next to a load kernel of THIS process on a second stream
//                                         # (1 fp32 fma loop, 2 v_mfma_f32_32x32x2_f32 loop, 3 packed-fp32 loop, 4 memory copy)
// Every lane computes w = x*x + y*y + z*z twice: with the instruction sequence hipcc 7.2 emits for those three lines behind a
// global_load_dwordx3 (two packed instructions, the last one `v_pk_add_f32 ... op_sel:[0,1]`: its low half takes z*z from the
// HIGH register of a pair) and with one-value-per-lane instructions.  A launch = 64 workgroups of 256 threads, like the layer-1
// kNN of a [16, 256] batch.  Prints how many lanes disagreed, which lanes, and what the wrong values were.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void probe(const float *__restrict__ pts, int ld, unsigned *__restrict__ wrong,
                                             unsigned *__restrict__ by_lane, float *__restrict__ sample, int variant)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float *row = pts + (size_t)i * ld;
    const float x = row[0], y = row[1], z = row[2];
    float w;
    if (variant == 0)
        asm volatile("v_mov_b32 v60, %1\n\t"
                     "v_mov_b32 v61, %3\n\t"
                     "v_mul_f32 v62, %2, %2\n\t"
                     "v_pk_mul_f32 v[60:61], v[60:61], v[60:61]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_add_f32 v[62:63], v[60:61], v[62:63] op_sel_hi:[1,0]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_add_f32 v[64:65], v[62:63], v[60:61] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                     "s_nop 0\n\t"
                     "v_mov_b32 %0, v64"
                     : "=&v"(w) : "v"(x), "v"(y), "v"(z) : "v60", "v61", "v62", "v63", "v64", "v65");
    else                                  // the same registers, the last addition as a plain v_add_f32
        asm volatile("v_mov_b32 v60, %1\n\t"
                     "v_mov_b32 v61, %3\n\t"
                     "v_mul_f32 v62, %2, %2\n\t"
                     "v_pk_mul_f32 v[60:61], v[60:61], v[60:61]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_add_f32 v[62:63], v[60:61], v[62:63] op_sel_hi:[1,0]\n\t"
                     "s_nop 0\n\t"
                     "v_add_f32 v64, v62, v61\n\t"
                     "s_nop 0\n\t"
                     "v_mov_b32 %0, v64"
                     : "=&v"(w) : "v"(x), "v"(y), "v"(z) : "v60", "v61", "v62", "v63", "v64", "v65");
    float a = x * x, b = y * y, c = z * z;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
    float e = a + b;
    asm volatile("" : "+v"(e));
    e = e + c;
    if (__float_as_uint(e) != __float_as_uint(w)) {
        const unsigned at = atomicAdd(wrong, 1u);
        atomicAdd(&by_lane[threadIdx.x & 63], 1u);
        if (at < 8) {
            sample[4 * at] = w;
            sample[4 * at + 1] = e;
            sample[4 * at + 2] = a + b;
            sample[4 * at + 3] = (float)(threadIdx.x & 63);
        }
    }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// something else for the SIMDs to do, from a second stream of the same process
__global__ __launch_bounds__(256) void load_kernel(float *__restrict__ buf, int iters, int kind, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float a = buf[i % n], b = 1.0001f;
    if (kind == 1) {
        for (int t = 0; t < iters; ++t)
            a = __builtin_fmaf(a, b, 0.5f);
    } else if (kind == 2) {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < iters / 16; ++t)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        a = acc[0] + acc[7];
    } else if (kind == 3) {
        f32x2 v = {a, b}, w = {b, a};
        for (int t = 0; t < iters; ++t)
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(w));
        a = v.x + v.y;
    } else {
        for (int t = 0; t < iters / 64; ++t)
            a += buf[(i + (size_t)t * 1048576) % n];
    }
    buf[i % n] = a;
}

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 20000;
    const int self_load = argc > 2 ? atoi(argv[2]) : 0;
    hipStream_t side;
    hipStreamCreate(&side);
    const size_t nbuf = 64u << 20;
    float *buf = nullptr;
    if (self_load) {
        hipMalloc(&buf, nbuf * 4);
        hipMemset(buf, 0, nbuf * 4);
        printf("load kernel %d of this process on a second stream\n", self_load);
    }
    const int n = 64 * 256, ld = 24;
    std::vector<float> h((size_t)n * ld);
    srand(1);
    for (auto &v : h)
        v = (float)rand() / RAND_MAX * 0.2f - 0.1f;
    float *pts, *sample;
    unsigned *wrong, *by_lane;
    hipMalloc(&pts, h.size() * 4);
    hipMalloc(&sample, 32 * 4);
    hipMalloc(&wrong, 4);
    hipMalloc(&by_lane, 64 * 4);
    hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; ++variant) {
        hipMemset(wrong, 0, 4);
        hipMemset(by_lane, 0, 64 * 4);
        hipMemset(sample, 0, 32 * 4);
        for (int l = 0; l < launches; ++l) {
            if (self_load && l % 8 == 0)
                hipLaunchKernelGGL(load_kernel, dim3(4096), dim3(256), 0, side, buf, 4096, self_load, nbuf);
            hipLaunchKernelGGL(probe, dim3(64), dim3(256), 0, 0, pts, ld, wrong, by_lane, sample, variant);
        }
        hipDeviceSynchronize();
        unsigned w = 0, lanes[64];
        float s[32];
        hipMemcpy(&w, wrong, 4, hipMemcpyDeviceToHost);
        hipMemcpy(lanes, by_lane, 64 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(s, sample, 32 * 4, hipMemcpyDeviceToHost);
        printf("%s: %d launches x %d lanes, wrong lanes: %u", variant == 0 ? "v_pk_add_f32 op_sel:[0,1]" : "v_add_f32             ",
               launches, n, w);
        if (w) {
            printf("  by lane:");
            for (int q = 0; q < 64; ++q)
                if (lanes[q])
                    printf(" %d:%u", q, lanes[q]);
            printf("\n   first: got %.9g, right %.9g, x*x + y*y = %.9g (lane %d)", s[0], s[1], s[2], (int)s[3]);
        }
        printf("\n");
    }
    return 0;
}
