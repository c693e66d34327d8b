// sampling.hip -- farthest point sampling, point gather and its gradient (gfx950).
//
// Replaces farthestpointsamplingLauncher / gatherpointLauncher /
// scatteraddpointLauncher (reference tf_ops/sampling/tf_sampling_g.cu:203-211).
//
// FPS is `m-1` dependent rounds; the cost is the per-round latency, so the
// design keeps everything a round touches on chip: one 512-thread workgroup per
// cloud, each thread owns points k = t, t+512, ... with their coordinates AND
// running minimum distance in registers (the reference round-trips the running
// minimum through a global `temp` array, tf_sampling_g.cu:139,144), the cloud is
// mirrored in LDS only to fetch the last pick's coordinates, and the arg-max is
// an order-independent u64 max per round over keys
//      bits(d2) << 32 | tie-break code of k
// d2 >= 0, so its bit pattern orders like the float.  The largest key is the
// reference's winner -- max value, then lowest k mod 512, then lowest k -- which
// its thread-strided scan + left-biased tree (:130-165) produces.  One barrier
// per round.  fps_kernel (clouds up to 16384 points) below; fps_big_kernel beyond.
#include "common.h"
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

constexpr int FPS_THREADS = 512;
constexpr int FPS_WAVES = FPS_THREADS / 64;

__device__ __forceinline__ float fps_sqdist(float x2, float y2, float z2, float x1, float y1, float z1)
{
    // tf_sampling_g.cu:142, un-fused, left to right
    const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
    return dx * dx + dy * dy + dz * dz;
}

// PPT = points per thread held in registers (n <= 512*PPT).  Round 6: a round is ONE chain of dependent steps, and it is as
// fast as that chain is short --
//   per lane   : the maximum of its PPT running minima (v_max3_f32); their bit patterns order like integers (minima are >= +0;
//                a slot past the cloud holds -1 and +inf coordinates: it stays -1 for ever)
//   per wave   : six v_max_i32 with DPP modifiers leave the wave's maximum in lane 63; every lane that holds it (one, unless
//                distances tie) finds its lowest slot holding it and sends ONE LDS atomic, ds_max_u64 of
//                    bits(maximum) << 32 | (511 - t) << 16 | 65535 - p            (k = t + 512 p)
//                to this round's cell: the largest key is the reference's winner -- max value, then lowest k mod 512, then
//                lowest k (its thread-strided scan + left-biased tree, tf_sampling_g.cu:130-165)
//   one barrier
//   every lane : reads the cell (a broadcast), decodes k, reads the winner's coordinates from the cloud's copy in LDS.
// Three cells in rotation: the one of round j + 1 is cleared while round j runs.  No 64-bit compares in registers, no
// table of per-wave candidates to reduce, two scalar-register hops per round.  (profiles/notes_fps_r6.md)
#define FPS_DPP_MAX(v, ctrl) asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl : "+v"(v))
__device__ __forceinline__ float fps_min(float a, float b)      // v_min_f32 as is (fminf first canonicalises b: one more instruction)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// LDS_XYZ: the cloud copied to dynamic LDS (12 n bytes) for the winner's coordinates; else they come from memory.
template <int PPT, bool LDS_XYZ>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(int n, int m, const float *__restrict__ inp, int *__restrict__ out)
{
    extern __shared__ float lds_xyz[];
    __shared__ unsigned long long cell[3];

    const int t = threadIdx.x;
    const int cloud = blockIdx.x;
    const float *P = inp + (size_t)cloud * n * 3;
    int *O = out + (size_t)cloud * m;

    float px[PPT], py[PPT], pz[PPT], run[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const int k = t + FPS_THREADS * p;
        const bool ok = k < n;
        px[p] = ok ? P[3 * k] : __builtin_inff();
        py[p] = ok ? P[3 * k + 1] : __builtin_inff();
        pz[p] = ok ? P[3 * k + 2] : __builtin_inff();
        run[p] = ok ? 1e38f : -1.0f;  // tf_sampling_g.cu:117
    }
    if (LDS_XYZ) {
        for (int f = t; f < n * 3; f += FPS_THREADS)
            lds_xyz[f] = P[f];
    }
    if (t < 3)
        cell[t] = 0;
    if (t == 0)
        O[0] = 0;
    float ox = P[0], oy = P[1], oz = P[2];      // idx_0 = 0 (:114)
    __syncthreads();

    int slot = 1;                               // j % 3
    for (int j = 1; j < m; ++j) {
        float lm = -1.0f;
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            const float d = fps_sqdist(px[p], py[p], pz[p], ox, oy, oz);
            run[p] = fps_min(d, run[p]);
            lm = fmaxf(lm, run[p]);
        }
        int v = __float_as_int(lm);
        FPS_DPP_MAX(v, "row_shr:1 row_mask:0xf bank_mask:0xf");
        FPS_DPP_MAX(v, "row_shr:2 row_mask:0xf bank_mask:0xf");
        FPS_DPP_MAX(v, "row_shr:4 row_mask:0xf bank_mask:0xf");
        FPS_DPP_MAX(v, "row_shr:8 row_mask:0xf bank_mask:0xf");        // lane 15 of a row: the row's maximum
        FPS_DPP_MAX(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");     // rows 1, 3 take in rows 0, 2
        FPS_DPP_MAX(v, "row_bcast:31 row_mask:0xc bank_mask:0xf");     // rows 2, 3 take in lane 31: lane 63 has the wave's
        const int umax = __builtin_amdgcn_readlane(v, 63);
        const int next = slot == 2 ? 0 : slot + 1;
        if (__float_as_int(lm) == umax && umax >= 0) {                 // (a wave wholly past the cloud holds -1: no key)
            int ps = PPT - 1;
#pragma unroll
            for (int p = PPT - 2; p >= 0; --p)
                ps = __float_as_int(run[p]) == umax ? p : ps;          // this thread's lowest k holding the maximum (:146)
            const unsigned long long key = ((unsigned long long)(unsigned)umax << 32) |
                                           (unsigned)(((FPS_THREADS - 1 - t) << 16) | (65535 - ps));
            // (the cell's index through a register the compiler cannot see through: with a uniform address it first combines the
            //  wave's lanes in a readlane loop -- longer than the one ds_max_u64 of the one lane that usually gets here)
            int opaque = slot;
            asm volatile("" : "+v"(opaque));
            __hip_atomic_fetch_max(&cell[opaque], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (t == 0)
            cell[next] = 0;
        __syncthreads();
        const unsigned code = (unsigned)cell[slot];
        const int last = (FPS_THREADS - 1 - (int)(code >> 16)) + FPS_THREADS * (65535 - (int)(code & 0xffffu));
        if (LDS_XYZ) {
            ox = lds_xyz[3 * last];
            oy = lds_xyz[3 * last + 1];
            oz = lds_xyz[3 * last + 2];
        } else {
            ox = P[3 * last];
            oy = P[3 * last + 1];
            oz = P[3 * last + 2];
        }
        if (t == 0)
            O[j] = last;
        slot = next;
    }
}

// n > 16384: running minimum in the caller's `temp` (32*n floats, one row per
// workgroup, as tf_sampling.cpp:115 sizes it); clouds strided over <= 32 groups.
__global__ __launch_bounds__(FPS_THREADS) void fps_big_kernel(int b, int n, int m,
                                                              const float *__restrict__ inp,
                                                              float *__restrict__ temp,
                                                              int *__restrict__ out)
{
    __shared__ unsigned long long slot[2][FPS_WAVES];
    const int t = threadIdx.x;
    float *run = temp + (size_t)blockIdx.x * n;
    for (int cloud = blockIdx.x; cloud < b; cloud += gridDim.x) {
        const float *P = inp + (size_t)cloud * n * 3;
        int *O = out + (size_t)cloud * m;
        for (int k = t; k < n; k += FPS_THREADS)
            run[k] = 1e38f;
        if (t == 0)
            O[0] = 0;
        __syncthreads();
        int last = 0;
        for (int j = 1; j < m; ++j) {
            const float ox = P[3 * last], oy = P[3 * last + 1], oz = P[3 * last + 2];
            unsigned bestv = 0, bestp = 0;
            bool any = false;
            unsigned p = 0;
            for (int k = t; k < n; k += FPS_THREADS, ++p) {
                const float d = fps_sqdist(P[3 * k], P[3 * k + 1], P[3 * k + 2], ox, oy, oz);
                const float d2 = fminf(d, run[k]);
                run[k] = d2;
                const unsigned u = __float_as_uint(d2);
                if (!any || u > bestv) {
                    bestv = u;
                    bestp = p;
                    any = true;
                }
            }
            unsigned long long key = 0;
            if (any)
                key = ((unsigned long long)bestv << 32) |
                      ((unsigned long long)(FPS_THREADS - 1 - t) << 23) | bestp;
            key = wave_max_u64(key);
            if ((t & 63) == 0)
                slot[j & 1][t >> 6] = key;
            __syncthreads();
            unsigned long long w = slot[j & 1][0];
            for (int i = 1; i < FPS_WAVES; ++i) {
                const unsigned long long o = slot[j & 1][i];
                w = o > w ? o : w;
            }
            const unsigned lo = (unsigned)w;
            last = (FPS_THREADS - 1 - (int)((lo >> 23) & 511u)) +
                   FPS_THREADS * (int)(lo & 0x7fffffu);
            if (t == 0)
                O[j] = last;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void gather_point_kernel(int n, int m,
                                                           const float *__restrict__ inp,
                                                           const int *__restrict__ idx,
                                                           float *__restrict__ out)
{
    const int cloud = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m)
        return;
    const int a = idx[(size_t)cloud * m + j];
    const float *src = inp + ((size_t)cloud * n + a) * 3;
    float *dst = out + ((size_t)cloud * m + j) * 3;
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
}

__global__ __launch_bounds__(256) void scatter_add_point_kernel(int n, int m,
                                                                const float *__restrict__ out_g,
                                                                const int *__restrict__ idx,
                                                                float *__restrict__ inp_g)
{
    const int cloud = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m)
        return;
    const int a = idx[(size_t)cloud * m + j];
    const float *src = out_g + ((size_t)cloud * m + j) * 3;
    float *dst = inp_g + ((size_t)cloud * n + a) * 3;
    atomicAdd(dst + 0, src[0]);
    atomicAdd(dst + 1, src[1]);
    atomicAdd(dst + 2, src[2]);
}

template <int PPT>
static int launch_fps_reg(int b, int n, int m, const float *inp, int *out, hipStream_t s)
{
    const size_t lds = (size_t)n * 3 * sizeof(float);
    if (lds <= 144 * 1024) {
        // opt in to > 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU)
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)fps_kernel<PPT, true>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                set_error("cloudaae_farthest_point_sample: %s", hipGetErrorString(e));
                return (int)e;
            }
        }
        hipLaunchKernelGGL((fps_kernel<PPT, true>), dim3(b), dim3(FPS_THREADS), lds, s, n, m,
                           inp, out);
    } else {
        hipLaunchKernelGGL((fps_kernel<PPT, false>), dim3(b), dim3(FPS_THREADS), 0, s, n, m,
                           inp, out);
    }
    return 0;
}

// ---- ProbSample (tf_sampling_g.cu:7-104) ------------------------------------------------
// Inclusive prefix sums of a row of category weights, then one search per uniform draw.  The
// association order of the sums is part of the result (it decides which index a draw next to a
// boundary gets), so it is the reference's, as restated in oracle_prob_sample: chunks of 8192
// values; quads a, a+b, c+(a+b), (d+c)+(a+b); the quad totals scanned in place by an up-sweep /
// down-sweep tree; a compensated carry between chunks.  One workgroup per row; the tree levels are
// data-parallel (every level touches disjoint pairs), one barrier per level.
constexpr int PS_CHUNK = 8192, PS_THREADS = 512;

__global__ __launch_bounds__(PS_THREADS) void prob_cumsum_kernel(int n, const float *__restrict__ inp,
                                                                float *__restrict__ cum)
{
    __shared__ float val[PS_CHUNK];
    __shared__ float tot[PS_CHUNK / 4];
    const float *p = inp + (size_t)blockIdx.x * n;
    float *c = cum + (size_t)blockIdx.x * n;
    const int t = threadIdx.x;
    float carry = 0.0f, carry2 = 0.0f;      // identical in every thread
    for (int j = 0; j < n; j += PS_CHUNK) {
        const int len = min(n - j, PS_CHUNK);
        const int padded = (len + 3) & ~3, quads = padded >> 2;
        for (int g = t; g < quads; g += PS_THREADS) {
            const int k = 4 * g;
            if (k + 3 < len) {
                const float a = p[j + k], b = p[j + k + 1], cc = p[j + k + 2], d = p[j + k + 3];
                const float ab = b + a, dc = d + cc;
                val[k] = a;
                val[k + 1] = ab;
                val[k + 2] = cc + ab;
                val[k + 3] = dc + ab;
                tot[g] = dc + ab;
            } else {                          // ragged last quad: left to right, padded with its total
                float v = 0.0f;
                for (int e = k; e < len; ++e) {
                    v = v + p[j + e];
                    val[e] = v;
                }
                for (int e = len; e < padded; ++e)
                    val[e] = v;
                tot[g] = v;
            }
        }
        int u = 0;
        for (; (2 << u) <= quads; ++u) {      // up-sweep
            __syncthreads();
            for (int k = t; k < (quads >> (u + 1)); k += PS_THREADS)
                tot[(((k << 1) + 2) << u) - 1] += tot[(((k << 1) + 1) << u) - 1];
        }
        for (--u; u >= 0; --u) {              // down-sweep
            __syncthreads();
            for (int k = t; k < ((quads - (1 << u)) >> (u + 1)); k += PS_THREADS)
                tot[(((k << 1) + 3) << u) - 1] += tot[(((k << 1) + 2) << u) - 1];
        }
        __syncthreads();
        for (int e = t; e < len; e += PS_THREADS) {
            const int g = e >> 2;
            const float v = g > 0 ? val[e] + tot[g - 1] : val[e];
            c[j + e] = v + carry;
        }
        const float tt = tot[quads - 1] + carry2;
        const float r2 = carry + tt;
        carry2 = tt - (r2 - carry);
        carry = r2;
        __syncthreads();                      // val/tot are rewritten by the next chunk
    }
}

__global__ __launch_bounds__(256) void prob_search_kernel(int n, int m, const float *__restrict__ cum,
                                                         const float *__restrict__ draws, int *__restrict__ out)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= m)
        return;
    const float *c = cum + (size_t)blockIdx.y * n;
    int base = 1;
    while (base < n)
        base <<= 1;
    const float key = draws[(size_t)blockIdx.y * m + q] * c[n - 1];
    int r = n - 1;
    for (int k = base; k >= 1; k >>= 1)
        if (r >= k && c[r - k] >= key)
            r -= k;
    out[(size_t)blockIdx.y * m + q] = r;
}

} // namespace cloudaae

using namespace cloudaae;

CLOUDAAE_API int cloudaae_farthest_point_sample(int b, int n, int m, const float *inp, float *temp,
                                                int *out, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_farthest_point_sample";
    CLOUDAAE_REQUIRE(b >= 0 && n >= 0 && m >= 0, name, "negative size");
    if (b == 0 || m == 0)
        return 0;  // tf_sampling_g.cu:106-107
    CLOUDAAE_REQUIRE(n > 0, name, "empty cloud");
    hipStream_t s = (hipStream_t)stream;
    int rc = 0;
    if (n <= 512)
        rc = launch_fps_reg<1>(b, n, m, inp, out, s);
    else if (n <= 1024)
        rc = launch_fps_reg<2>(b, n, m, inp, out, s);
    else if (n <= 2048)
        rc = launch_fps_reg<4>(b, n, m, inp, out, s);
    else if (n <= 4096)
        rc = launch_fps_reg<8>(b, n, m, inp, out, s);
    else if (n <= 8192)
        rc = launch_fps_reg<16>(b, n, m, inp, out, s);
    else if (n <= 16384)
        rc = launch_fps_reg<32>(b, n, m, inp, out, s);
    else {
        CLOUDAAE_REQUIRE(temp != nullptr, name, "n > 16384 needs the 32*n float workspace");
        hipLaunchKernelGGL(fps_big_kernel, dim3(b < 32 ? b : 32), dim3(FPS_THREADS), 0, s, b, n, m,
                           inp, temp, out);
    }
    if (rc)
        return rc;
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_gather_point(int b, int n, int m, const float *inp, const int *idx,
                                       float *out, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_gather_point";
    CLOUDAAE_REQUIRE(b >= 0 && n >= 0 && m >= 0 && b <= 65535, name, "bad size");
    if (b == 0 || m == 0)
        return 0;
    hipLaunchKernelGGL(gather_point_kernel, dim3(ceil_div(m, 256), b), dim3(256), 0,
                       (hipStream_t)stream, n, m, inp, idx, out);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx,
                                            float *inp_g, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_gather_point_grad";
    CLOUDAAE_REQUIRE(b >= 0 && n >= 0 && m >= 0 && b <= 65535, name, "bad size");
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)b * n)
        CLOUDAAE_CHECK_HIP(hipMemsetAsync(inp_g, 0, sizeof(float) * (size_t)b * n * 3, s), name);
    if (b == 0 || m == 0)
        return 0;
    hipLaunchKernelGGL(scatter_add_point_kernel, dim3(ceil_div(m, 256), b), dim3(256), 0, s, n, m,
                       out_g, idx, inp_g);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp,
                                      int *out, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_prob_sample";
    CLOUDAAE_REQUIRE(b >= 0 && n >= 1 && m >= 0 && b <= 65535, name, "bad size");
    if (b == 0)
        return 0;
    CLOUDAAE_REQUIRE(inp_p && temp && (m == 0 || (inp_r && out)), name, "null argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(prob_cumsum_kernel, dim3(b), dim3(PS_THREADS), 0, s, n, inp_p, temp);
    if (m > 0)
        hipLaunchKernelGGL(prob_search_kernel, dim3(ceil_div(m, 256), b), dim3(256), 0, s, n, m, temp, inp_r, out);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}
