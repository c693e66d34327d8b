// bn16.hip -- batch norm + ReLU + mean pool over the points of a cloud on an activation stored as bfloat16.
//
// The dgcnn_agg block of BASELINE configs[2] (models/pointnet_ycb_23_decoder_4.py:410-419: conv2d -> batch norm ->
// ReLU -> reduce_mean over the points) when its 1024-channel output y is kept in HBM as bf16 (gemm_b16.hip): the two
// passes over y read half the bytes, and the gradient dy is written as bf16 for the two gradient products that read
// it.  Training mode only; statistics come from the fp32 accumulators of the product (its column sums), so
// the only difference from bn.hip's arithmetic is that y itself was rounded to bf16 when it was stored.
// A thread owns 8 consecutive channels (one 16-byte load per row), 32 threads span 256 channels, 8 rows in flight per
// workgroup; per-channel arithmetic is bn.hip's (bn_apply_meanpool_kernel, bn_bwd_apply_kernel).
#include "bn_common.h"
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned bf16_pair(float lo, float hi)
{
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 p = {(__bf16)lo, (__bf16)hi};
    unsigned w;
    __builtin_memcpy(&w, &p, 4);
    return w;
}
__device__ __forceinline__ void unpack8(const u32x4 &v, float (&f)[8])
{
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f[2 * m] = bf16_lo(v[m]);
        f[2 * m + 1] = bf16_hi(v[m]);
    }
}

constexpr int B16_U = 4;        // row loads in flight per thread

// grid (C / 256, clouds): pooled[cloud][c] = mean over the cloud's rows of relu(y * sc + sh), and per (cloud, c) the
// three counts the backward pass starts from (rows passing the ReLU, sum of their x_hat, sum of all x_hat)
__global__ __launch_bounds__(256) void bn_apply_meanpool16_kernel(int C, const uint16_t *__restrict__ y, int ldy,
                                                                 const float *__restrict__ scale_shift, int rows,
                                                                 float *__restrict__ pooled,
                                                                 const float *__restrict__ save_mean,
                                                                 const float *__restrict__ save_var,
                                                                 double *__restrict__ pool_stats)
{
    __shared__ double red[8][256];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 256 + 8 * cg;
    float sc[8], sh[8], mean[8], rstd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = scale_shift[c0 + j];
        sh[j] = scale_shift[C + c0 + j];
        mean[j] = save_mean[c0 + j];
        rstd[j] = bn_rsqrt(save_var[c0 + j] + BN_EPS);
    }
    const uint16_t *p = y + ((size_t)blockIdx.y * rows + rl) * ldy + c0;
    float acc[8], cnt[8];
    double sx[8], sall[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[j] = 0.0f;
        cnt[j] = 0.0f;
        sx[j] = 0.0;
        sall[j] = 0.0;
    }
    for (int it = 0; it < rows / (8 * B16_U); ++it) {
        u32x4 v[B16_U];
#pragma unroll
        for (int u = 0; u < B16_U; ++u)
            v[u] = *reinterpret_cast<const u32x4 *>(p + (size_t)(8 * u) * ldy);
        p += (size_t)(8 * B16_U) * ldy;
        float bx[8], ball[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bx[j] = 0.0f;
            ball[j] = 0.0f;
        }
#pragma unroll
        for (int u = 0; u < B16_U; ++u) {
            float f[8];
            unpack8(v[u], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float z = fmaxf(f[j] * sc[j] + sh[j], 0.0f);
                acc[j] = acc[j] + z;
                const float xh = (f[j] - mean[j]) * rstd[j];
                ball[j] += xh;
                cnt[j] += z > 0.0f ? 1.0f : 0.0f;
                bx[j] += z > 0.0f ? xh : 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sx[j] += (double)bx[j];
            sall[j] += (double)ball[j];
        }
    }
    // four column reductions over the 8 row lanes through one 16 KB array (occupancy: the pass is a pure stream)
    const int t = threadIdx.x, c = blockIdx.x * 256 + t;
    double tot[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w > 0)
            __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j)
            red[rl][8 * cg + j] = w == 0 ? (double)acc[j] : w == 1 ? (double)cnt[j] : w == 2 ? sx[j] : sall[j];
        __syncthreads();
        if (w == 0) {       // the activations are summed in fp32, as bn_apply_meanpool_kernel does
            float s = 0.0f;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                s += (float)red[r][t];
            tot[0] = (double)s;
        } else {
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                a += red[r][t];
            tot[w] = a;
        }
    }
    pooled[(size_t)blockIdx.y * C + c] = (float)tot[0] / (float)rows;
    double *ps = pool_stats + (size_t)blockIdx.y * 3 * C + c;
    ps[0] = tot[1];
    ps[C] = tot[2];
    ps[2 * (size_t)C] = tot[3];
}

// grid (C / 256, M / slab): dy = gamma * rstd * ((dz - m1) - x_hat * m2), dz = [relu passes] * dpooled[cloud][c] / rows
__global__ __launch_bounds__(256) void bn_bwd_apply_meanpool16_kernel(int C, const uint16_t *__restrict__ y, int ldy,
                                                                     const float *__restrict__ gamma,
                                                                     const float *__restrict__ beta,
                                                                     const float *__restrict__ save_mean,
                                                                     const float *__restrict__ save_var,
                                                                     const float *__restrict__ m12, int rows,
                                                                     const float *__restrict__ dpooled,
                                                                     uint16_t *__restrict__ dy, int lddy, int slab)
{
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 256 + 8 * cg;
    const int r0 = blockIdx.y * slab;           // slab | rows: the whole slab lies in one cloud
    const int cloud = r0 / rows;
    float sc[8], sh[8], mean[8], rstd[8], m1[8], m2[8], gr[8], gc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bn_scale_shift_of(gamma, beta, save_mean, save_var, c0 + j, sc[j], sh[j]);
        mean[j] = save_mean[c0 + j];
        rstd[j] = bn_rsqrt(save_var[c0 + j] + BN_EPS);
        m1[j] = m12[c0 + j];
        m2[j] = m12[C + c0 + j];
        gr[j] = gamma[c0 + j] * rstd[j];
        gc[j] = dpooled[(size_t)cloud * C + c0 + j] / (float)rows;
    }
    const uint16_t *p = y + (size_t)(r0 + rl) * ldy + c0;
    uint16_t *q = dy + (size_t)(r0 + rl) * lddy + c0;
    for (int it = 0; it < slab / (8 * B16_U); ++it) {
        u32x4 v[B16_U];
#pragma unroll
        for (int u = 0; u < B16_U; ++u)
            v[u] = *reinterpret_cast<const u32x4 *>(p + (size_t)(8 * u) * ldy);
        p += (size_t)(8 * B16_U) * ldy;
#pragma unroll
        for (int u = 0; u < B16_U; ++u) {
            float f[8], d[8];
            unpack8(v[u], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float z = fmaxf(f[j] * sc[j] + sh[j], 0.0f);
                const float dz = z > 0.0f ? gc[j] : 0.0f;
                const float xh = (f[j] - mean[j]) * rstd[j];
                d[j] = gr[j] * ((dz - m1[j]) - xh * m2[j]);
            }
            u32x4 o;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                o[m] = bf16_pair(d[2 * m], d[2 * m + 1]);
            *reinterpret_cast<u32x4 *>(q + (size_t)(8 * u) * lddy) = o;
        }
        q += (size_t)(8 * B16_U) * lddy;
    }
}

} // namespace cloudaae

using namespace cloudaae;

static bool bn16_shape_ok(int M, int C, int ld, int pool_rows)
{
    return M > 0 && C > 0 && C % 256 == 0 && ld >= C && ld % 8 == 0 && pool_rows > 0 && M % pool_rows == 0 &&
           pool_rows % 64 == 0 && M / pool_rows <= 65535;
}

CLOUDAAE_API int cloudaae_bn_meanpool_forward16(int M, int C, const uint16_t *y, int ldy, const float *gamma,
                                                const float *beta, const float *decay, float *ema_mean, float *ema_var,
                                                float *save_mean, float *save_var, int pool_rows, float *pooled,
                                                double *pool_stats, void *workspace, const double *colstats,
                                                int colstats_parts, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_bn_meanpool_forward16";
    CLOUDAAE_REQUIRE(bn16_shape_ok(M, C, ldy, pool_rows), name,
                     "needs C % 256 == 0, 16-byte aligned rows and pool_rows a multiple of 64 dividing M");
    CLOUDAAE_REQUIRE(y && gamma && beta && save_mean && save_var && pooled && pool_stats && workspace, name, "null argument");
    CLOUDAAE_REQUIRE(((uintptr_t)y & 15) == 0, name, "y must be 16-byte aligned");
    CLOUDAAE_REQUIRE(colstats != nullptr && colstats_parts > 0, name, "the column sums of the fp32 product are needed");
    CLOUDAAE_REQUIRE(!ema_mean || decay, name, "EMA update needs the decay scalar");
    hipStream_t s = (hipStream_t)stream;
    double *partial = (double *)workspace;
    float *scale_shift = (float *)(partial + (size_t)BN_MAX_PARTS * 4 * C);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, BN_FIN_CH)), dim3(BN_FIN_THREADS), 0, s, C, colstats,
                       colstats_parts, (double)M, 1, decay, ema_mean, ema_var, gamma, beta, save_mean, save_var, scale_shift);
    hipLaunchKernelGGL(bn_apply_meanpool16_kernel, dim3(C / 256, M / pool_rows), dim3(256), 0, s, C, y, ldy, scale_shift,
                       pool_rows, pooled, save_mean, save_var, pool_stats);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_bn_meanpool_backward16(int M, int C, const uint16_t *y, int ldy, const float *gamma,
                                                 const float *beta, const float *save_mean, const float *save_var,
                                                 int pool_rows, const float *dpooled, uint16_t *dy, int lddy,
                                                 float *dgamma, float *dbeta, float *dbias, int accumulate_param_grads,
                                                 const double *pool_stats, void *workspace, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_bn_meanpool_backward16";
    CLOUDAAE_REQUIRE(bn16_shape_ok(M, C, ldy, pool_rows) && lddy >= C && lddy % 8 == 0, name,
                     "needs C % 256 == 0, 16-byte aligned rows and pool_rows a multiple of 64 dividing M");
    CLOUDAAE_REQUIRE(y && gamma && beta && save_mean && save_var && dpooled && dy && pool_stats && workspace, name,
                     "null argument");
    CLOUDAAE_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)dy & 15) == 0, name, "y and dy must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    float *scratch = (float *)((double *)workspace + (size_t)BN_MAX_PARTS * 4 * C);
    float *m12 = scratch + 2 * (size_t)C;
    const int groups = M / pool_rows;
    const int parts = groups < BN_MAX_PARTS ? groups : BN_MAX_PARTS;
    // (the partial-sum rows are formed inside the finalise: bn_common.h)
    hipLaunchKernelGGL(bn_bwd_finalize_pool_kernel, dim3(ceil_div(C, BN_FIN_CH)), dim3(BN_FIN_THREADS), 0, s, C, groups,
                       pool_rows, parts, dpooled, pool_stats, (double)M, 1, dgamma, dbeta, accumulate_param_grads, m12, dbias,
                       gamma, save_var);
    const int slab = 64;
    hipLaunchKernelGGL(bn_bwd_apply_meanpool16_kernel, dim3(C / 256, M / slab), dim3(256), 0, s, C, y, ldy, gamma, beta,
                       save_mean, save_var, m12, pool_rows, dpooled, dy, lddy, slab);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}
