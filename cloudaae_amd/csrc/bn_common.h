// bn_common.h -- per-channel finalise kernels shared by bn.hip and edgeconv.hip.
// Partial-sum layout everywhere: double partial[parts][2][C].
#pragma once
#include "common.h"

namespace cloudaae {

constexpr float BN_EPS = 1e-3f;         // tf_util.py:510
constexpr int BN_MAX_PARTS = 128;       // row-slices of the column reductions

// workspace layout (doubles): [parts][2][C] partial sums, then 4*C floats of
// per-channel scratch (scale, shift | m1, m2) packed into 2*C doubles.
__host__ __device__ inline size_t bn_ws_doubles(int C) { return (size_t)BN_MAX_PARTS * 2 * C + 2 * (size_t)C; }

__device__ __forceinline__ float bn_rsqrt(float v) { return 1.0f / sqrtf(v); }

// per-channel finalise.  training: moments from the partial sums + EMA update;
// inference: moments = EMA shadows.  Also derives inv/shift for the apply pass.
static __global__ void bn_finalize_kernel(int C, const double *__restrict__ partial, int parts, double count,
                                   int training, const float *__restrict__ decay,
                                   float *__restrict__ ema_mean, float *__restrict__ ema_var,
                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                   float *__restrict__ save_mean, float *__restrict__ save_var,
                                   float *__restrict__ scale_shift)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C)
        return;
    float mean, var;
    if (training) {
        double s = 0.0, s2 = 0.0;
        for (int p = 0; p < parts; ++p) {
            s += partial[((size_t)p * 2 + 0) * C + c];
            s2 += partial[((size_t)p * 2 + 1) * C + c];
        }
        const double mu = s / count;
        double v = s2 / count - mu * mu;
        v = v > 0.0 ? v : 0.0;
        mean = (float)mu;
        var = (float)v;
        if (ema_mean != nullptr) {
            const float om = 1.0f - decay[0];
            ema_mean[c] = ema_mean[c] - (ema_mean[c] - mean) * om;
            ema_var[c] = ema_var[c] - (ema_var[c] - var) * om;
        }
    } else {
        mean = ema_mean[c];
        var = ema_var[c];
    }
    save_mean[c] = mean;
    save_var[c] = var;
    const float inv = gamma[c] * bn_rsqrt(var + BN_EPS);
    scale_shift[c] = inv;
    scale_shift[C + c] = beta[c] - mean * inv;
}

// inv / shift from saved moments (backward recomputes them: its workspace may differ)
static __global__ void bn_scale_shift_kernel(int C, const float *__restrict__ gamma,
                                      const float *__restrict__ beta, const float *__restrict__ mean,
                                      const float *__restrict__ var, float *__restrict__ scale_shift)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C)
        return;
    const float inv = gamma[c] * bn_rsqrt(var[c] + BN_EPS);
    scale_shift[c] = inv;
    scale_shift[C + c] = beta[c] - mean[c] * inv;
}

// dbeta = sum dz, dgamma = sum dz*xhat; m1/m2 = their means (0 in inference mode,
// where the statistics do not depend on the batch)
static __global__ void bn_bwd_finalize_kernel(int C, const double *__restrict__ partial, int parts, double count,
                                       int training, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                       int accumulate, float *__restrict__ m12)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C)
        return;
    double s = 0.0, s2 = 0.0;
    for (int p = 0; p < parts; ++p) {
        s += partial[((size_t)p * 2 + 0) * C + c];
        s2 += partial[((size_t)p * 2 + 1) * C + c];
    }
    if (dbeta != nullptr)
        dbeta[c] = (accumulate ? dbeta[c] : 0.0f) + (float)s;
    if (dgamma != nullptr)
        dgamma[c] = (accumulate ? dgamma[c] : 0.0f) + (float)s2;
    m12[c] = training ? (float)(s / count) : 0.0f;
    m12[C + c] = training ? (float)(s2 / count) : 0.0f;
}


} // namespace cloudaae
