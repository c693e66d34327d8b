// bn_common.h -- per-channel finalise kernels shared by bn.hip and edgeconv.hip.
// Partial-sum layout everywhere: double partial[parts][2][C].
#pragma once
#include "common.h"
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

constexpr float BN_EPS = 1e-3f;         // tf_util.py:510
constexpr int BN_MAX_PARTS = 128;       // row-slices of the column reductions

// workspace layout (doubles): [parts][2][C] partial sums, a second [parts][2][C] block (third sum of
// the backward pass), then 4*C floats of
// per-channel scratch (scale, shift | m1, m2) packed into 2*C doubles.
__host__ __device__ inline size_t bn_ws_doubles(int C) { return (size_t)BN_MAX_PARTS * 4 * C + 2 * (size_t)C; }

__device__ __forceinline__ float bn_rsqrt(float v) { return 1.0f / sqrtf(v); }

// scale and shift of the normalisation, z = y * sc + sh, exactly as the finalise kernels derive them
// (the backward kernels recompute them per lane instead of spending a launch on a 2*C table)
__device__ __forceinline__ void bn_scale_shift_of(const float *__restrict__ gamma, const float *__restrict__ beta,
                                                  const float *__restrict__ mean, const float *__restrict__ var,
                                                  int c, float &sc, float &sh)
{
    const float inv = gamma[c] * bn_rsqrt(var[c] + BN_EPS);
    sc = inv;
    sh = beta[c] - mean[c] * inv;
}

// Sum the [parts][2][C] partial sums of one group of BN_FIN_CH channels.  A finalise kernel runs
// BN_FIN_THREADS = 1024 threads = 32 channels x 32 part-lanes, each lane with independent accumulators:
// what costs is the CHAIN of dependent partial loads, so the shape is chosen for short chains (a single
// thread walking 1024 loads took ~250 us; 64 channels x 16 lanes 16 us at 256 parts; this one 8 loads
// per lane in two rounds).  Fixed-shape tree, so the result does not depend on scheduling.
constexpr int BN_FIN_CH = 32;
constexpr int BN_FIN_LANES = 32;
constexpr int BN_FIN_THREADS = BN_FIN_CH * BN_FIN_LANES;
__device__ __forceinline__ int bn_fin_channel() { return blockIdx.x * BN_FIN_CH + (int)(threadIdx.x % BN_FIN_CH); }
__device__ __forceinline__ int bn_fin_lane() { return (int)(threadIdx.x / BN_FIN_CH); }

// The reduction itself, over any source of partial-sum rows: A(p), B(p), C3(p) give row p's three terms for this thread's
// channel (C3 only with THREE).  Fixed shape -- two accumulators per lane over rows pl, pl + 32, ... (bn_accumulate_rows), then
// the tree over the lanes (bn_reduce_lanes) -- so every caller (the finalise kernels, a finalise folded into the producing
// kernel, the pooled backward that forms its rows on the fly) gets the same bits from the same rows.
struct BnLaneSums {
    double a, b, c;
};
template <bool THREE, class FA, class FB, class FC>
__device__ __forceinline__ BnLaneSums bn_accumulate_rows(int parts, bool active, int pl, FA A, FB B, FC C3)
{
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0, c0 = 0.0, c1 = 0.0;
    if (active) {
        int p = pl;
        for (; p + BN_FIN_LANES < parts; p += 2 * BN_FIN_LANES) {
            a0 += A(p);
            b0 += B(p);
            a1 += A(p + BN_FIN_LANES);
            b1 += B(p + BN_FIN_LANES);
            if constexpr (THREE) {
                c0 += C3(p);
                c1 += C3(p + BN_FIN_LANES);
            }
        }
        for (; p < parts; p += BN_FIN_LANES) {
            a0 += A(p);
            b0 += B(p);
            if constexpr (THREE)
                c0 += C3(p);
        }
    }
    return BnLaneSums{a0 + a1, b0 + b1, c0 + c1};
}
__device__ __forceinline__ void bn_reduce_lanes(const BnLaneSums &v, int pl, int l, double &s, double &s2, double &s3)
{
    __shared__ double red[3][BN_FIN_LANES][BN_FIN_CH];
    red[0][pl][l] = v.a;
    red[1][pl][l] = v.b;
    red[2][pl][l] = v.c;
    __syncthreads();
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int q = 0; q < BN_FIN_LANES; q += 4) {
        t0 += (red[0][q][l] + red[0][q + 1][l]) + (red[0][q + 2][l] + red[0][q + 3][l]);
        t1 += (red[1][q][l] + red[1][q + 1][l]) + (red[1][q + 2][l] + red[1][q + 3][l]);
        t2 += (red[2][q][l] + red[2][q + 1][l]) + (red[2][q + 2][l] + red[2][q + 3][l]);
    }
    s = t0;
    s2 = t1;
    s3 = t2;
}
template <bool THREE, class FA, class FB, class FC>
__device__ __forceinline__ void bn_reduce_rows(int parts, bool active, int pl, int l, FA A, FB B, FC C3, double &s, double &s2,
                                               double &s3)
{
    bn_reduce_lanes(bn_accumulate_rows<THREE>(parts, active, pl, A, B, C3), pl, l, s, s2, s3);
}

// s = sum_p partial[p][0][c], s2 = sum_p partial[p][1][c]; with partial3 != nullptr also
// s3 = sum_p partial3[p][0][c] in the same pass (its loads travel with the others)
__device__ __forceinline__ void bn_reduce_partials(const double *__restrict__ partial, int parts, int C, int c,
                                                   int pl, double &s, double &s2,
                                                   const double *__restrict__ partial3 = nullptr,
                                                   double *s3 = nullptr)
{
    const int l = threadIdx.x % BN_FIN_CH;
    auto A = [&](int p) { return partial[((size_t)p * 2 + 0) * C + c]; };
    auto B = [&](int p) { return partial[((size_t)p * 2 + 1) * C + c]; };
    auto C3 = [&](int p) { return partial3[((size_t)p * 2 + 0) * C + c]; };
    double t2 = 0.0;
    if (partial3 != nullptr)
        bn_reduce_rows<true>(parts, c < C, pl, l, A, B, C3, s, s2, t2);
    else
        bn_reduce_rows<false>(parts, c < C, pl, l, A, B, C3, s, s2, t2);
    if (s3 != nullptr)
        *s3 = t2;
}

// what one channel's forward finalise writes, from its two sums
__device__ __forceinline__ void bn_finalize_channel(int C, int c, double s, double s2, double count, int training,
                                                    const float *__restrict__ decay, float *__restrict__ ema_mean,
                                                    float *__restrict__ ema_var, const float *__restrict__ gamma,
                                                    const float *__restrict__ beta, float *__restrict__ save_mean,
                                                    float *__restrict__ save_var, float *__restrict__ scale_shift)
{
    float mean, var;
    if (training) {
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

// per-channel finalise (grid = ceil(C/BN_FIN_CH) blocks of BN_FIN_THREADS threads).  training: moments from
// the partial sums + EMA update; inference: moments = EMA shadows.  Also derives inv/shift
// for the apply pass.
static __global__ __launch_bounds__(BN_FIN_THREADS) void bn_finalize_kernel(
    int C, const double *__restrict__ partial, int parts, double count, int training,
    const float *__restrict__ decay, float *__restrict__ ema_mean, float *__restrict__ ema_var,
    const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ save_mean,
    float *__restrict__ save_var, float *__restrict__ scale_shift)
{
    const int c = bn_fin_channel(), pl = bn_fin_lane();
    double s = 0.0, s2 = 0.0;
    if (training)
        bn_reduce_partials(partial, parts, C, c, pl, s, s2);
    if (c >= C || pl != 0)
        return;
    bn_finalize_channel(C, c, s, s2, count, training, decay, ema_mean, ema_var, gamma, beta, save_mean, save_var, scale_shift);
}

// what one channel's backward finalise writes, from its three sums
__device__ __forceinline__ void bn_bwd_finalize_channel(int C, int c, double s, double s2, double s3, double count, int training,
                                                        float *__restrict__ dgamma, float *__restrict__ dbeta, int accumulate,
                                                        float *__restrict__ m12, float *__restrict__ dbias,
                                                        const float *__restrict__ gamma, const float *__restrict__ save_var,
                                                        const double *__restrict__ gsums, double gcount)
{
    if (dbeta != nullptr)
        dbeta[c] = (accumulate ? dbeta[c] : 0.0f) + (float)s;
    if (dgamma != nullptr)
        dgamma[c] = (accumulate ? dgamma[c] : 0.0f) + (float)s2;
    const float m1 = training ? (gsums != nullptr ? (float)(gsums[c] / gcount) : (float)(s / count)) : 0.0f;
    const float m2 = training ? (gsums != nullptr ? (float)(gsums[C + c] / gcount) : (float)(s2 / count)) : 0.0f;
    m12[c] = m1;
    m12[C + c] = m2;
    if (dbias != nullptr) {
        // gradient of a bias added right in front of this BN = sum_r dy = gamma*rstd*((sum dz - M*m1) -
        // m2 * sum xhat): analytically zero, numerically the same round-off a column sum of dy gives
        const double gr = (double)gamma[c] * (double)bn_rsqrt(save_var[c] + BN_EPS);
        dbias[c] = (accumulate ? dbias[c] : 0.0f) + (float)(gr * ((s - count * (double)m1) - (double)m2 * s3));
    }
}

// dbeta = sum dz, dgamma = sum dz*xhat; m1/m2 = their means (0 in inference mode,
// where the statistics do not depend on the batch).  grid = ceil(C/BN_FIN_CH) x BN_FIN_THREADS threads.
static __global__ __launch_bounds__(BN_FIN_THREADS) void bn_bwd_finalize_kernel(
    int C, const double *__restrict__ partial, int parts, double count, int training,
    float *__restrict__ dgamma, float *__restrict__ dbeta, int accumulate, float *__restrict__ m12,
    float *__restrict__ dbias, const float *__restrict__ gamma, const float *__restrict__ save_var,
    const double *__restrict__ gsums = nullptr, double gcount = 0.0)
{
    // gsums != nullptr (SyncBN): gsums[2][C] = the two sums over ALL ranks' rows, gcount = their row count;
    // the means the input gradient needs come from those, dgamma / dbeta / dbias stay this rank's sums
    const int c = bn_fin_channel(), pl = bn_fin_lane();
    double s, s2, s3 = 0.0;
    bn_reduce_partials(partial, parts, C, c, pl, s, s2,
                       dbias != nullptr ? partial + (size_t)BN_MAX_PARTS * 2 * C : nullptr, &s3);
    if (c >= C || pl != 0)
        return;
    bn_bwd_finalize_channel(C, c, s, s2, s3, count, training, dgamma, dbeta, accumulate, m12, dbias, gamma, save_var, gsums,
                            gcount);
}

// Mean pool over groups of `rows` rows with nothing else consuming the activation: the upstream gradient
// of every row of group g is the same number dpooled[g][c] / rows (times the ReLU mask), so the column
// sums the backward pass starts with are that number times what the FORWARD apply pass counted per group
// (rows passing the ReLU, sum of their x_hat, sum of all x_hat).  One partial-sum row per group replaces a
// pass over y (134 MB for dgcnn_agg).
static __global__ __launch_bounds__(256) void bn_bwd_pool_partials_kernel(int C, int groups, int rows,
                                                                  const float *__restrict__ dpooled,
                                                                  const double *__restrict__ pool_stats,
                                                                  double *__restrict__ partial)
{
    // grid.y = parts <= BN_MAX_PARTS partial-sum rows; part p takes the groups p, p + parts, ... (a batch of more
    // than 128 clouds per GPU: BASELINE configs[2] runs 256)
    const int c = blockIdx.x * 256 + threadIdx.x, parts = gridDim.y;
    if (c >= C)
        return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int g = blockIdx.y; g < groups; g += parts) {
        const double gv = (double)(dpooled[(size_t)g * C + c] / (float)rows);
        const double *ps = pool_stats + (size_t)g * 3 * C + c;
        s0 += gv * ps[0];
        s1 += gv * ps[C];
        s2 += ps[2 * (size_t)C];
    }
    partial[((size_t)blockIdx.y * 2 + 0) * C + c] = s0;
    partial[((size_t)blockIdx.y * 2 + 1) * C + c] = s1;
    double *p3 = partial + (size_t)BN_MAX_PARTS * 2 * C;
    p3[((size_t)blockIdx.y * 2 + 0) * C + c] = s2;
    p3[((size_t)blockIdx.y * 2 + 1) * C + c] = 0.0;
}

// The two kernels above in one (no SyncBN exchange between them): the finalise forms partial-sum row p on the fly -- the
// same products in the same order as bn_bwd_pool_partials_kernel stores them -- and reduces the rows as ever: same bits,
// one launch less.
static __global__ __launch_bounds__(BN_FIN_THREADS) void bn_bwd_finalize_pool_kernel(
    int C, int groups, int rows, int parts, const float *__restrict__ dpooled, const double *__restrict__ pool_stats,
    double count, int training, float *__restrict__ dgamma, float *__restrict__ dbeta, int accumulate,
    float *__restrict__ m12, float *__restrict__ dbias, const float *__restrict__ gamma, const float *__restrict__ save_var)
{
    const int c = bn_fin_channel(), pl = bn_fin_lane(), l = threadIdx.x % BN_FIN_CH;
    auto A = [&](int p) {
        double acc = 0.0;
        for (int g = p; g < groups; g += parts)
            acc += (double)(dpooled[(size_t)g * C + c] / (float)rows) * pool_stats[(size_t)g * 3 * C + c];
        return acc;
    };
    auto B = [&](int p) {
        double acc = 0.0;
        for (int g = p; g < groups; g += parts)
            acc += (double)(dpooled[(size_t)g * C + c] / (float)rows) * pool_stats[(size_t)g * 3 * C + C + c];
        return acc;
    };
    auto C3 = [&](int p) {
        double acc = 0.0;
        for (int g = p; g < groups; g += parts)
            acc += pool_stats[(size_t)g * 3 * C + 2 * (size_t)C + c];
        return acc;
    };
    double s, s2, s3;
    if (dbias != nullptr)
        bn_reduce_rows<true>(parts, c < C, pl, l, A, B, C3, s, s2, s3);
    else
        bn_reduce_rows<false>(parts, c < C, pl, l, A, B, C3, s, s2, s3);
    if (c >= C || pl != 0)
        return;
    bn_bwd_finalize_channel(C, c, s, s2, s3, count, training, dgamma, dbeta, accumulate, m12, dbias, gamma, save_var, nullptr,
                            0.0);
}

// ---- SyncBN: the sums of a layer leave for the other ranks between its statistics pass and its finalise ----
// sums[2][C] = sum over the parts of partial[parts][2][C]  (grid = ceil(C/BN_FIN_CH) x BN_FIN_THREADS)
static __global__ __launch_bounds__(BN_FIN_THREADS) void bn_sum_partials_kernel(int C, const double *__restrict__ partial,
                                                                              int parts, double *__restrict__ sums)
{
    const int c = bn_fin_channel(), pl = bn_fin_lane();
    double s, s2;
    bn_reduce_partials(partial, parts, C, c, pl, s, s2);
    if (c >= C || pl != 0)
        return;
    sums[c] = s;
    sums[C + c] = s2;
}

// partial[parts][2][C] -> sync->buf[2][C], summed over all ranks by the host's all-reduce.  Afterwards the
// caller finalises from (sync->buf, 1 part, count * world).
static inline int bn_sync_exchange(const char *name, const cloudaae_bn_sync *sync, int C, const double *partial, int parts,
                                   hipStream_t s)
{
    CLOUDAAE_REQUIRE(sync->allreduce != nullptr && sync->buf != nullptr && sync->world >= 1, name,
                     "bad cloudaae_bn_sync (allreduce, buf and world >= 1 are needed)");
    hipLaunchKernelGGL(bn_sum_partials_kernel, dim3(ceil_div(C, BN_FIN_CH)), dim3(BN_FIN_THREADS), 0, s, C, partial,
                       parts, sync->buf);
    CLOUDAAE_CHECK_LAUNCH(name);
    const int rc = sync->allreduce(sync->ctx, sync->buf, 2 * C, (cloudaae_stream_t)s);
    if (rc != 0) {
        set_error("%s: the host's all-reduce callback failed (%d)", name, rc);
        return rc > 0 ? rc : (int)hipErrorUnknown;
    }
    return 0;
}

} // namespace cloudaae
