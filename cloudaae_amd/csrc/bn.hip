// bn.hip -- batch normalisation (+ReLU, +pool over points) on dense [M,C] rows.
//
// Replaces batch_norm_template and its wrappers (reference utils/tf_util.py:473-570)
// as used after every conv2d / fully_connected (tf_util.py:168-173, 354-357), and the
// symmetric pooling over the point axis that follows dgcnn_agg / pn_conv5
// (models/pointnet_ycb_23_decoder_4.py:419 mean, :684 and :59-60 max).
//   training : mean, var = tf.nn.moments (biased variance) over all M rows;
//              EMA shadows   s <- s - (s - stat) * (1 - decay)   (assign_moving_average,
//              shadows start at 0, no zero-debias: tf_util.py:493-500)
//   inference: mean, var = EMA shadows (tf_util.py:507-509)
//   output   : tf.nn.batch_normalization with eps 1e-3 (tf_util.py:510):
//              inv = gamma * rsqrt(var + eps);  z = y * inv + (beta - mean * inv)
// Forward is a column reduction (fp64 partial sums, two-level, no atomics, so the
// moments are deterministic), one tiny per-channel finalise, and one streaming
// apply pass that can write the activation, its pool over groups of `pool_rows`
// rows, or both.  Backward is the same shape.  All passes are HBM streaming:
// a wave reads 64 consecutive channels of a row (256 B), four rows in flight per
// workgroup.
#include "bn_common.h"
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

constexpr int BN_U = 8;   // independent row loads in flight per thread in the streaming passes

// ---- forward: column sums ------------------------------------------------------
__global__ __launch_bounds__(256) void bn_colsum_kernel(int M, int C, const float *__restrict__ y,
                                                       int ldy, double *__restrict__ partial, int parts)
{
    __shared__ double red[2][4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s = 0.0, s2 = 0.0;
    if (c < C) {
        // BN_U loads in flight per thread: these passes are pure HBM streaming and a single
        // dependent load per iteration leaves them latency-bound (1.6 TB/s measured)
        // (a block walks one contiguous slice of rows: consecutive 256-byte row pieces of the same
        // channel group, kinder to DRAM pages than rows interleaved across all blocks)
        const int chunk = ((M + parts - 1) / parts + 3) / 4 * 4;
        const int r_lo = blockIdx.y * chunk, r_hi = min(M, r_lo + chunk);
        for (int r = r_lo + rl; r < r_hi; r += 4 * BN_U) {
            float v[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                const int rr = r + 4 * u;
                v[u] = rr < r_hi ? y[(size_t)rr * ldy + c] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                s += (double)v[u];
                s2 += (double)v[u] * (double)v[u];
            }
        }
    }
    red[0][rl][lane] = s;
    red[1][rl][lane] = s2;
    __syncthreads();
    if (rl == 0 && c < C) {
        partial[((size_t)blockIdx.y * 2 + 0) * C + c] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        partial[((size_t)blockIdx.y * 2 + 1) * C + c] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
    }
}

// ---- forward: apply (+ReLU), optional activation output, optional pooling ---------
// grid (ceil(C/64), groups); block 64 channels x 4 row lanes; a group is
// `rows` consecutive rows (the N points of one cloud when pooling, else a slab).
template <int POOL>  // 0 none, 1 mean, 2 max
__global__ __launch_bounds__(256) void bn_apply_kernel(int M, int C, const float *__restrict__ y,
                                                      int ldy, const float *__restrict__ scale_shift,
                                                      int relu, float *__restrict__ out, int ldo,
                                                      int rows, float *__restrict__ pooled,
                                                      float *__restrict__ ties, const float *__restrict__ save_mean,
                                                      const float *__restrict__ save_var,
                                                      double *__restrict__ pool_stats)
{
    __shared__ float red[4][64];
    __shared__ float redc[4][64];
    __shared__ double redd[2][4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows;
    const int r1 = min(M, r0 + rows);
    float acc = POOL == 2 ? -__builtin_inff() : 0.0f;
    float cnt = 0.0f;
    // mean pool, training: what the backward pass needs of this group besides the pooled value -- rows
    // that pass the ReLU, the sum of their x_hat, the sum of all x_hat (see bn_bwd_pool_partials_kernel)
    const bool stats = POOL == 1 && pool_stats != nullptr;
    double sx = 0.0, sall = 0.0;
    if (c < C) {
        const float sc = scale_shift[c], sh = scale_shift[C + c];
        const float mean = stats ? save_mean[c] : 0.0f, rstd = stats ? bn_rsqrt(save_var[c] + BN_EPS) : 0.0f;
        for (int rb = r0 + rl; rb < r1; rb += 4 * BN_U) {
            float v[BN_U];
            float bx = 0.0f, ball = 0.0f;
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                const int r = rb + 4 * u;
                v[u] = r < r1 ? y[(size_t)r * ldy + c] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                const int r = rb + 4 * u;
                if (r < r1) {
                    float z = v[u] * sc + sh;
                    if (relu)
                        z = fmaxf(z, 0.0f);
                    if (out != nullptr)
                        out[(size_t)r * ldo + c] = z;
                    if (POOL == 1)
                        acc = acc + z;
                    if (stats) {    // fp32 inside the batch of BN_U rows, fp64 across batches
                        const float xh = (v[u] - mean) * rstd;
                        ball += xh;
                        cnt += z > 0.0f ? 1.0f : 0.0f;
                        bx += z > 0.0f ? xh : 0.0f;
                    }
                    if (POOL == 2) {
                        if (z > acc) {
                            acc = z;
                            cnt = 1.0f;
                        } else if (z == acc) {
                            cnt += 1.0f;
                        }
                    }
                }
            }
            if (stats) {
                sx += (double)bx;
                sall += (double)ball;
            }
        }
    }
    if (POOL != 0) {
        red[rl][lane] = acc;
        redc[rl][lane] = cnt;
        if (stats) {
            redd[0][rl][lane] = sx;
            redd[1][rl][lane] = sall;
        }
        __syncthreads();
        if (rl == 0 && c < C) {
            if (POOL == 1) {
                const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
                pooled[(size_t)blockIdx.y * C + c] = s / (float)(r1 - r0);
                if (stats) {
                    double *ps = pool_stats + (size_t)blockIdx.y * 3 * C + c;
                    ps[0] = (double)((redc[0][lane] + redc[1][lane]) + (redc[2][lane] + redc[3][lane]));
                    ps[C] = (redd[0][0][lane] + redd[0][1][lane]) + (redd[0][2][lane] + redd[0][3][lane]);
                    ps[2 * (size_t)C] = (redd[1][0][lane] + redd[1][1][lane]) + (redd[1][2][lane] + redd[1][3][lane]);
                }
            } else {
                float m = red[0][lane], n = redc[0][lane];
                for (int i = 1; i < 4; ++i) {
                    if (red[i][lane] > m) {
                        m = red[i][lane];
                        n = redc[i][lane];
                    } else if (red[i][lane] == m) {
                        n += redc[i][lane];
                    }
                }
                pooled[(size_t)blockIdx.y * C + c] = m;
                if (ties != nullptr)
                    ties[(size_t)blockIdx.y * C + c] = n;
            }
        }
    }
}

// The hot instance of the kernel above, without a branch in its loop: mean pool + ReLU in training mode,
// activation not written, per-group statistics wanted, whole batches of rows, whole channel blocks
// (dgcnn_agg: 1024 rows per cloud, 1024 channels; the generic kernel ran it at 3.0 TB/s, guard branches
// around every load and every optional output).  Same order of operations, same results.
__global__ __launch_bounds__(256) void bn_apply_meanpool_kernel(int C, const float *__restrict__ y, int ldy,
                                                               const float *__restrict__ scale_shift, int rows,
                                                               float *__restrict__ pooled,
                                                               const float *__restrict__ save_mean,
                                                               const float *__restrict__ save_var,
                                                               double *__restrict__ pool_stats)
{
    __shared__ float red[2][4][64];
    __shared__ double redd[2][4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const float sc = scale_shift[c], sh = scale_shift[C + c];
    const float mean = save_mean[c], rstd = bn_rsqrt(save_var[c] + BN_EPS);
    const float *p = y + ((size_t)blockIdx.y * rows + rl) * ldy + c;
    float acc = 0.0f, cnt = 0.0f;
    double sx = 0.0, sall = 0.0;
    for (int it = 0; it < rows / (4 * BN_U); ++it) {
        float v[BN_U];
#pragma unroll
        for (int u = 0; u < BN_U; ++u)
            v[u] = p[(size_t)(4 * u) * ldy];
        p += (size_t)(4 * BN_U) * ldy;
        float bx = 0.0f, ball = 0.0f;
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            const float z = fmaxf(v[u] * sc + sh, 0.0f);
            acc = acc + z;
            const float xh = (v[u] - mean) * rstd;
            ball += xh;
            cnt += z > 0.0f ? 1.0f : 0.0f;
            bx += z > 0.0f ? xh : 0.0f;
        }
        sx += (double)bx;
        sall += (double)ball;
    }
    red[0][rl][lane] = acc;
    red[1][rl][lane] = cnt;
    redd[0][rl][lane] = sx;
    redd[1][rl][lane] = sall;
    __syncthreads();
    if (rl == 0) {
        const float s = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        pooled[(size_t)blockIdx.y * C + c] = s / (float)rows;
        double *ps = pool_stats + (size_t)blockIdx.y * 3 * C + c;
        ps[0] = (double)((red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]));
        ps[C] = (redd[0][0][lane] + redd[0][1][lane]) + (redd[0][2][lane] + redd[0][3][lane]);
        ps[2 * (size_t)C] = (redd[1][0][lane] + redd[1][1][lane]) + (redd[1][2][lane] + redd[1][3][lane]);
    }
}

// ---- backward ---------------------------------------------------------------------
// upstream gradient of the activation at (r, c):
//   dout[r][c]                                   (plain activation output)
// + dpooled[g][c] / rows                         (mean pool, g = r / rows)
// + dpooled[g][c] * [z == max] / ties            (max pool: tf.reduce_max shares
//                                                 the gradient among equal maxima)
// times the ReLU mask [z > 0].
struct BnBwdArgs {
    int M, C, ldy, lddo, rows, rows_shift, pool, relu, training, want_xhat_sum;   // rows_shift: log2(rows) or -1
    const float *y, *dout, *dpooled, *pooled, *ties;
    const float *gamma, *beta, *save_mean, *save_var;
};

__device__ __forceinline__ float bn_upstream(const BnBwdArgs &a, int r, int c, float z)
{
    float g = 0.0f;
    if (a.dout != nullptr)
        g = a.dout[(size_t)r * a.lddo + c];
    // group of row r (a power-of-two group size -- N points per cloud usually is -- avoids an integer
    // division per element of a 134 MB tensor)
    const int grp = a.rows_shift >= 0 ? (r >> a.rows_shift) : (r / a.rows);
    if (a.pool == 1) {
        g += a.dpooled[(size_t)grp * a.C + c] / (float)a.rows;
    } else if (a.pool == 2) {
        const size_t gi = (size_t)grp * a.C + c;
        if (z == a.pooled[gi])
            g += a.dpooled[gi] / a.ties[gi];
    }
    if (a.relu && !(z > 0.0f))
        g = 0.0f;
    return g;
}

__global__ __launch_bounds__(256) void bn_bwd_colsum_kernel(BnBwdArgs a, double *__restrict__ partial,
                                                           int parts)
{
    __shared__ double red[3][4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < a.C) {
        float sc, sh;
        bn_scale_shift_of(a.gamma, a.beta, a.save_mean, a.save_var, c, sc, sh);
        const float mean = a.save_mean[c], rstd = bn_rsqrt(a.save_var[c] + BN_EPS);
        // a block walks one contiguous slice of rows; when the slice lies inside one pooling group
        // (256 of a cloud's 1024 points, typically) and the only upstream is the mean pool, its
        // gradient dpooled[g][c] / rows is a per-lane constant instead of a load and a divide per element
        const int chunk = ((a.M + parts - 1) / parts + 3) / 4 * 4;
        const int r_lo = blockIdx.y * chunk, r_hi = min(a.M, r_lo + chunk);
        const int g_lo = r_lo < a.M ? (a.rows_shift >= 0 ? (r_lo >> a.rows_shift) : r_lo / a.rows) : 0;
        const int g_hi = r_hi > r_lo ? (a.rows_shift >= 0 ? ((r_hi - 1) >> a.rows_shift) : (r_hi - 1) / a.rows) : g_lo;
        const bool hoist = a.pool == 1 && a.dout == nullptr && g_lo == g_hi;
        const float gconst = hoist && r_lo < a.M ? a.dpooled[(size_t)g_lo * a.C + c] / (float)a.rows : 0.0f;
        for (int rb = r_lo + rl; rb < r_hi; rb += 4 * BN_U) {
            float v[BN_U];
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                const int r = rb + 4 * u;
                v[u] = r < r_hi ? a.y[(size_t)r * a.ldy + c] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < BN_U; ++u) {
                const int r = rb + 4 * u;
                if (r < r_hi) {
                    float z = v[u] * sc + sh;
                    if (a.relu)
                        z = fmaxf(z, 0.0f);
                    const float dz = hoist ? ((a.relu && !(z > 0.0f)) ? 0.0f : gconst) : bn_upstream(a, r, c, z);
                    const float xh = (v[u] - mean) * rstd;
                    s += (double)dz;
                    s2 += (double)dz * (double)xh;
                    if (a.want_xhat_sum)
                        s3 += (double)xh;
                }
            }
        }
    }
    red[0][rl][lane] = s;
    red[1][rl][lane] = s2;
    red[2][rl][lane] = s3;
    __syncthreads();
    if (rl == 0 && c < a.C) {
        partial[((size_t)blockIdx.y * 2 + 0) * a.C + c] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        partial[((size_t)blockIdx.y * 2 + 1) * a.C + c] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
        if (a.want_xhat_sum) {      // third sum (sum of x_hat, ~0) for the gradient of a bias in front of this BN
            double *p3 = partial + (size_t)BN_MAX_PARTS * 2 * a.C;
            p3[((size_t)blockIdx.y * 2 + 0) * a.C + c] = (red[2][0][lane] + red[2][1][lane]) + (red[2][2][lane] + red[2][3][lane]);
            p3[((size_t)blockIdx.y * 2 + 1) * a.C + c] = 0.0;
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a, const float *__restrict__ m12,
                                                          float *__restrict__ dy, int lddy, int slab)
{
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    if (c >= a.C)
        return;
    float sc, sh;
    bn_scale_shift_of(a.gamma, a.beta, a.save_mean, a.save_var, c, sc, sh);
    const float mean = a.save_mean[c], rstd = bn_rsqrt(a.save_var[c] + BN_EPS);
    const float m1 = m12[c], m2 = m12[a.C + c];
    const float gr = a.gamma[c] * rstd;
    const int r0 = blockIdx.y * slab, r1 = min(a.M, r0 + slab);
    const int g_lo = a.rows_shift >= 0 ? (r0 >> a.rows_shift) : r0 / a.rows;
    const int g_hi = a.rows_shift >= 0 ? ((r1 - 1) >> a.rows_shift) : (r1 - 1) / a.rows;
    const bool hoist = a.pool == 1 && a.dout == nullptr && g_lo == g_hi;      // see bn_bwd_colsum_kernel
    const float gconst = hoist ? a.dpooled[(size_t)g_lo * a.C + c] / (float)a.rows : 0.0f;
    for (int rb = r0 + rl; rb < r1; rb += 4 * BN_U) {
        float v[BN_U];
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            const int r = rb + 4 * u;
            v[u] = r < r1 ? a.y[(size_t)r * a.ldy + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < BN_U; ++u) {
            const int r = rb + 4 * u;
            if (r < r1) {
                float z = v[u] * sc + sh;
                if (a.relu)
                    z = fmaxf(z, 0.0f);
                const float dz = hoist ? ((a.relu && !(z > 0.0f)) ? 0.0f : gconst) : bn_upstream(a, r, c, z);
                const float xh = (v[u] - mean) * rstd;
                dy[(size_t)r * lddy + c] = gr * ((dz - m1) - xh * m2);
            }
        }
    }
}

// plain column sums (bias gradients): out[c] (+)= sum_r x[r][c]; grid = ceil(C/64) x 256
__global__ __launch_bounds__(BN_FIN_THREADS) void colsum_finalize_kernel(int C, const double *__restrict__ partial,
                                                             int parts, float *__restrict__ out, int accumulate)
{
    const int c = bn_fin_channel(), pl = bn_fin_lane();
    double s, s2;
    bn_reduce_partials(partial, parts, C, c, pl, s, s2);
    if (c >= C || pl != 0)
        return;
    out[c] = (accumulate ? out[c] : 0.0f) + (float)s;
}

// ---- small batches (the FC layers: M = batch per GPU) -----------------------------------------
// With M <= 128 rows a 64-channel column block fits in the registers of one workgroup
// (64 channels x 4 row lanes, <= 32 rows per thread), so moments, EMA update, normalise(+ReLU)
// are ONE launch instead of three, and the backward ONE instead of four.  At B=32 that is
// 6 layers x 5 launches of ~6 us each per step.  Same arithmetic as the large-M path (fp64
// sums, then the fp32 formulas), so results are identical up to the summation order.
constexpr int BN_SMALL_M = 128;
constexpr int BN_SMALL_R = BN_SMALL_M / 4;

__global__ __launch_bounds__(256) void bn_small_fwd_kernel(int M, int C, const float *__restrict__ y, int ldy,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, int training,
                                                          const float *__restrict__ decay,
                                                          float *__restrict__ ema_mean, float *__restrict__ ema_var,
                                                          float *__restrict__ save_mean, float *__restrict__ save_var,
                                                          int relu, float *__restrict__ out, int ldo)
{
    __shared__ double red[2][4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool ok = c < C;
    float v[BN_SMALL_R];
    double s = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < BN_SMALL_R; ++i) {
        const int r = rl + 4 * i;
        v[i] = (ok && r < M) ? y[(size_t)r * ldy + c] : 0.0f;
        s += (double)v[i];
        s2 += (double)v[i] * (double)v[i];
    }
    float mean, var;
    if (training) {
        red[0][rl][lane] = s;
        red[1][rl][lane] = s2;
        __syncthreads();
        const double ts = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        const double ts2 = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
        const double mu = ts / (double)M;
        double vv = ts2 / (double)M - mu * mu;
        vv = vv > 0.0 ? vv : 0.0;
        mean = (float)mu;
        var = (float)vv;
        if (ok && rl == 0 && ema_mean != nullptr) {
            const float om = 1.0f - decay[0];
            ema_mean[c] = ema_mean[c] - (ema_mean[c] - mean) * om;
            ema_var[c] = ema_var[c] - (ema_var[c] - var) * om;
        }
    } else {
        mean = ok ? ema_mean[c] : 0.0f;
        var = ok ? ema_var[c] : 1.0f;
    }
    if (!ok)
        return;
    if (rl == 0) {
        save_mean[c] = mean;
        save_var[c] = var;
    }
    const float inv = gamma[c] * bn_rsqrt(var + BN_EPS);
    const float sh = beta[c] - mean * inv;
#pragma unroll
    for (int i = 0; i < BN_SMALL_R; ++i) {
        const int r = rl + 4 * i;
        if (r < M) {
            float z = v[i] * inv + sh;
            if (relu)
                z = fmaxf(z, 0.0f);
            out[(size_t)r * ldo + c] = z;
        }
    }
}

__global__ __launch_bounds__(256) void bn_small_bwd_kernel(int M, int C, const float *__restrict__ y, int ldy,
                                                          const float *__restrict__ gamma,
                                                          const float *__restrict__ beta,
                                                          const float *__restrict__ save_mean,
                                                          const float *__restrict__ save_var, int training, int relu,
                                                          const float *__restrict__ dout, int lddo,
                                                          float *__restrict__ dy, int lddy,
                                                          float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                          float *__restrict__ dbias, int accumulate)
{
    __shared__ double red[2][4][64];
    const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool ok = c < C;
    const float mean = ok ? save_mean[c] : 0.0f, var = ok ? save_var[c] : 1.0f;
    const float g = ok ? gamma[c] : 0.0f, b = ok ? beta[c] : 0.0f;
    const float rstd = bn_rsqrt(var + BN_EPS);
    const float inv = g * rstd, sh = b - mean * inv;
    float xh[BN_SMALL_R], dz[BN_SMALL_R];
    double s = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < BN_SMALL_R; ++i) {
        const int r = rl + 4 * i;
        const bool in = ok && r < M;
        const float v = in ? y[(size_t)r * ldy + c] : 0.0f;
        float z = v * inv + sh;
        if (relu)
            z = fmaxf(z, 0.0f);
        float d = in ? dout[(size_t)r * lddo + c] : 0.0f;
        if (relu && !(z > 0.0f))
            d = 0.0f;
        dz[i] = d;
        xh[i] = (v - mean) * rstd;
        if (in) {
            s += (double)d;
            s2 += (double)d * (double)xh[i];
        }
    }
    red[0][rl][lane] = s;
    red[1][rl][lane] = s2;
    __syncthreads();
    if (!ok)
        return;
    const double ts = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
    const double ts2 = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
    if (rl == 0) {
        if (dbeta != nullptr)
            dbeta[c] = (accumulate ? dbeta[c] : 0.0f) + (float)ts;
        if (dgamma != nullptr)
            dgamma[c] = (accumulate ? dgamma[c] : 0.0f) + (float)ts2;
    }
    const float m1 = training ? (float)(ts / (double)M) : 0.0f;
    const float m2 = training ? (float)(ts2 / (double)M) : 0.0f;
    const float gr = g * rstd;
    double sdy = 0.0;
#pragma unroll
    for (int i = 0; i < BN_SMALL_R; ++i) {
        const int r = rl + 4 * i;
        if (r < M) {
            const float v = gr * ((dz[i] - m1) - xh[i] * m2);
            dy[(size_t)r * lddy + c] = v;
            sdy += (double)v;
        }
    }
    if (dbias != nullptr) {     // gradient of a bias added right in front of this BN: column sum of dy
        __syncthreads();
        red[0][rl][lane] = sdy;
        __syncthreads();
        if (rl == 0)
            dbias[c] = (accumulate ? dbias[c] : 0.0f) +
                       (float)((red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]));
    }
}

static int bn_parts(int M)
{
    int p = M / 64;
    if (p < 1)
        p = 1;
    if (p > BN_MAX_PARTS)
        p = BN_MAX_PARTS;
    return p;
}

} // namespace cloudaae

using namespace cloudaae;

CLOUDAAE_API long long cloudaae_bn_workspace_bytes(int C) { return (long long)(bn_ws_doubles(C) * sizeof(double)); }

// colstats != NULL: colstats[colstats_parts][2][C] already holds the column sums of y (written by
// cloudaae_gemm_f32_colstats): the statistics pass over y is skipped
static int bn_forward_impl(const char *name, int M, int C, const float *y, int ldy, const float *gamma,
                           const float *beta, int training, const float *decay, float *ema_mean, float *ema_var,
                           float *save_mean, float *save_var, int relu, float *out, int ldo, int pool_rows,
                           int pool_mode, float *pooled, float *tie_count, double *pool_stats, void *workspace,
                           const double *colstats, int colstats_parts, const cloudaae_bn_sync *sync,
                           cloudaae_stream_t stream)
{
    CLOUDAAE_REQUIRE(M > 0 && C > 0 && ldy >= C, name, "bad size");
    CLOUDAAE_REQUIRE(workspace != nullptr && gamma && beta && save_mean && save_var, name, "null argument");
    CLOUDAAE_REQUIRE(training || (ema_mean && ema_var), name, "inference needs the EMA statistics");
    CLOUDAAE_REQUIRE(!training || !ema_mean || decay, name, "EMA update needs the decay scalar");
    CLOUDAAE_REQUIRE(pool_mode >= 0 && pool_mode <= 2, name, "pool_mode must be 0 (none), 1 (mean) or 2 (max)");
    CLOUDAAE_REQUIRE(pool_mode == 0 || (pool_rows > 0 && M % pool_rows == 0 && pooled), name,
                     "pooling needs pool_rows | M and an output");
    CLOUDAAE_REQUIRE(pool_mode != 0 || out != nullptr, name, "no output requested");
    hipStream_t s = (hipStream_t)stream;
    if (!(training && sync != nullptr))
        sync = nullptr;                        // inference: the moments are the EMA shadows, nothing to exchange
    if (pool_mode == 0 && M <= BN_SMALL_M && sync == nullptr) {   // FC layers: one fused launch
        hipLaunchKernelGGL(bn_small_fwd_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, s, M, C, y, ldy, gamma, beta,
                           training, decay, ema_mean, ema_var, save_mean, save_var, relu, out, ldo);
        CLOUDAAE_CHECK_LAUNCH(name);
        return 0;
    }
    double *partial = (double *)workspace;
    float *scale_shift = (float *)(partial + (size_t)BN_MAX_PARTS * 4 * C);
    const int parts = colstats != nullptr ? colstats_parts : bn_parts(M);
    const int cb = ceil_div(C, 64);
    if (training && colstats == nullptr)
        hipLaunchKernelGGL(bn_colsum_kernel, dim3(cb, parts), dim3(256), 0, s, M, C, y, ldy, partial, parts);
    const double *sums = colstats != nullptr ? colstats : partial;
    int fin_parts = parts;
    double count = (double)M;
    if (sync != nullptr) {      // SyncBN: this rank's sums -> the sums of the global batch
        if (int rc = bn_sync_exchange(name, sync, C, sums, parts, s))
            return rc;
        sums = sync->buf;
        fin_parts = 1;
        count *= (double)sync->world;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, BN_FIN_CH)), dim3(BN_FIN_THREADS), 0, s, C, sums, fin_parts,
                       count, training, decay, ema_mean, ema_var, gamma, beta, save_mean, save_var, scale_shift);
    if (pool_mode == 0) {
        const int slab = 64;
        hipLaunchKernelGGL(bn_apply_kernel<0>, dim3(cb, ceil_div(M, slab)), dim3(256), 0, s, M, C, y, ldy,
                           scale_shift, relu, out, ldo, slab, nullptr, nullptr, nullptr, nullptr, nullptr);
    } else {
        CLOUDAAE_REQUIRE(M / pool_rows <= 65535, name, "too many pooling groups");
        if (pool_mode == 1 && relu && training && pool_stats != nullptr && out == nullptr && C % 64 == 0 &&
            pool_rows % (4 * BN_U) == 0)
            hipLaunchKernelGGL(bn_apply_meanpool_kernel, dim3(cb, M / pool_rows), dim3(256), 0, s, C, y, ldy,
                               scale_shift, pool_rows, pooled, save_mean, save_var, pool_stats);
        else if (pool_mode == 1)
            hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(cb, M / pool_rows), dim3(256), 0, s, M, C, y, ldy,
                               scale_shift, relu, out, ldo, pool_rows, pooled, tie_count, save_mean, save_var,
                               (relu && training) ? pool_stats : nullptr);
        else
            hipLaunchKernelGGL(bn_apply_kernel<2>, dim3(cb, M / pool_rows), dim3(256), 0, s, M, C, y, ldy,
                               scale_shift, relu, out, ldo, pool_rows, pooled, tie_count, nullptr, nullptr, nullptr);
    }
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_bn_forward(int M, int C, const float *y, int ldy, const float *gamma,
                                     const float *beta, int training, const float *decay,
                                     float *ema_mean, float *ema_var, float *save_mean, float *save_var,
                                     int relu, float *out, int ldo, int pool_rows, int pool_mode,
                                     float *pooled, float *tie_count, double *pool_stats, void *workspace,
                                     cloudaae_stream_t stream)
{
    return bn_forward_impl("cloudaae_bn_forward", M, C, y, ldy, gamma, beta, training, decay, ema_mean, ema_var,
                           save_mean, save_var, relu, out, ldo, pool_rows, pool_mode, pooled, tie_count, pool_stats,
                           workspace, nullptr, 0, nullptr, stream);
}

CLOUDAAE_API int cloudaae_bn_forward_colstats(int M, int C, const float *y, int ldy, const float *gamma,
                                              const float *beta, int training, const float *decay,
                                              float *ema_mean, float *ema_var, float *save_mean, float *save_var,
                                              int relu, float *out, int ldo, int pool_rows, int pool_mode,
                                              float *pooled, float *tie_count, double *pool_stats, void *workspace,
                                              const double *colstats, int colstats_parts, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_bn_forward_colstats";
    CLOUDAAE_REQUIRE(colstats != nullptr && colstats_parts > 0, name, "no column sums given");
    CLOUDAAE_REQUIRE(M > BN_SMALL_M || pool_mode != 0, name, "small batches take cloudaae_bn_forward");
    return bn_forward_impl(name, M, C, y, ldy, gamma, beta, training, decay, ema_mean, ema_var, save_mean, save_var,
                           relu, out, ldo, pool_rows, pool_mode, pooled, tie_count, pool_stats, workspace, colstats,
                           colstats_parts, nullptr, stream);
}

CLOUDAAE_API int cloudaae_bn_forward_sync(int M, int C, const float *y, int ldy, const float *gamma,
                                          const float *beta, int training, const float *decay, float *ema_mean,
                                          float *ema_var, float *save_mean, float *save_var, int relu, float *out,
                                          int ldo, int pool_rows, int pool_mode, float *pooled, float *tie_count,
                                          double *pool_stats, void *workspace, const double *colstats,
                                          int colstats_parts, const cloudaae_bn_sync *sync, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_bn_forward_sync";
    CLOUDAAE_REQUIRE(colstats == nullptr || colstats_parts > 0, name, "column sums given without their part count");
    return bn_forward_impl(name, M, C, y, ldy, gamma, beta, training, decay, ema_mean, ema_var, save_mean, save_var,
                           relu, out, ldo, pool_rows, pool_mode, pooled, tie_count, pool_stats, workspace, colstats,
                           colstats_parts, sync, stream);
}

static int bn_backward_impl(const char *name, int M, int C, const float *y, int ldy, const float *gamma,
                            const float *beta, const float *save_mean, const float *save_var,
                            int training, int relu, const float *dout, int lddo, int pool_rows,
                            int pool_mode, const float *dpooled, const float *pooled,
                            const float *tie_count, float *dy, int lddy, float *dgamma,
                            float *dbeta, float *dbias, int accumulate_param_grads,
                            const double *pool_stats, void *workspace, const cloudaae_bn_sync *sync,
                            cloudaae_stream_t stream)
{
    CLOUDAAE_REQUIRE(M > 0 && C > 0 && ldy >= C, name, "bad size");
    CLOUDAAE_REQUIRE(workspace && gamma && beta && save_mean && save_var && dy, name, "null argument");
    CLOUDAAE_REQUIRE(pool_mode >= 0 && pool_mode <= 2, name, "bad pool_mode");
    CLOUDAAE_REQUIRE(pool_mode == 0 || (pool_rows > 0 && M % pool_rows == 0 && dpooled), name,
                     "pooled gradient needs pool_rows | M");
    CLOUDAAE_REQUIRE(pool_mode != 2 || (pooled && tie_count), name, "max pool backward needs max and tie count");
    CLOUDAAE_REQUIRE(dout != nullptr || pool_mode != 0, name, "no upstream gradient");
    hipStream_t s = (hipStream_t)stream;
    if (!(training && sync != nullptr))
        sync = nullptr;                        // inference-mode statistics do not depend on the batch
    if (pool_mode == 0 && M <= BN_SMALL_M && sync == nullptr) {
        hipLaunchKernelGGL(bn_small_bwd_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, s, M, C, y, ldy, gamma, beta,
                           save_mean, save_var, training, relu, dout, lddo, dy, lddy, dgamma, dbeta, dbias,
                           accumulate_param_grads);
        CLOUDAAE_CHECK_LAUNCH(name);
        return 0;
    }
    double *partial = (double *)workspace;
    float *scratch = (float *)(partial + (size_t)BN_MAX_PARTS * 4 * C);
    float *m12 = scratch + 2 * (size_t)C;
    BnBwdArgs a;
    a.want_xhat_sum = dbias != nullptr;
    a.M = M; a.C = C; a.ldy = ldy; a.lddo = lddo; a.rows = pool_rows > 0 ? pool_rows : 1;
    a.rows_shift = (a.rows & (a.rows - 1)) == 0 ? __builtin_ctz((unsigned)a.rows) : -1;
    a.pool = pool_mode; a.relu = relu; a.training = training;
    a.y = y; a.dout = dout; a.dpooled = dpooled; a.pooled = pooled; a.ties = tie_count;
    a.gamma = gamma; a.beta = beta; a.save_mean = save_mean; a.save_var = save_var;
    int parts = bn_parts(M);
    const int cb = ceil_div(C, 64);
    const int groups = pool_mode == 1 ? M / pool_rows : 0;
    bool finalised = false;
    if (pool_stats != nullptr && pool_mode == 1 && dout == nullptr && relu && training) {
        parts = groups < BN_MAX_PARTS ? groups : BN_MAX_PARTS;     // partial-sum rows from what the forward pass counted
        if (sync == nullptr) {
            // no exchange between the rows and their sums: the finalise forms the rows itself (one launch, same bits)
            hipLaunchKernelGGL(bn_bwd_finalize_pool_kernel, dim3(ceil_div(C, BN_FIN_CH)), dim3(BN_FIN_THREADS), 0, s, C, groups,
                               pool_rows, parts, dpooled, pool_stats, (double)M, training, dgamma, dbeta,
                               accumulate_param_grads, m12, dbias, gamma, save_var);
            finalised = true;
        } else {
            hipLaunchKernelGGL(bn_bwd_pool_partials_kernel, dim3(ceil_div(C, 256), parts), dim3(256), 0, s, C, groups,
                               pool_rows, dpooled, pool_stats, partial);
        }
    } else {
        hipLaunchKernelGGL(bn_bwd_colsum_kernel, dim3(cb, parts), dim3(256), 0, s, a, partial, parts);
    }
    const double *gsums = nullptr;
    double gcount = 0.0;
    if (sync != nullptr) {      // SyncBN: the means of dz and dz * x_hat are over the global batch
        if (int rc = bn_sync_exchange(name, sync, C, partial, parts, s))
            return rc;
        gsums = sync->buf;
        gcount = (double)M * (double)sync->world;
    }
    if (!finalised)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, BN_FIN_CH)), dim3(BN_FIN_THREADS), 0, s, C, partial, parts,
                           (double)M, training, dgamma, dbeta, accumulate_param_grads, m12, dbias, gamma, save_var, gsums,
                           gcount);
    const int slab = 64;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(cb, ceil_div(M, slab)), dim3(256), 0, s, a, m12, dy, lddy,
                       slab);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_bn_backward(int M, int C, const float *y, int ldy, const float *gamma,
                                      const float *beta, const float *save_mean, const float *save_var,
                                      int training, int relu, const float *dout, int lddo, int pool_rows,
                                      int pool_mode, const float *dpooled, const float *pooled,
                                      const float *tie_count, float *dy, int lddy, float *dgamma,
                                      float *dbeta, float *dbias, int accumulate_param_grads,
                                      const double *pool_stats, void *workspace, cloudaae_stream_t stream)
{
    return bn_backward_impl("cloudaae_bn_backward", M, C, y, ldy, gamma, beta, save_mean, save_var, training, relu, dout,
                            lddo, pool_rows, pool_mode, dpooled, pooled, tie_count, dy, lddy, dgamma, dbeta, dbias,
                            accumulate_param_grads, pool_stats, workspace, nullptr, stream);
}

CLOUDAAE_API int cloudaae_bn_backward_sync(int M, int C, const float *y, int ldy, const float *gamma,
                                           const float *beta, const float *save_mean, const float *save_var,
                                           int training, int relu, const float *dout, int lddo, int pool_rows,
                                           int pool_mode, const float *dpooled, const float *pooled,
                                           const float *tie_count, float *dy, int lddy, float *dgamma,
                                           float *dbeta, float *dbias, int accumulate_param_grads,
                                           const double *pool_stats, void *workspace, const cloudaae_bn_sync *sync,
                                           cloudaae_stream_t stream)
{
    return bn_backward_impl("cloudaae_bn_backward_sync", M, C, y, ldy, gamma, beta, save_mean, save_var, training, relu,
                            dout, lddo, pool_rows, pool_mode, dpooled, pooled, tie_count, dy, lddy, dgamma, dbeta, dbias,
                            accumulate_param_grads, pool_stats, workspace, sync, stream);
}

CLOUDAAE_API int cloudaae_colsum_f32(int M, int C, const float *x, int ldx, float *out, int accumulate,
                                     void *workspace, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_colsum_f32";
    CLOUDAAE_REQUIRE(M > 0 && C > 0 && ldx >= C && workspace && out, name, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    double *partial = (double *)workspace;
    const int parts = bn_parts(M);
    hipLaunchKernelGGL(bn_colsum_kernel, dim3(ceil_div(C, 64), parts), dim3(256), 0, s, M, C, x, ldx, partial,
                       parts);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(ceil_div(C, BN_FIN_CH)), dim3(BN_FIN_THREADS), 0, s, C, partial, parts, out,
                       accumulate);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}
