// step.hip -- the small kernels around the network in one training step (gfx950):
// input assembly, loss terms, scalar bookkeeping and the optimiser.  All of them are
// HBM-streaming or tiny; device-side scalars (step counter, beta powers, bn decay,
// upstream loss gradients) keep the whole step free of host synchronisation, so it
// can be captured in a hipGraph.
//
// Reference sites: train_cloudAAE_ycbv.py:194-273 (graph assembly), losses/*.py.
#include "common.h"
#include "philox.h"
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

// ---- input assembly: train_cloudAAE_ycbv.py:206-226 ---------------------------------
// one workgroup per cloud: v = visible[:N] + noise; mean over N; pc = [v - mean, onehot]
// noise: the caller's [B,N,3] array, or (noise == NULL, noise_std > 0) drawn HERE: the tf.random.normal(stddev) of
// :217 as a pure function of (seed, step counter, cloud, point) -- Philox4x32-10 + Box-Muller -- so a recorded step
// draws fresh noise at every replay without a generator kernel in front of it.
constexpr int IA_THREADS = 1024, IA_KEEP = 4;      // a thread keeps its first IA_KEEP points in registers
constexpr int IA_LDS_N = 4096;                     // clouds up to this many points assemble their rows through LDS
__global__ __launch_bounds__(IA_THREADS) void input_assemble_kernel(int P, int N, int num_class,
                                                                   const float *__restrict__ visible,
                                                                   const float *__restrict__ noise,
                                                                   const long long *__restrict__ class_id,
                                                                   float *__restrict__ pc, float *__restrict__ mean,
                                                                   float *__restrict__ noisy, float noise_std,
                                                                   unsigned long long seed, unsigned long long *__restrict__ draws)
{
    constexpr int NWV = IA_THREADS / 64;
    __shared__ float red[3][NWV];
    __shared__ float mu[3];
    __shared__ float sxyz[3][IA_LDS_N];
    const int b = blockIdx.x, t = threadIdx.x;
    const float *V = visible + (size_t)b * P * 3;
    const float *Z = noise ? noise + (size_t)b * N * 3 : nullptr;
    const bool draw = Z == nullptr && noise_std > 0.0f;
    // draws[0]: how many times this kernel has drawn (the Philox stream of THIS launch; its own counter, not the float
    // global-step variable: that one stops changing at 2^24 and follows a restored checkpoint); draws[1]: arrival ticket.
    // Every workgroup reads the counter when it starts and the LAST one to finish advances it -- by then all have read.
    const unsigned long long draw_no = draw && draws != nullptr ? draws[0] : 0ull;
    const unsigned stream_id = (unsigned)draw_no ^ (unsigned)(draw_no >> 32) * 0x9E3779B9u;
    auto point = [&](int j, float &x, float &y, float &z) {
        x = V[3 * j];
        y = V[3 * j + 1];
        z = V[3 * j + 2];
        if (Z) {
            x = x + Z[3 * j];
            y = y + Z[3 * j + 1];
            z = z + Z[3 * j + 2];
        } else if (draw) {
            unsigned r[4];
            philox4x32(seed, ((unsigned long long)(unsigned)b << 32) | (unsigned)j, stream_id, r);
            float n0, n1, n2, n3;
            normal2(r[0], r[1], n0, n1);
            normal2(r[2], r[3], n2, n3);
            x = x + n0 * noise_std;
            y = y + n1 * noise_std;
            z = z + n2 * noise_std;
        }
    };
    // the sums run over j = t, t + 1024, ... per thread, then lanes, then waves in index order (fixed order)
    float kx[IA_KEEP], ky[IA_KEEP], kz[IA_KEEP];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    int it = 0;
    for (int j = t; j < N; j += IA_THREADS, ++it) {
        float x, y, z;
        point(j, x, y, z);
#pragma unroll
        for (int u = 0; u < IA_KEEP; ++u)
            if (u == it) {
                kx[u] = x;
                ky[u] = y;
                kz[u] = z;
            }
        sx += x;
        sy += y;
        sz += z;
    }
    sx = wave_sum(sx);
    sy = wave_sum(sy);
    sz = wave_sum(sz);
    if ((t & 63) == 0) {
        red[0][t >> 6] = sx;
        red[1][t >> 6] = sy;
        red[2][t >> 6] = sz;
    }
    __syncthreads();
    if (t < 3) {
        float s = 0.0f;
        for (int w = 0; w < NWV; ++w)
            s += red[t][w];
        mu[t] = s / (float)N;
        mean[(size_t)b * 3 + t] = mu[t];
    }
    __syncthreads();
    const int C = 3 + num_class;
    const long long cls = class_id ? class_id[b] : -1;
    if (N <= IA_LDS_N) {
        // rows of C = 3 + num_class floats written by their point's thread are 96-byte strided stores, 24 per thread,
        // each touching 48 cache lines (9 of this kernel's 12 us); the centred coordinates go through LDS instead and the
        // cloud's [N, C] block leaves as whole 16-byte pieces in address order
        it = 0;
        for (int j = t; j < N; j += IA_THREADS, ++it) {
            float x = 0.f, y = 0.f, z = 0.f;
            if (it < IA_KEEP) {
#pragma unroll
                for (int u = 0; u < IA_KEEP; ++u)
                    if (u == it) {
                        x = kx[u];
                        y = ky[u];
                        z = kz[u];
                    }
            } else {
                point(j, x, y, z);
            }
            if (noisy) {
                noisy[((size_t)b * N + j) * 3 + 0] = x;
                noisy[((size_t)b * N + j) * 3 + 1] = y;
                noisy[((size_t)b * N + j) * 3 + 2] = z;
            }
            sxyz[0][j] = x - mu[0];
            sxyz[1][j] = y - mu[1];
            sxyz[2][j] = z - mu[2];
        }
        __syncthreads();
        float *rows = pc + (size_t)b * N * C;
        const int total = N * C;
        auto value = [&](int j, int c) { return c < 3 ? sxyz[c][j] : ((long long)(c - 3) == cls ? 1.0f : 0.0f); };
        if ((C & 3) == 0 && ((uintptr_t)rows & 15) == 0) {
            for (int e = 4 * t; e < total; e += 4 * IA_THREADS) {      // (C % 4 == 0: the four lie in one row)
                const int j = e / C, c = e - j * C;
                float4v v;
                v.x = value(j, c);
                v.y = value(j, c + 1);
                v.z = value(j, c + 2);
                v.w = value(j, c + 3);
                *reinterpret_cast<float4v *>(rows + e) = v;
            }
        } else {
            for (int e = t; e < total; e += IA_THREADS) {
                const int j = e / C;
                rows[e] = value(j, e - j * C);
            }
        }
    } else {
    it = 0;
    for (int j = t; j < N; j += IA_THREADS, ++it) {
        float x = 0.f, y = 0.f, z = 0.f;
        if (it < IA_KEEP) {
#pragma unroll
            for (int u = 0; u < IA_KEEP; ++u)
                if (u == it) {
                    x = kx[u];
                    y = ky[u];
                    z = kz[u];
                }
        } else {
            point(j, x, y, z);      // (the same values again: the draw is a function of its indices)
        }
        if (noisy) {
            noisy[((size_t)b * N + j) * 3 + 0] = x;
            noisy[((size_t)b * N + j) * 3 + 1] = y;
            noisy[((size_t)b * N + j) * 3 + 2] = z;
        }
        float *row = pc + ((size_t)b * N + j) * C;
        row[0] = x - mu[0];
        row[1] = y - mu[1];
        row[2] = z - mu[2];
        for (int c = 0; c < num_class; ++c)
            row[3 + c] = (c == cls) ? 1.0f : 0.0f;
    }
    }
    if (draw && draws != nullptr && t == 0) {
        if (atomicAdd(&draws[1], 1ull) == (unsigned long long)gridDim.x - 1ull) {
            draws[1] = 0ull;
            draws[0] = draw_no + 1ull;
        }
    }
}

// several buffers cleared by ONE launch (the zero zones of a recorded step + the gradient slots its split-K
// products add into): a launch per buffer costs ~5 us of stream time whatever it moves
constexpr int ZERO_SEGMENTS = 8;
struct ZeroSegments {
    int count;
    float4v *p[ZERO_SEGMENTS];
    long long n4[ZERO_SEGMENTS];        // 16-byte units
};
__global__ __launch_bounds__(256) void zero_segments_kernel(ZeroSegments z)
{
    const float4v zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s = 0; s < z.count; ++s)
        for (long long i = blockIdx.x * 256LL + threadIdx.x; i < z.n4[s]; i += 256LL * gridDim.x)
            z.p[s][i] = zero;
}

// out[b,r,:] = x[b,r,:] + v[b,:]   (train_cloudAAE_ycbv.py:232-233)
__global__ void add_rowvec_kernel(long long total, int R, int D, const float *__restrict__ x,
                                  const float *__restrict__ v, float *__restrict__ out)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / ((long long)R * D);
        const int d = (int)(i % D);
        out[i] = x[i] + v[b * D + d];
    }
}

__global__ void add_kernel(long long n, const float *__restrict__ a, const float *__restrict__ b,
                           float *__restrict__ out)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}

// out = a + b*c (a may be NULL): VAE reparameterisation z_mean + z_std * eps
// (models/pointnet_ycb_23_decoder_4.py:953) and its gradient d(z_std) = g * eps
__global__ void mul_add_kernel(long long n, const float *__restrict__ a, const float *__restrict__ b,
                               const float *__restrict__ c, float *__restrict__ out)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float p = b[i] * c[i];
        out[i] = a ? a[i] + p : p;
    }
}

// ---- unfused graph pieces kept for API completeness -----------------------------------
// get_edge_feature (utils/tf_util.py:635-669): out[b,i,j,:] = [x_i, x_nbr(i,j) - x_i];
// wo_center variant (:672-706): only the second half.  One wave per (point, neighbour).
__global__ __launch_bounds__(256) void edge_feature_kernel(int P, int N, int k, int C, int with_center,
                                                          const float *__restrict__ x, int ldx,
                                                          const int *__restrict__ nn_idx, float *__restrict__ out)
{
    const int W = with_center ? 2 * C : C;
    const long long total = (long long)P * k * C;
    for (long long t = blockIdx.x * 256LL + threadIdx.x; t < total; t += 256LL * gridDim.x) {
        const int c = (int)(t % C);
        const long long e = t / C;                 // edge id = pt*k + j
        const int pt = (int)(e / k);
        const int nb = (pt / N) * N + nn_idx[e];
        const float ci = x[(size_t)pt * ldx + c];
        float *row = out + (size_t)e * W;
        if (with_center) {
            row[c] = ci;
            row[C + c] = x[(size_t)nb * ldx + c] - ci;
        } else {
            row[c] = x[(size_t)nb * ldx + c] - ci;
        }
    }
}
// gradient: dx_i += sum_j (g_center[i,j] - g_diff[i,j]); dx_nbr(i,j) += g_diff[i,j]  (dx zero-filled)
__global__ __launch_bounds__(256) void edge_feature_grad_kernel(int P, int N, int k, int C, int with_center,
                                                               const float *__restrict__ g,
                                                               const int *__restrict__ nn_idx, float *__restrict__ dx)
{
    const int W = with_center ? 2 * C : C;
    const long long total = (long long)P * k * C;
    for (long long t = blockIdx.x * 256LL + threadIdx.x; t < total; t += 256LL * gridDim.x) {
        const int c = (int)(t % C);
        const long long e = t / C;
        const int pt = (int)(e / k);
        const int nb = (pt / N) * N + nn_idx[e];
        const float *row = g + (size_t)e * W;
        const float gd = with_center ? row[C + c] : row[c];
        const float gc = with_center ? row[c] : 0.0f;
        atomicAdd(dx + (size_t)pt * C + c, gc - gd);
        atomicAdd(dx + (size_t)nb * C + c, gd);
    }
}

// tf.reduce_mean / tf.reduce_max over groups of R consecutive rows of x[G*R, C]
// (axis=-2 of [B,N,k,C], or axis=1 of [B,N,1,C]); max also returns the tie count.
__global__ __launch_bounds__(256) void pool_rows_kernel(int G, int R, int C, int mode, const float *__restrict__ x,
                                                       float *__restrict__ out, float *__restrict__ ties)
{
    const long long total = (long long)G * C;
    for (long long t = blockIdx.x * 256LL + threadIdx.x; t < total; t += 256LL * gridDim.x) {
        const int c = (int)(t % C);
        const long long gidx = t / C;
        const float *p = x + (size_t)gidx * R * C + c;
        if (mode == 1) {
            float s = 0.0f;
            for (int r = 0; r < R; ++r)
                s = s + p[(size_t)r * C];
            out[t] = s / (float)R;
        } else {
            float m = -__builtin_inff(), n = 0.0f;
            for (int r = 0; r < R; ++r) {
                const float v = p[(size_t)r * C];
                n = v > m ? 1.0f : (v == m ? n + 1.0f : n);
                m = fmaxf(m, v);
            }
            out[t] = m;
            if (ties)
                ties[t] = n;
        }
    }
}
__global__ __launch_bounds__(256) void pool_rows_grad_kernel(int G, int R, int C, int mode, const float *__restrict__ x,
                                                            const float *__restrict__ out, const float *__restrict__ ties,
                                                            const float *__restrict__ g, float *__restrict__ dx)
{
    const long long total = (long long)G * R * C;
    for (long long t = blockIdx.x * 256LL + threadIdx.x; t < total; t += 256LL * gridDim.x) {
        const int c = (int)(t % C);
        const long long gidx = t / ((long long)R * C);
        const size_t o = (size_t)gidx * C + c;
        if (mode == 1)
            dx[t] = g[o] / (float)R;
        else
            dx[t] = x[t] == out[o] ? g[o] / ties[o] : 0.0f;   // tf.reduce_max shares among equal maxima
    }
}

// out[i] = scalar[0] * scale  (+ add[i])   -- gradient of a mean, broadcast
__global__ void fill_scaled_kernel(long long n, const float *__restrict__ scalar, float scale,
                                   const float *__restrict__ add, float *__restrict__ out)
{
    const float g = scalar[0] * scale;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = add ? g + add[i] : g;
}

// two-level mean with fp64 partial sums (deterministic)
constexpr int MEAN_BLOCKS = 256;
__global__ __launch_bounds__(256) void mean_stage1_kernel(long long n, const float *__restrict__ x,
                                                         double *__restrict__ partial)
{
    __shared__ double red[4];
    double s = 0.0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x)
        s += (double)x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// per[i] = a[i] + b[i] and the first stage of its mean in one pass (losses/chamfer_loss.py:13-14:
// loss_per_sample = dists_forward + dists_backward; loss = reduce_mean(loss_per_sample))
__global__ __launch_bounds__(256) void add_mean_stage1_kernel(long long n, const float *__restrict__ a,
                                                             const float *__restrict__ b, float *__restrict__ per,
                                                             double *__restrict__ partial)
{
    __shared__ double red[4];
    double s = 0.0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) {
        const float v = a[i] + b[i];
        per[i] = v;
        s += (double)v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void mean_stage2_kernel(int parts, double n, const double *__restrict__ partial,
                                                         float *__restrict__ out)
{
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < parts; i += 256)
        s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        out[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / n);
}

// ---- translation error: losses/trans_distance.py:4-9 --------------------------------
__global__ void trans_error_kernel(int b, const float *__restrict__ pred, const float *__restrict__ label,
                                   float *__restrict__ per)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b)
        return;
    const float dx = label[3 * i] - pred[3 * i], dy = label[3 * i + 1] - pred[3 * i + 1],
                dz = label[3 * i + 2] - pred[3 * i + 2];
    per[i] = sqrtf(dx * dx + dy * dy + dz * dz);
}
// d per / d pred = -(label - pred) / per
__global__ void trans_error_grad_kernel(int b, const float *__restrict__ pred, const float *__restrict__ label,
                                        const float *__restrict__ per, const float *__restrict__ gper,
                                        float *__restrict__ dpred)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b)
        return;
    const float g = gper[i] / per[i];
#pragma unroll
    for (int a = 0; a < 3; ++a)
        dpred[3 * i + a] = -(g * (label[3 * i + a] - pred[3 * i + a]));
}

// ---- SO(3) geodesic error in float64: losses/angular_distance_taylor.py:30-116 ------
// forward-mode duals carry d/d(pred) through exactly the reference's op sequence, so
// the gradient is what TF's autodiff of that graph yields (selected tf.where branch,
// clip_by_value passing the gradient only inside the range).
struct Dual {
    double v, d[3];
};
__device__ __forceinline__ Dual dconst(double c) { return Dual{c, {0.0, 0.0, 0.0}}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
__device__ __forceinline__ Dual operator-(Dual a) { return Dual{-a.v, {-a.d[0], -a.d[1], -a.d[2]}}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b)
{
    return Dual{a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ __forceinline__ Dual operator/(Dual a, Dual b)
{
    const double q = a.v / b.v;
    return Dual{q, {(a.d[0] - q * b.d[0]) / b.v, (a.d[1] - q * b.d[1]) / b.v, (a.d[2] - q * b.d[2]) / b.v}};
}
__device__ __forceinline__ Dual operator/(Dual a, double c) { return Dual{a.v / c, {a.d[0] / c, a.d[1] / c, a.d[2] / c}}; }
__device__ __forceinline__ Dual operator*(double c, Dual a) { return Dual{c * a.v, {c * a.d[0], c * a.d[1], c * a.d[2]}}; }
__device__ __forceinline__ Dual dsqrt(Dual a)
{
    const double r = sqrt(a.v), k = 0.5 / r;
    return Dual{r, {k * a.d[0], k * a.d[1], k * a.d[2]}};
}
__device__ __forceinline__ Dual dsin(Dual a)
{
    const double c = cos(a.v);
    return Dual{sin(a.v), {c * a.d[0], c * a.d[1], c * a.d[2]}};
}
__device__ __forceinline__ Dual dcos(Dual a)
{
    const double s = -sin(a.v);
    return Dual{cos(a.v), {s * a.d[0], s * a.d[1], s * a.d[2]}};
}

// exponential_map, angular_distance_taylor.py:30-66 (EPS = 1e-2 on theta^2)
__device__ void exp_map(const Dual ax[3], Dual R[3][3])
{
    const Dual zero = dconst(0.0);
    Dual ss[3][3] = {{zero, -ax[2], ax[1]}, {ax[2], zero, -ax[0]}, {-ax[1], ax[0], zero}};
    const Dual tsq = (ax[0] * ax[0] + ax[1] * ax[1]) + ax[2] * ax[2];
    Dual t1, t2;
    if (tsq.v < 1e-2) {
        const Dual p4 = tsq * tsq, p6 = (tsq * tsq) * tsq, p8 = ((tsq * tsq) * tsq) * tsq;
        t1 = (((dconst(1.0) - (tsq / 6.0)) + (p4 / 120.0)) - (p6 / 5040.0)) + (p8 / 362880.0);
        t2 = (((dconst(0.5) - (tsq / 24.0)) + (p4 / 720.0)) - (p6 / 40320.0)) + (p8 / 3628800.0);
    } else {
        const Dual th = dsqrt(tsq);
        t1 = dsin(th) / th;
        t2 = (dconst(1.0) - dcos(th)) / tsq;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Dual sq = zero;
            for (int k = 0; k < 3; ++k)
                sq = sq + ss[i][k] * ss[k][j];
            R[i][j] = (dconst(i == j ? 1.0 : 0.0) + t1 * ss[i][j]) + t2 * sq;
        }
}

__global__ void rotation_error_kernel(int b, const float *__restrict__ pred, const double *__restrict__ label,
                                      double *__restrict__ per, double *__restrict__ jac)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b)
        return;
    Dual p[3], l[3];
    for (int a = 0; a < 3; ++a) {
        p[a] = dconst((double)pred[3 * i + a]);  // tf.cast(rot_pred, tf.float64), train...:249
        p[a].d[a] = 1.0;
        l[a] = dconst(label[3 * i + a]);
    }
    Dual Rp[3][3], Rl[3][3];
    exp_map(p, Rp);
    exp_map(l, Rl);
    // R = R_label * R_pred^T ; only its trace is needed (angular_distance_taylor.py:113,77-84)
    Dual tr = dconst(0.0);
    for (int r = 0; r < 3; ++r) {
        Dual e = dconst(0.0);
        for (int k = 0; k < 3; ++k)
            e = e + Rl[r][k] * Rp[r][k];
        tr = tr + e;
    }
    Dual t = (tr - dconst(1.0)) / 2.0;
    const double lim = 0.9999999;
    if (t.v < -lim)
        t = dconst(-lim);
    else if (t.v > lim)
        t = dconst(lim);
    const double theta = acos(t.v);
    const double k = -1.0 / sqrt(1.0 - t.v * t.v);
    per[i] = theta;
    if (jac) {
        jac[3 * i + 0] = k * t.d[0];
        jac[3 * i + 1] = k * t.d[1];
        jac[3 * i + 2] = k * t.d[2];
    }
}

// exponential_map alone (train_cloudAAE_ycbv.py:79-85: rotation matrix of the GT pose)
__global__ void exp_map_kernel(int b, const double *__restrict__ axag, double *__restrict__ R)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b)
        return;
    Dual a[3], M[3][3];
    for (int k = 0; k < 3; ++k)
        a[k] = dconst(axag[3 * i + k]);
    exp_map(a, M);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            R[9 * i + 3 * r + c] = M[r][c].v;
}

// loss = mean(per) (fp64 -> fp32, train...:253); dpred = gloss/b * jac (fp64 -> fp32)
__global__ void rotation_reduce_kernel(int b, const double *__restrict__ per, float *__restrict__ loss)
{
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < b; i += 256)
        s += per[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        loss[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)b);
}
__global__ void rotation_grad_kernel(int b, const double *__restrict__ jac, const float *__restrict__ gloss,
                                     float *__restrict__ dpred)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * b)
        return;
    dpred[i] = (float)((double)gloss[0] / (double)b * jac[i]);
}

// total = w0*a + w1*b + w2*c (train...:268) and its gradient fan-out
__global__ void loss_mix_kernel(const float *a, const float *b, const float *c, float w0, float w1, float w2,
                                float *out)
{
    out[0] = (w0 * a[0] + w1 * b[0]) + w2 * c[0];
}
__global__ void loss_mix_grad_kernel(const float *g, float w0, float w1, float w2, float *ga, float *gb, float *gc)
{
    ga[0] = g[0] * w0;
    gb[0] = g[0] * w1;
    gc[0] = g[0] * w2;
}

// ---- the loss tail of the step in one launch (train...:241-268) ----------------------------
// translation error per sample + mean, SO(3) error per sample (fp64) + Jacobian + mean, and the
// weighted total; the backward twin turns d(total) into the three upstream gradients.  Same
// arithmetic as the single-purpose kernels above (ten launches of ~4.5 us become two).
__device__ __forceinline__ void pose_losses_body(int b, const float *__restrict__ tpred,
                                                 const float *__restrict__ tlabel,
                                                 const float *__restrict__ rpred,
                                                 const double *__restrict__ rlabel, float xyz_loss, float w0, float w1,
                                                 float w2, float *__restrict__ tper,
                                                 float *__restrict__ tloss, double *__restrict__ rper,
                                                 double *__restrict__ rjac, float *__restrict__ rloss,
                                                 float *__restrict__ total, double (*red)[4])
{
    double st = 0.0, sr = 0.0;
    for (int i = threadIdx.x; i < b; i += 256) {
        const float dx = tlabel[3 * i] - tpred[3 * i], dy = tlabel[3 * i + 1] - tpred[3 * i + 1],
                    dz = tlabel[3 * i + 2] - tpred[3 * i + 2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        tper[i] = d;
        st += (double)d;
        Dual p[3], l[3];
        for (int a = 0; a < 3; ++a) {
            p[a] = dconst((double)rpred[3 * i + a]);
            p[a].d[a] = 1.0;
            l[a] = dconst(rlabel[3 * i + a]);
        }
        Dual Rp[3][3], Rl[3][3];
        exp_map(p, Rp);
        exp_map(l, Rl);
        Dual tr = dconst(0.0);
        for (int r = 0; r < 3; ++r) {
            Dual e = dconst(0.0);
            for (int k = 0; k < 3; ++k)
                e = e + Rl[r][k] * Rp[r][k];
            tr = tr + e;
        }
        Dual t = (tr - dconst(1.0)) / 2.0;
        const double lim = 0.9999999;
        if (t.v < -lim)
            t = dconst(-lim);
        else if (t.v > lim)
            t = dconst(lim);
        const double theta = acos(t.v);
        const double k = -1.0 / sqrt(1.0 - t.v * t.v);
        rper[i] = theta;
        rjac[3 * i + 0] = k * t.d[0];
        rjac[3 * i + 1] = k * t.d[1];
        rjac[3 * i + 2] = k * t.d[2];
        sr += theta;
    }
    st = wave_sum(st);
    sr = wave_sum(sr);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = st;
        red[1][threadIdx.x >> 6] = sr;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tl = (float)(((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (double)b);
        const float rl = (float)(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (double)b);
        tloss[0] = tl;
        rloss[0] = rl;
        total[0] = (w0 * xyz_loss + w1 * tl) + w2 * rl;
    }
}

__global__ __launch_bounds__(256) void pose_losses_kernel(int b, const float *__restrict__ tpred,
                                                         const float *__restrict__ tlabel,
                                                         const float *__restrict__ rpred,
                                                         const double *__restrict__ rlabel,
                                                         const float *__restrict__ xyz_loss, float w0, float w1,
                                                         float w2, float *__restrict__ tper,
                                                         float *__restrict__ tloss, double *__restrict__ rper,
                                                         double *__restrict__ rjac, float *__restrict__ rloss,
                                                         float *__restrict__ total)
{
    __shared__ double red[2][4];
    pose_losses_body(b, tpred, tlabel, rpred, rlabel, xyz_loss[0], w0, w1, w2, tper, tloss, rper, rjac, rloss, total, red);
}

// d(total)/d(xyz_loss, trans_pred, rot_pred) of rows i = threadIdx.x, +256, ... for the upstream gradient g
__device__ __forceinline__ void pose_losses_grad_rows(int b, int i0, int stride, const float *__restrict__ tpred,
                                                      const float *__restrict__ tlabel, const float *__restrict__ tper,
                                                      const double *__restrict__ rjac, float g, float w0, float w1,
                                                      float w2, float *__restrict__ dxyz, float *__restrict__ dtpred,
                                                      float *__restrict__ drpred)
{
    if (i0 == 0)
        dxyz[0] = g * w0;
    const float gt = g * w1, gr = g * w2;
    for (int i = i0; i < b; i += stride) {
        const float gp = (gt / (float)b) / tper[i];           // d mean / d per = 1/b, d per / d pred = -(l - p)/per
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            dtpred[3 * i + a] = -(gp * (tlabel[3 * i + a] - tpred[3 * i + a]));
            drpred[3 * i + a] = (float)((double)gr / (double)b * rjac[3 * i + a]);
        }
    }
}

// ---- the WHOLE loss tail of a training step in one launch (losses/chamfer_loss.py:12-14 + train...:241-268) ----
// per = dist1 + dist2 and its mean (256 workgroups, fp64 partial sums), and in the workgroup that finishes last
// (arrival counter; partial sums published with agent-scope stores and read back with agent-scope loads, summed in
// index order: deterministic) the pose losses, the weighted total and -- the step's upstream gradient d(total) is a
// constant known now -- the three gradients the backward pass starts from.  Four launches of ~5 us become one.
struct LossTail {
    int b;
    const float *tpred, *tlabel, *rpred;
    const double *rlabel;
    float w0, w1, w2;
    float *xyz_loss, *tper, *tloss;
    double *rper, *rjac;
    float *rloss, *total;
    const float *g_total;           // NULL: no gradients
    float *dxyz, *dtpred, *drpred;
    int *ticket;
};
__global__ __launch_bounds__(256) void loss_tail_kernel(long long n, const float *__restrict__ a,
                                                       const float *__restrict__ bb, float *__restrict__ per,
                                                       double *__restrict__ partial, LossTail t)
{
    __shared__ double red[2][4];
    __shared__ int last;
    double s = 0.0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) {
        const float v = a[i] + bb[i];
        per[i] = v;
        s += (double)v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0)
        red[0][threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&partial[blockIdx.x], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the sum has reached the memory side before the ticket
        const int tk = __hip_atomic_fetch_add(t.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = tk == (int)gridDim.x - 1;
        if (last)
            __hip_atomic_store(t.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last)
        return;
    double p = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256)
        p += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p = wave_sum(p);
    if ((threadIdx.x & 63) == 0)
        red[1][threadIdx.x >> 6] = p;
    __syncthreads();
    const float xyz = (float)(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (double)n);
    __syncthreads();
    if (threadIdx.x == 0)
        t.xyz_loss[0] = xyz;
    pose_losses_body(t.b, t.tpred, t.tlabel, t.rpred, t.rlabel, xyz, t.w0, t.w1, t.w2, t.tper, t.tloss, t.rper, t.rjac,
                     t.rloss, t.total, red);
    if (t.g_total != nullptr)       // (a thread reads back the rows it wrote itself)
        pose_losses_grad_rows(t.b, (int)threadIdx.x, 256, t.tpred, t.tlabel, t.tper, t.rjac, t.g_total[0], t.w0, t.w1,
                              t.w2, t.dxyz, t.dtpred, t.drpred);
}

__global__ void pose_losses_grad_kernel(int b, const float *__restrict__ tpred, const float *__restrict__ tlabel,
                                        const float *__restrict__ tper, const double *__restrict__ rjac,
                                        const float *__restrict__ g, float w0, float w1, float w2,
                                        float *__restrict__ dxyz, float *__restrict__ dtpred,
                                        float *__restrict__ drpred)
{
    pose_losses_grad_rows(b, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, tpred, tlabel, tper, rjac, g[0],
                          w0, w1, w2, dxyz, dtpred, drpred);
}

// ---- optimiser: tf.train.AdamOptimizer (train...:263-273), TF-1.x ApplyAdam form ----
//   lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
//   m += (g - m) * (1 - beta1);  v += (g*g - v) * (1 - beta2)
//   var -= lr_t * m / (sqrt(v) + eps)
// What the end of a training step does to its device scalars, done by the LAST workgroup of the optimiser
// kernel to finish (it is the last reader of the beta powers): beta powers advance, the step counter
// (`batch`, train...:192) goes up, and the batch-norm decay of the NEXT step is derived from it
// (train...:194-202).  Three one-thread launches (~14 us of timeline) otherwise.  ticket: one int holding
// zero, left zero.
struct AdamTail {
    int *ticket;        // NULL: none of this
    float *b1p, *b2p, *step, *bn_decay;
    float step_inc, batch_size, bn_init, bn_decay_step, bn_rate, bn_clip;
};

__global__ __launch_bounds__(256) void adam_tf_kernel(long long n, float *__restrict__ param,
                                                     const float *__restrict__ grad, float *__restrict__ m,
                                                     float *__restrict__ v, float lr, float beta1, float beta2,
                                                     float eps, const float *b1p, const float *b2p, float gscale,
                                                     AdamTail tail)
{
    const float lr_t = lr * sqrtf(1.0f - b2p[0]) / (1.0f - b1p[0]);
    const float om1 = 1.0f - beta1, om2 = 1.0f - beta2;
    const long long n4 = n / 4;
    float4v *P4 = reinterpret_cast<float4v *>(param);
    const float4v *G4 = reinterpret_cast<const float4v *>(grad);
    float4v *M4 = reinterpret_cast<float4v *>(m);
    float4v *V4 = reinterpret_cast<float4v *>(v);
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += 256LL * gridDim.x) {
        float4v p = P4[i], g = G4[i] * gscale, mm = M4[i], vv = V4[i];
        mm = mm + (g - mm) * om1;
        vv = vv + (g * g - vv) * om2;
        float4v den;
        den.x = sqrtf(vv.x) + eps;
        den.y = sqrtf(vv.y) + eps;
        den.z = sqrtf(vv.z) + eps;
        den.w = sqrtf(vv.w) + eps;
        p = p - (mm * lr_t) / den;
        P4[i] = p;
        M4[i] = mm;
        V4[i] = vv;
    }
    for (long long i = n4 * 4 + blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) {
        const float g = grad[i] * gscale;
        const float mm = m[i] + (g - m[i]) * om1;
        const float vv = v[i] + (g * g - v[i]) * om2;
        param[i] = param[i] - (mm * lr_t) / (sqrtf(vv) + eps);
        m[i] = mm;
        v[i] = vv;
    }
    if (tail.ticket != nullptr) {
        __syncthreads();    // every thread of this workgroup has read the beta powers
        if (threadIdx.x == 0) {
            const int t = __hip_atomic_fetch_add(tail.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (int)gridDim.x - 1) {
                __hip_atomic_store(tail.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tail.b1p[0] = tail.b1p[0] * beta1;
                tail.b2p[0] = tail.b2p[0] * beta2;
                const float step = tail.step[0] + tail.step_inc;
                tail.step[0] = step;
                if (tail.bn_decay != nullptr) {
                    const float p = floorf(step * tail.batch_size / tail.bn_decay_step);
                    tail.bn_decay[0] = fminf(tail.bn_clip, 1.0f - tail.bn_init * powf(tail.bn_rate, p));
                }
            }
        }
    }
}
__global__ void adam_advance_kernel(float *b1p, float *b2p, float beta1, float beta2)
{
    b1p[0] = b1p[0] * beta1;
    b2p[0] = b2p[0] * beta2;
}
// tf.train.GradientDescentOptimizer: var -= lr * g
__global__ __launch_bounds__(256) void sgd_kernel(long long n, float *__restrict__ param,
                                                 const float *__restrict__ grad, float lr, float gscale)
{
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x)
        param[i] = param[i] - lr * (grad[i] * gscale);
}

// bn_decay(step) = min(clip, 1 - init * rate^floor(step*batch/decay_step))   (train...:194-202)
__global__ void bn_decay_kernel(const float *step, float batch_size, float init, float decay_step, float rate,
                                float clip, float *out)
{
    const float p = floorf(step[0] * batch_size / decay_step);
    const float mom = init * powf(rate, p);
    out[0] = fminf(clip, 1.0f - mom);
}
__global__ void increment_kernel(float *x, float by) { x[0] = x[0] + by; }

static int stream_grid(long long n)
{
    long long g = (n + 255) / 256;
    if (g > 2048)
        g = 2048;
    if (g < 1)
        g = 1;
    return (int)g;
}

} // namespace cloudaae

using namespace cloudaae;

CLOUDAAE_API int cloudaae_input_assemble(int b, int p, int n, int num_class, const float *visible,
                                         const float *noise, const long long *class_id, float *pc,
                                         float *mean, float *noisy, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_input_assemble";
    CLOUDAAE_REQUIRE(b >= 0 && n > 0 && p >= n && num_class >= 0, name, "bad size (need n <= rows of visible)");
    if (b == 0)
        return 0;
    hipLaunchKernelGGL(input_assemble_kernel, dim3(b), dim3(IA_THREADS), 0, (hipStream_t)stream, p, n, num_class,
                       visible, noise, class_id, pc, mean, noisy, 0.0f, 0ull, (unsigned long long *)nullptr);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_input_assemble_noise(int b, int p, int n, int num_class, const float *visible,
                                               const long long *class_id, float *pc, float *mean, float *noisy,
                                               float noise_std, unsigned long long seed, unsigned long long *draws,
                                               cloudaae_stream_t stream)
{
    const char *name = "cloudaae_input_assemble_noise";
    CLOUDAAE_REQUIRE(b >= 0 && n > 0 && p >= n && num_class >= 0, name, "bad size (need n <= rows of visible)");
    CLOUDAAE_REQUIRE(noise_std >= 0.0f, name, "negative standard deviation");
    if (b == 0)
        return 0;
    hipLaunchKernelGGL(input_assemble_kernel, dim3(b), dim3(IA_THREADS), 0, (hipStream_t)stream, p, n, num_class,
                       visible, (const float *)nullptr, class_id, pc, mean, noisy, noise_std, seed, draws);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_zero_buffers(int count, void *const *buffers, const long long *bytes, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_zero_buffers";
    CLOUDAAE_REQUIRE(count >= 0 && (count == 0 || (buffers && bytes)), name, "null argument");
    for (int first = 0; first < count; first += ZERO_SEGMENTS) {
        ZeroSegments z = {};
        long long most = 0;
        for (int i = first; i < count && i < first + ZERO_SEGMENTS; ++i) {
            CLOUDAAE_REQUIRE(((uintptr_t)buffers[i] & 15) == 0 && bytes[i] >= 0 && bytes[i] % 16 == 0, name,
                             "buffers must be 16-byte aligned and a multiple of 16 bytes long");
            z.p[z.count] = (float4v *)buffers[i];
            z.n4[z.count] = bytes[i] / 16;
            most = z.n4[z.count] > most ? z.n4[z.count] : most;
            ++z.count;
        }
        if (most == 0)
            continue;
        hipLaunchKernelGGL(zero_segments_kernel, dim3(stream_grid(most)), dim3(256), 0, (hipStream_t)stream, z);
    }
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_add_rowvec(int b, int r, int d, const float *x, const float *v, float *out,
                                     cloudaae_stream_t stream)
{
    const long long total = (long long)b * r * d;
    if (total == 0)
        return 0;
    hipLaunchKernelGGL(add_rowvec_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, total, r,
                       d, x, v, out);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_add_rowvec");
    return 0;
}

CLOUDAAE_API int cloudaae_add_f32(long long n, const float *a, const float *b, float *out,
                                  cloudaae_stream_t stream)
{
    if (n == 0)
        return 0;
    hipLaunchKernelGGL(add_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, n, a, b, out);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_add_f32");
    return 0;
}

CLOUDAAE_API int cloudaae_mul_add_f32(long long n, const float *a, const float *b, const float *c, float *out,
                                      cloudaae_stream_t stream)
{
    if (n == 0)
        return 0;
    hipLaunchKernelGGL(mul_add_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, n, a, b, c, out);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_mul_add_f32");
    return 0;
}

CLOUDAAE_API int cloudaae_edge_feature(int b, int n, int k, int c, int with_center, const float *x, int ldx,
                                       const int *nn_idx, float *out, cloudaae_stream_t stream)
{
    const long long total = (long long)b * n * k * c;
    if (total == 0)
        return 0;
    hipLaunchKernelGGL(edge_feature_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, b * n, n,
                       k, c, with_center, x, ldx, nn_idx, out);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_edge_feature");
    return 0;
}

CLOUDAAE_API int cloudaae_edge_feature_grad(int b, int n, int k, int c, int with_center, const float *g,
                                            const int *nn_idx, float *dx, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_edge_feature_grad";
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)b * n * k * c;
    if ((size_t)b * n * c)
        CLOUDAAE_CHECK_HIP(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)b * n * c, s), name);
    if (total == 0)
        return 0;
    hipLaunchKernelGGL(edge_feature_grad_kernel, dim3(stream_grid(total)), dim3(256), 0, s, b * n, n, k, c,
                       with_center, g, nn_idx, dx);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_pool_rows(int groups, int rows, int c, int mode, const float *x, float *out, float *ties,
                                    cloudaae_stream_t stream)
{
    const char *name = "cloudaae_pool_rows";
    CLOUDAAE_REQUIRE(rows > 0 && (mode == 1 || mode == 2), name, "rows > 0 and mode 1 (mean) / 2 (max)");
    CLOUDAAE_REQUIRE(mode == 1 || ties != nullptr, name, "max pooling needs the tie_count output");
    const long long total = (long long)groups * c;
    if (total == 0)
        return 0;
    hipLaunchKernelGGL(pool_rows_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, groups, rows, c,
                       mode, x, out, ties);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_pool_rows_grad(int groups, int rows, int c, int mode, const float *x, const float *out,
                                         const float *ties, const float *g, float *dx, cloudaae_stream_t stream)
{
    const long long total = (long long)groups * rows * c;
    if (total == 0)
        return 0;
    hipLaunchKernelGGL(pool_rows_grad_kernel, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, groups,
                       rows, c, mode, x, out, ties, g, dx);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_pool_rows_grad");
    return 0;
}

CLOUDAAE_API int cloudaae_fill_scaled(long long n, const float *scalar, float scale, const float *add,
                                      float *out, cloudaae_stream_t stream)
{
    if (n == 0)
        return 0;
    hipLaunchKernelGGL(fill_scaled_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, n, scalar,
                       scale, add, out);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_fill_scaled");
    return 0;
}

CLOUDAAE_API long long cloudaae_mean_workspace_bytes(void) { return (long long)MEAN_BLOCKS * sizeof(double); }

CLOUDAAE_API int cloudaae_mean_f32(long long n, const float *x, float *out, void *workspace,
                                   cloudaae_stream_t stream)
{
    const char *name = "cloudaae_mean_f32";
    CLOUDAAE_REQUIRE(n > 0 && workspace, name, "empty input or no workspace");
    hipStream_t s = (hipStream_t)stream;
    int parts = stream_grid(n);
    if (parts > MEAN_BLOCKS)
        parts = MEAN_BLOCKS;
    hipLaunchKernelGGL(mean_stage1_kernel, dim3(parts), dim3(256), 0, s, n, x, (double *)workspace);
    hipLaunchKernelGGL(mean_stage2_kernel, dim3(1), dim3(256), 0, s, parts, (double)n, (const double *)workspace,
                       out);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_add_mean_f32(long long n, const float *a, const float *b, float *per, float *out,
                                       void *workspace, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_add_mean_f32";
    CLOUDAAE_REQUIRE(n > 0 && a && b && per && out && workspace, name, "empty input or null argument");
    hipStream_t s = (hipStream_t)stream;
    int parts = stream_grid(n);
    if (parts > MEAN_BLOCKS)
        parts = MEAN_BLOCKS;
    hipLaunchKernelGGL(add_mean_stage1_kernel, dim3(parts), dim3(256), 0, s, n, a, b, per, (double *)workspace);
    hipLaunchKernelGGL(mean_stage2_kernel, dim3(1), dim3(256), 0, s, parts, (double)n, (const double *)workspace,
                       out);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_trans_error(int b, const float *pred, const float *label, float *per,
                                      cloudaae_stream_t stream)
{
    if (b == 0)
        return 0;
    hipLaunchKernelGGL(trans_error_kernel, dim3(ceil_div(b, 256)), dim3(256), 0, (hipStream_t)stream, b, pred,
                       label, per);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_trans_error");
    return 0;
}

CLOUDAAE_API int cloudaae_trans_error_grad(int b, const float *pred, const float *label, const float *per,
                                           const float *gper, float *dpred, cloudaae_stream_t stream)
{
    if (b == 0)
        return 0;
    hipLaunchKernelGGL(trans_error_grad_kernel, dim3(ceil_div(b, 256)), dim3(256), 0, (hipStream_t)stream, b,
                       pred, label, per, gper, dpred);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_trans_error_grad");
    return 0;
}

CLOUDAAE_API int cloudaae_rotation_error(int b, const float *pred, const double *label, double *per,
                                         double *jac, float *loss, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_rotation_error";
    CLOUDAAE_REQUIRE(b > 0, name, "empty batch");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(rotation_error_kernel, dim3(ceil_div(b, 64)), dim3(64), 0, s, b, pred, label, per, jac);
    if (loss)
        hipLaunchKernelGGL(rotation_reduce_kernel, dim3(1), dim3(256), 0, s, b, per, loss);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_exponential_map(int b, const double *axag, double *rot, cloudaae_stream_t stream)
{
    if (b == 0)
        return 0;
    hipLaunchKernelGGL(exp_map_kernel, dim3(ceil_div(b, 64)), dim3(64), 0, (hipStream_t)stream, b, axag, rot);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_exponential_map");
    return 0;
}

CLOUDAAE_API int cloudaae_rotation_error_grad(int b, const double *jac, const float *gloss, float *dpred,
                                              cloudaae_stream_t stream)
{
    if (b == 0)
        return 0;
    hipLaunchKernelGGL(rotation_grad_kernel, dim3(ceil_div(3 * b, 256)), dim3(256), 0, (hipStream_t)stream, b,
                       jac, gloss, dpred);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_rotation_error_grad");
    return 0;
}

CLOUDAAE_API int cloudaae_loss_mix(const float *a, const float *b, const float *c, float w0, float w1, float w2,
                                   float *out, cloudaae_stream_t stream)
{
    hipLaunchKernelGGL(loss_mix_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, b, c, w0, w1, w2, out);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_loss_mix");
    return 0;
}

CLOUDAAE_API int cloudaae_loss_mix_grad(const float *g, float w0, float w1, float w2, float *ga, float *gb,
                                        float *gc, cloudaae_stream_t stream)
{
    hipLaunchKernelGGL(loss_mix_grad_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, g, w0, w1, w2, ga, gb, gc);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_loss_mix_grad");
    return 0;
}

CLOUDAAE_API int cloudaae_adam_tf(long long n, float *param, const float *grad, float *m, float *v, float lr,
                                  float beta1, float beta2, float eps, float *beta1_power, float *beta2_power,
                                  float grad_scale, int advance, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_adam_tf";
    CLOUDAAE_REQUIRE(n >= 0 && beta1_power && beta2_power, name, "bad argument");
    CLOUDAAE_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0, name,
                     "buffers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    AdamTail none = {};
    if (n > 0)
        hipLaunchKernelGGL(adam_tf_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, s, n, param, grad, m, v,
                           lr, beta1, beta2, eps, beta1_power, beta2_power, grad_scale, none);
    if (advance)
        hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, s, beta1_power, beta2_power, beta1, beta2);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_adam_tf_step(long long n, float *param, const float *grad, float *m, float *v, float lr,
                                       float beta1, float beta2, float eps, float *beta1_power, float *beta2_power,
                                       float grad_scale, float *step, float step_inc, float batch_size,
                                       float bn_init, float bn_decay_step, float bn_rate, float bn_clip,
                                       float *bn_decay, int *ticket, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_adam_tf_step";
    CLOUDAAE_REQUIRE(n > 0 && beta1_power && beta2_power && step && ticket, name, "bad argument");
    CLOUDAAE_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0, name,
                     "buffers must be 16-byte aligned");
    AdamTail tail;
    tail.ticket = ticket; tail.b1p = beta1_power; tail.b2p = beta2_power; tail.step = step; tail.bn_decay = bn_decay;
    tail.step_inc = step_inc; tail.batch_size = batch_size; tail.bn_init = bn_init; tail.bn_decay_step = bn_decay_step;
    tail.bn_rate = bn_rate; tail.bn_clip = bn_clip;
    hipLaunchKernelGGL(adam_tf_kernel, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, n, param,
                       grad, m, v, lr, beta1, beta2, eps, beta1_power, beta2_power, grad_scale, tail);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_sgd(long long n, float *param, const float *grad, float lr, float grad_scale,
                              cloudaae_stream_t stream)
{
    if (n == 0)
        return 0;
    hipLaunchKernelGGL(sgd_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, n, param, grad, lr,
                       grad_scale);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_sgd");
    return 0;
}

CLOUDAAE_API int cloudaae_bn_decay_schedule(const float *step, float batch_size, float init, float decay_step,
                                            float rate, float clip, float *out, cloudaae_stream_t stream)
{
    hipLaunchKernelGGL(bn_decay_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, batch_size, init,
                       decay_step, rate, clip, out);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_bn_decay_schedule");
    return 0;
}

CLOUDAAE_API int cloudaae_increment(float *x, float by, cloudaae_stream_t stream)
{
    hipLaunchKernelGGL(increment_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, x, by);
    CLOUDAAE_CHECK_LAUNCH("cloudaae_increment");
    return 0;
}

CLOUDAAE_API int cloudaae_pose_losses(int b, const float *trans_pred, const float *trans_label,
                                      const float *rot_pred, const double *rot_label, const float *xyz_loss,
                                      float w_xyz, float w_trans, float w_rot, float *trans_per,
                                      float *trans_loss, double *rot_per, double *rot_jac, float *rot_loss,
                                      float *total, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_pose_losses";
    CLOUDAAE_REQUIRE(b > 0, name, "empty batch");
    CLOUDAAE_REQUIRE(trans_pred && trans_label && rot_pred && rot_label && xyz_loss && trans_per && trans_loss &&
                         rot_per && rot_jac && rot_loss && total, name, "null argument");
    hipLaunchKernelGGL(pose_losses_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, b, trans_pred, trans_label,
                       rot_pred, rot_label, xyz_loss, w_xyz, w_trans, w_rot, trans_per, trans_loss, rot_per, rot_jac,
                       rot_loss, total);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API long long cloudaae_loss_tail_workspace_bytes(void) { return (long long)MEAN_BLOCKS * sizeof(double); }

CLOUDAAE_API int cloudaae_loss_tail(long long n, const float *dist1, const float *dist2, float *per, float *xyz_loss,
                                    int b, const float *trans_pred, const float *trans_label, const float *rot_pred,
                                    const double *rot_label, float w_xyz, float w_trans, float w_rot, float *trans_per,
                                    float *trans_loss, double *rot_per, double *rot_jac, float *rot_loss, float *total,
                                    const float *g_total, float *d_xyz_loss, float *d_trans_pred, float *d_rot_pred,
                                    void *workspace, int *ticket, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_loss_tail";
    CLOUDAAE_REQUIRE(n > 0 && b > 0, name, "empty input");
    CLOUDAAE_REQUIRE(dist1 && dist2 && per && xyz_loss && trans_pred && trans_label && rot_pred && rot_label && trans_per &&
                         trans_loss && rot_per && rot_jac && rot_loss && total && workspace && ticket, name,
                     "null argument");
    CLOUDAAE_REQUIRE(g_total == nullptr || (d_xyz_loss && d_trans_pred && d_rot_pred), name, "gradient outputs missing");
    int parts = stream_grid(n);
    if (parts > MEAN_BLOCKS)
        parts = MEAN_BLOCKS;
    LossTail t;
    t.b = b; t.tpred = trans_pred; t.tlabel = trans_label; t.rpred = rot_pred; t.rlabel = rot_label;
    t.w0 = w_xyz; t.w1 = w_trans; t.w2 = w_rot; t.xyz_loss = xyz_loss; t.tper = trans_per; t.tloss = trans_loss;
    t.rper = rot_per; t.rjac = rot_jac; t.rloss = rot_loss; t.total = total; t.g_total = g_total;
    t.dxyz = d_xyz_loss; t.dtpred = d_trans_pred; t.drpred = d_rot_pred; t.ticket = ticket;
    hipLaunchKernelGGL(loss_tail_kernel, dim3(parts), dim3(256), 0, (hipStream_t)stream, n, dist1, dist2, per,
                       (double *)workspace, t);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_pose_losses_grad(int b, const float *trans_pred, const float *trans_label,
                                           const float *trans_per, const double *rot_jac, const float *g_total,
                                           float w_xyz, float w_trans, float w_rot, float *d_xyz_loss,
                                           float *d_trans_pred, float *d_rot_pred, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_pose_losses_grad";
    CLOUDAAE_REQUIRE(b > 0, name, "empty batch");
    CLOUDAAE_REQUIRE(trans_pred && trans_label && trans_per && rot_jac && g_total && d_xyz_loss && d_trans_pred &&
                         d_rot_pred, name, "null argument");
    hipLaunchKernelGGL(pose_losses_grad_kernel, dim3(ceil_div(b, 256)), dim3(256), 0, (hipStream_t)stream, b,
                       trans_pred, trans_label, trans_per, rot_jac, g_total, w_xyz, w_trans, w_rot, d_xyz_loss,
                       d_trans_pred, d_rot_pred);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}
