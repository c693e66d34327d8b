// edgeconv.hip -- the DGCNN edge-convolution block, fused (gfx950).
//
// Replaces, per encoder layer (reference models/pointnet_ycb_23_decoder_4.py:337-350
// and the three repeats below it):
//     get_edge_feature   utils/tf_util.py:635-669   e_ij = [c_i, n_ij - c_i]     [B,N,k,2C]
//     conv2d 1x1 + bias  utils/tf_util.py:161-166   y_ij = e_ij W + b            [B,N,k,Cout]
//     batch norm + ReLU  utils/tf_util.py:168-173
//     reduce_mean / reduce_max over k (models/...:350 / :615)                     [B,N,1,Cout]
// The reference materialises the k-fold edge tensor (63-168 MB per layer at B=32) and
// runs the GEMM over B*N*k rows.  Here the 1x1 convolution is split by linearity,
//     y_ij = c_i W_c + (n_ij - c_i) W_n + b
//          = (P_i - Q_i + b) + Q_{nbr(i,j)},     P = X W_c,  Q = X W_n,
// (W_c = W[0:C], W_n = W[C:2C]), so the MFMA GEMM runs over B*N rows (k times fewer
// flops) and the k-fold tensor never exists: the statistics pass and the
// normalise+ReLU+pool pass both re-gather Q rows, which are L2-resident (a cloud's Q
// is 256 KiB).  One wave owns one point; lanes own channels (coalesced 256/512 B row
// reads); the k neighbour indices are loaded once per point and broadcast by
// lane shuffles.  Backward mirrors it without atomics: a counting sort in LDS builds
// the reverse neighbour lists of each cloud, so dQ_m is GATHERED (sum over the points
// that have m as a neighbour) by the wave that owns m; then four small GEMMs produce
// dX and dW.
// This is an algebraic refactoring of the reference arithmetic: results agree with
// the edge-tensor formulation to fp32 round-off, not bitwise.
// The file is compiled TWICE (csrc/Makefile): edgeconv.o holds the forward kernels and entry points
// (-DCLOUDAAE_EC_PART=1, without the packed-fp32 feature: the compiler paired the two channels of a lane in
// ec_stats_kernel<2, ...> with an `op_sel` form -- Makefile, tests/test_isa_rules.py), edgeconv_bwd.o the backward
// ones (-DCLOUDAAE_EC_PART=2, with packed arithmetic: the gradient pass is bound by its vector instructions and 15 % of
// them are packed; the ISA test reads that object too).  Without the macro everything is in one object.
#ifndef CLOUDAAE_EC_PART
#define CLOUDAAE_EC_PART 0
#endif
#define EC_FWD (CLOUDAAE_EC_PART != 2)
#define EC_BWD (CLOUDAAE_EC_PART != 1)
#include "bn_common.h"
#include "gemm.h"
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

constexpr int EC_MAX_PARTS = 256;   // partial-sum rows of the statistics passes
constexpr int EC_WAVES = 4;         // apply passes: 4 waves per workgroup, grid sized by the work
constexpr int EC_STAT_WAVES = 16;   // statistics passes: 16 waves per workgroup (<= 256 workgroups)

// [parts][2][C] partial sums + a second [parts][2][C] block (third sum) + 4*C floats of scratch
__host__ __device__ inline size_t ec_ws_doubles(int C) { return (size_t)EC_MAX_PARTS * 4 * C + 2 * (size_t)C; }

struct EcArgs {
    int P;        // total points B*N
    int N, k, cout, ldpq;
    const float *pq;       // [P][2*cout]: P' | Q
    const float *bias;     // [cout]
    const int *nn_idx;     // [P][k], indices within the cloud
    const float *scale_shift;  // [2*cout]
    const float *gamma, *beta, *save_mean, *save_var;
    const float *dout;     // [P][lddo]
    int lddo;
    int training;
    unsigned short *out16;     // forward, optional: the pooled output once more as bfloat16 (round to nearest even) ...
    int ldo16;                 // ... rows ldo16 elements apart: the concat's bf16 twin (configs[2]: no separate conversion pass)
};

// fp32 -> bfloat16 bits, round to nearest even (NaN stays NaN): the conversion of cloudaae_to_bf16 (gemm_b16.hip)
__device__ __forceinline__ unsigned short ec_bf16_bits(float v)
{
    const __bf16 b = (__bf16)v;
    unsigned short w;
    __builtin_memcpy(&w, &b, 2);
    return w;
}

// x / d for a divisor that is the same for a whole launch (the neighbour count k), rd = 1.0f / d computed once: the
// IEEE sequence the compiler emits for `x / d` (scale both operands, quarter-rate reciprocal, two fma to refine it, product,
// two remainder / correction pairs, the last one as div_fmas, fix-up: eleven instructions) with the part that only
// depends on d taken out: rd is the correctly rounded reciprocal (better than the refined estimate), and d itself needs
// no scaling (1 <= d <= 2^20), so what is left per quotient is: scale the numerator, did that change it, product, FIX
// remainder / correction pairs (the last as div_fmas), fix-up: 6 or 8 instructions.  The gradient pass below divides once
// per gathered element and is bound by its vector instructions.
// Same quotient as `x / d`, bit for bit, for EVERY float x -- subnormal quotients, zeros, infinities, NaN included:
// cloudaae_selftest_div_by walks all 2^32 numerators (tests/test_01_layers_gpu.py).  FIX = 2: no difference for any
// divisor tried (1 .. 64 and a dozen others).  FIX = 1: none for most, k = 10 and k = 20 among them (what the model
// families use; the launcher takes FIX = 1 for exactly these), but e.g. d = 26 rounds 209 724 subnormal quotients the other
// way (a quotient that is exactly half way between two subnormals needs the first correction to land on it exactly).
// (The flag for div_fmas is "the scaling changed the numerator", not div_scale's own: that one is also set where the
//  hardware sequence expects a scaled DENOMINATOR, |x| >= 2^96 d.)
template <int FIX>
__device__ __forceinline__ float ec_div_by(float x, float d, float rd)
{
    bool unused;
    const float xs = __builtin_amdgcn_div_scalef(x, d, true, &unused);
    float q = xs * rd;
    if (FIX == 2)
        q = __builtin_fmaf(__builtin_fmaf(-d, q, xs), rd, q);
    return __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(__builtin_fmaf(-d, q, xs), rd, q, xs != x), d, x);
}
static int ec_div_corrections(int k) { return (k == 10 || k == 20) ? 1 : 2; }

// all k pre-activation rows of one point, for this lane's CPL channels
// US: the P' half of pq already holds U = P' - Q + bias (true in the backward kernels: every forward
// call leaves it so)
//
// FAST (the launcher takes it when k == KCAP: every configuration of the model): the loops over the neighbours carry no
// `j < k` test, and the j-th neighbour's index comes out of lane j with v_readlane -- a constant lane, so the index and the
// row address built from it live in SCALAR registers: a gather costs the vector unit nothing but the load itself.  (The
// general form asks for the index with a cross-lane shuffle and forms a 64-bit address per lane and gather: the forward
// kernels issued 27 vector instructions per edge value and were bound by that: `r06_step_all_pmc.json`.)
template <int CPL, int KCAP, bool US = false, bool FAST = false>
struct EcPoint {
    float y[KCAP][CPL];
    int nb[KCAP];

    __device__ __forceinline__ void load(const EcArgs &a, int pt, int lane)
    {
        const int base = (pt / a.N) * a.N;
        const int mine = lane < a.k ? a.nn_idx[(size_t)pt * a.k + lane] : 0;
        const float *row = a.pq + (size_t)pt * a.ldpq;
        float u[CPL];
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            const int c = lane + 64 * e;
            u[e] = US ? row[c] : (row[c] - row[a.cout + c]) + a.bias[c];
        }
#pragma unroll
        for (int j = 0; j < KCAP; ++j) {
            if (FAST || j < a.k) {
                nb[j] = base + (FAST ? __builtin_amdgcn_readlane(mine, j) : __shfl(mine, j, 64));
                const float *q = a.pq + (size_t)nb[j] * a.ldpq + a.cout;
#pragma unroll
                for (int e = 0; e < CPL; ++e)
                    y[j][e] = u[e] + q[(unsigned)lane + 64u * e];     // (unsigned: scalar base + 32-bit lane offset, no 64-bit vector address)
            }
        }
    }
};

// Two points at once: both index loads, then both own rows, then all 2k neighbour rows -- two dependent
// memory round trips for the PAIR.  A wave that handles its points one after the other pays the two
// round trips per point, and (vmcnt retires in order) the stores of one point in front of the loads of
// the next: ec_apply_kernel went from 11.5 to 26 us per layer when it got four more rows to store.
template <int CPL, int KCAP, bool US, bool FAST>
__device__ __forceinline__ void ec_load_pair(const EcArgs &a, int pt0, int pt1, int lane, EcPoint<CPL, KCAP, US, FAST> &p0,
                                             EcPoint<CPL, KCAP, US, FAST> &p1)
{
    const int mine0 = lane < a.k ? a.nn_idx[(size_t)pt0 * a.k + lane] : 0;
    const int mine1 = lane < a.k ? a.nn_idx[(size_t)pt1 * a.k + lane] : 0;
    const float *row0 = a.pq + (size_t)pt0 * a.ldpq, *row1 = a.pq + (size_t)pt1 * a.ldpq;
    float u0[CPL], u1[CPL], r0[2][CPL], r1[2][CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        const int c = lane + 64 * e;
        r0[0][e] = row0[c];
        r1[0][e] = row1[c];
        r0[1][e] = US ? 0.0f : row0[a.cout + c];
        r1[1][e] = US ? 0.0f : row1[a.cout + c];
    }
    const int base0 = (pt0 / a.N) * a.N, base1 = (pt1 / a.N) * a.N;
    float q0[KCAP][CPL], q1[KCAP][CPL];
#pragma unroll
    for (int j = 0; j < KCAP; ++j)
        if (FAST || j < a.k) {
            p0.nb[j] = base0 + (FAST ? __builtin_amdgcn_readlane(mine0, j) : __shfl(mine0, j, 64));
            p1.nb[j] = base1 + (FAST ? __builtin_amdgcn_readlane(mine1, j) : __shfl(mine1, j, 64));
            const float *g0 = a.pq + (size_t)p0.nb[j] * a.ldpq + a.cout;
            const float *g1 = a.pq + (size_t)p1.nb[j] * a.ldpq + a.cout;
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                q0[j][e] = g0[(unsigned)lane + 64u * e];
                q1[j][e] = g1[(unsigned)lane + 64u * e];
            }
        }
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        const float b = US ? 0.0f : a.bias[lane + 64 * e];
        u0[e] = US ? r0[0][e] : (r0[0][e] - r0[1][e]) + b;
        u1[e] = US ? r1[0][e] : (r1[0][e] - r1[1][e]) + b;
    }
#pragma unroll
    for (int j = 0; j < KCAP; ++j)
        if (FAST || j < a.k) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                p0.y[j][e] = u0[e] + q0[j][e];
                p1.y[j][e] = u1[e] + q1[j][e];
            }
        }
}

// Point -> wave assignment.  MI355X has 8 XCDs with private 4 MiB L2s and workgroup b runs
// on XCD b % 8 (observed placement; used for speed only).  A cloud's P'/Q/dOut rows are
// re-gathered ~k times, so all points of a cloud are given to workgroups of ONE XCD
// (cloud c -> XCD c % 8): the gather working set of an XCD is B/8 clouds (< 4 MiB at
// B=32) instead of the whole batch, and the re-reads hit L2 instead of the fabric.
template <int NW, typename F>
__device__ __forceinline__ void ec_for_each_point(const EcArgs &a, int wave, F &&body)
{
    const int B = a.P / a.N;
    if (B >= 8 && (gridDim.x & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
        for (int cloud = xcd; cloud < B; cloud += 8)
            for (int local = slot * NW + wave; local < a.N; local += nslot * NW)
                body(cloud * a.N + local);
    } else {
        for (int pt = blockIdx.x * NW + wave; pt < a.P; pt += gridDim.x * NW)
            body(pt);
    }
}

// the same sequence of points, handed out two at a time (pt1 = -1 when the wave's count is odd)
template <int NW, typename F>
__device__ __forceinline__ void ec_for_each_pair(const EcArgs &a, int wave, F &&body)
{
    int pend = -1;
    ec_for_each_point<NW>(a, wave, [&](int pt) {
        if (pend < 0) {
            pend = pt;
        } else {
            body(pend, pt);
            pend = -1;
        }
    });
    if (pend >= 0)
        body(pend, -1);
}

template <int CPL>
__device__ __forceinline__ void ec_block_reduce_store(double (&s)[CPL], double (&s2)[CPL], double *partial,
                                                      int cout, int lane, int wave)
{
    __shared__ double red[2][EC_STAT_WAVES][64 * CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        red[0][wave][lane + 64 * e] = s[e];
        red[1][wave][lane + 64 * e] = s2[e];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            const int c = lane + 64 * e;
            double a = red[0][0][c], b = red[1][0][c];
            for (int w = 1; w < EC_STAT_WAVES; ++w) {
                a += red[0][w][c];
                b += red[1][w][c];
            }
            partial[((size_t)blockIdx.x * 2 + 0) * cout + c] = a;
            partial[((size_t)blockIdx.x * 2 + 1) * cout + c] = b;
        }
    }
}

#if EC_FWD
template <int CPL, int KCAP, bool FAST>
__global__ __launch_bounds__(64 * EC_STAT_WAVES) void ec_stats_kernel(EcArgs a, double *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (the wave index in a scalar register: what follows from it -- point, cloud, row addresses -- is scalar arithmetic)
    double s[CPL], s2[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e)
        s[e] = s2[e] = 0.0;
    if constexpr (CPL * KCAP >= 40) {
        // 128 channels x 20 neighbours: two points at a time are 80 edge values per lane -- under the 128 registers of a
        // 16-wave workgroup the pair form spilled 200 of them (467 us for [32, 4096], layer 4 of BASELINE configs[4]).
        // One point at a time: forty gathers in flight per lane are enough, and the sums take the same order.  (At 64
        // channels x 20 neighbours the pair form stays: 86 against 96 us.)
        ec_for_each_point<EC_STAT_WAVES>(a, wave, [&](int pt) {
            EcPoint<CPL, KCAP, false, FAST> p;
            p.load(a, pt, lane);
#pragma unroll
            for (int j = 0; j < KCAP; ++j)
                if (FAST || j < a.k) {
#pragma unroll
                    for (int e = 0; e < CPL; ++e) {
                        s[e] += (double)p.y[j][e];
                        s2[e] += (double)p.y[j][e] * (double)p.y[j][e];
                    }
                }
        });
        ec_block_reduce_store<CPL>(s, s2, partial, a.cout, lane, wave);
        return;
    }
    ec_for_each_pair<EC_STAT_WAVES>(a, wave, [&](int pt0, int pt1) {
        EcPoint<CPL, KCAP, false, FAST> p0, p1;
        ec_load_pair(a, pt0, pt1 >= 0 ? pt1 : pt0, lane, p0, p1);
        const double w1 = pt1 >= 0 ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < KCAP; ++j)
            if (FAST || j < a.k) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    s[e] += (double)p0.y[j][e];
                    s2[e] += (double)p0.y[j][e] * (double)p0.y[j][e];
                }
            }
#pragma unroll
        for (int j = 0; j < KCAP; ++j)
            if (FAST || j < a.k) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    s[e] += w1 * (double)p1.y[j][e];
                    s2[e] += w1 * ((double)p1.y[j][e] * (double)p1.y[j][e]);
                }
            }
    });
    ec_block_reduce_store<CPL>(s, s2, partial, a.cout, lane, wave);
}

template <int CPL, int KCAP, int POOL, bool FAST, bool STATS>
__global__ __launch_bounds__(64 * EC_WAVES) void ec_apply_kernel(EcArgs a, float *__restrict__ out, int ldo,
                                                                float *__restrict__ ties,
                                                                float *__restrict__ edge_stats,
                                                                float *__restrict__ u_out)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (the wave index in a scalar register: what follows from it -- point, cloud, row addresses -- is scalar arithmetic)
    // mean pool in training mode: per point and channel, what the backward statistics need of its k
    // edges -- how many pass the ReLU, the sum of their x_hat, the sum of all x_hat (edge_stats[P][3][cout]).
    // The upstream gradient of every edge of a point is the same number, so backward gets its column
    // sums from these without gathering a single neighbour (ec_bwd_stats_pool_kernel).
    constexpr bool stats = STATS;      // (the launcher: POOL == 1 && edge_stats != nullptr)
    float sc[CPL], sh[CPL], mean[CPL], rstd[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        sc[e] = a.scale_shift[lane + 64 * e];
        sh[e] = a.scale_shift[a.cout + lane + 64 * e];
        mean[e] = stats ? a.save_mean[lane + 64 * e] : 0.0f;
        rstd[e] = stats ? bn_rsqrt(a.save_var[lane + 64 * e] + BN_EPS) : 0.0f;
    }
    auto finish = [&](const EcPoint<CPL, KCAP, false, FAST> &p, int pt, bool store) {
        float acc[CPL], cnt[CPL], sx[CPL], sall[CPL];
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            acc[e] = POOL == 2 ? -__builtin_inff() : 0.0f;
            cnt[e] = sx[e] = sall[e] = 0.0f;
        }
#pragma unroll
        for (int j = 0; j < KCAP; ++j)
            if (FAST || j < a.k) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    const float z = fmaxf(p.y[j][e] * sc[e] + sh[e], 0.0f);
                    if (POOL == 2) {
                        cnt[e] = z > acc[e] ? 1.0f : (z == acc[e] ? cnt[e] + 1.0f : cnt[e]);
                        acc[e] = fmaxf(acc[e], z);
                    } else {
                        acc[e] = acc[e] + z;
                        if (stats) {
                            const float xh = (p.y[j][e] - mean[e]) * rstd[e];
                            sall[e] += xh;
                            cnt[e] += z > 0.0f ? 1.0f : 0.0f;
                            sx[e] += z > 0.0f ? xh : 0.0f;
                        }
                    }
                }
            }
        if (!store)
            return;
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            const float o = POOL == 2 ? acc[e] : acc[e] / (float)a.k;
            out[(size_t)pt * ldo + lane + 64 * e] = o;
            if (a.out16 != nullptr)
                a.out16[(size_t)pt * a.ldo16 + lane + 64 * e] = ec_bf16_bits(o);
            if (POOL == 2 && ties != nullptr)
                ties[(size_t)pt * a.cout + lane + 64 * e] = cnt[e];
            if (stats) {
                float *es = edge_stats + (size_t)pt * 3 * a.cout + lane + 64 * e;
                es[0] = cnt[e];
                es[a.cout] = sx[e];
                es[2 * a.cout] = sall[e];
            }
            if (u_out != nullptr) {
                // U_i = P'_i - Q_i + b replaces P'_i (only this wave reads that half-row): the backward
                // passes fetch ONE row per source point instead of two
                const int c = lane + 64 * e;
                const float *row = a.pq + (size_t)pt * a.ldpq;
                u_out[(size_t)pt * a.ldpq + c] = (row[c] - row[a.cout + c]) + a.bias[c];
            }
        }
    };
    if constexpr (KCAP >= 20) {
        // one point at a time from 20 neighbours up: twenty or forty gathers per lane are in flight anyway, and the pair
        // form needs 113 / 135 registers (64 / 128 channels: four / three waves per SIMD).  Measured at [32, 4096, k = 20]:
        // 103 -> 86 us (64 channels), 215 -> 169 us (128); at k = 10 the pair form stays (16.7 against 17.7 us at [32, 1024])
        ec_for_each_point<EC_WAVES>(a, wave, [&](int pt) {
            EcPoint<CPL, KCAP, false, FAST> p;
            p.load(a, pt, lane);
            finish(p, pt, true);
        });
        return;
    }
    ec_for_each_pair<EC_WAVES>(a, wave, [&](int pt0, int pt1) {
        EcPoint<CPL, KCAP, false, FAST> p0, p1;
        ec_load_pair(a, pt0, pt1 >= 0 ? pt1 : pt0, lane, p0, p1);
        finish(p0, pt0, true);
        finish(p1, pt1, pt1 >= 0);
    });
}

#endif   // EC_FWD

#if EC_BWD
// upstream gradient of z_ij for one point: mean -> dout/k; max -> dout shared among
// the equal maxima (tf.reduce_max gradient), both masked by ReLU.
template <int CPL, int KCAP, int POOL>
__device__ __forceinline__ void ec_upstream(const EcArgs &a, const EcPoint<CPL, KCAP, true> &p, int pt, int lane,
                                            const float (&sc)[CPL], const float (&sh)[CPL],
                                            float (&dz)[KCAP][CPL])
{
    float g[CPL], zmax[CPL], ties[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        g[e] = a.dout[(size_t)pt * a.lddo + lane + 64 * e];
        zmax[e] = -__builtin_inff();
        ties[e] = 0.0f;
    }
    if (POOL == 2) {
#pragma unroll
        for (int j = 0; j < KCAP; ++j)
            if (j < a.k) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    const float z = fmaxf(p.y[j][e] * sc[e] + sh[e], 0.0f);
                    if (z > zmax[e]) {
                        zmax[e] = z;
                        ties[e] = 1.0f;
                    } else if (z == zmax[e]) {
                        ties[e] += 1.0f;
                    }
                }
            }
    }
#pragma unroll
    for (int j = 0; j < KCAP; ++j)
        if (j < a.k) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                const float z = fmaxf(p.y[j][e] * sc[e] + sh[e], 0.0f);
                float d = POOL == 2 ? (z == zmax[e] ? g[e] / ties[e] : 0.0f) : g[e] / (float)a.k;
                if (!(z > 0.0f))
                    d = 0.0f;
                dz[j][e] = d;
            }
        }
}

template <int CPL, int KCAP, int POOL>
__global__ __launch_bounds__(64 * EC_STAT_WAVES) void ec_bwd_stats_kernel(EcArgs a, double *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (the wave index in a scalar register: what follows from it -- point, cloud, row addresses -- is scalar arithmetic)
    float sc[CPL], sh[CPL], mean[CPL], rstd[CPL];
    double s[CPL], s2[CPL], s3[CPL], zero[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        const int c = lane + 64 * e;
        bn_scale_shift_of(a.gamma, a.beta, a.save_mean, a.save_var, c, sc[e], sh[e]);
        mean[e] = a.save_mean[c];
        rstd[e] = bn_rsqrt(a.save_var[c] + BN_EPS);
        s[e] = s2[e] = s3[e] = zero[e] = 0.0;
    }
    ec_for_each_point<EC_STAT_WAVES>(a, wave, [&](int pt) {
        EcPoint<CPL, KCAP, true> p;
        p.load(a, pt, lane);
        float dz[KCAP][CPL];
        ec_upstream<CPL, KCAP, POOL>(a, p, pt, lane, sc, sh, dz);
#pragma unroll
        for (int j = 0; j < KCAP; ++j)
            if (j < a.k) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    const float xh = (p.y[j][e] - mean[e]) * rstd[e];
                    s[e] += (double)dz[j][e];
                    s2[e] += (double)dz[j][e] * (double)xh;
                    s3[e] += (double)xh;
                }
            }
    });
    ec_block_reduce_store<CPL>(s, s2, partial, a.cout, lane, wave);
    __syncthreads();
    // third sum (sum of x_hat, ~0): needed for the conv-bias gradient, see ec_bwd_finalize_kernel
    ec_block_reduce_store<CPL>(s3, zero, partial + (size_t)EC_MAX_PARTS * 2 * a.cout, a.cout, lane, wave);
}

// The same three sums for the mean pool, from what the forward apply pass left per point (edge_stats):
// every edge of point i has the upstream gradient dout_i / k (times its ReLU mask), so
//   sum dz = sum_i (dout_i / k) cnt_i,  sum dz x_hat = sum_i (dout_i / k) sx_i,  sum x_hat = sum_i sall_i
// -- a streaming pass over four [P][cout] arrays instead of k gathers per point.
template <int CPL>
__global__ __launch_bounds__(64 * EC_STAT_WAVES) void ec_bwd_stats_pool_kernel(EcArgs a,
                                                                              const float *__restrict__ edge_stats,
                                                                              double *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (the wave index in a scalar register: what follows from it -- point, cloud, row addresses -- is scalar arithmetic)
    double s[CPL], s2[CPL], s3[CPL], zero[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e)
        s[e] = s2[e] = s3[e] = zero[e] = 0.0;
    ec_for_each_pair<EC_STAT_WAVES>(a, wave, [&](int pt0, int pt1) {
        const int q1 = pt1 >= 0 ? pt1 : pt0;
        const double w1 = pt1 >= 0 ? 1.0 : 0.0;
        float v0[4][CPL], v1[4][CPL];       // both points' loads first
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            const int c = lane + 64 * e;
            const float *e0 = edge_stats + (size_t)pt0 * 3 * a.cout + c, *e1 = edge_stats + (size_t)q1 * 3 * a.cout + c;
            v0[0][e] = a.dout[(size_t)pt0 * a.lddo + c];
            v1[0][e] = a.dout[(size_t)q1 * a.lddo + c];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                v0[1 + t][e] = e0[t * a.cout];
                v1[1 + t][e] = e1[t * a.cout];
            }
        }
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            const double g0 = (double)(v0[0][e] / (float)a.k), g1 = w1 * (double)(v1[0][e] / (float)a.k);
            s[e] += g0 * (double)v0[1][e];
            s2[e] += g0 * (double)v0[2][e];
            s3[e] += (double)v0[3][e];
            s[e] += g1 * (double)v1[1][e];
            s2[e] += g1 * (double)v1[2][e];
            s3[e] += w1 * (double)v1[3][e];
        }
    });
    ec_block_reduce_store<CPL>(s, s2, partial, a.cout, lane, wave);
    __syncthreads();
    ec_block_reduce_store<CPL>(s3, zero, partial + (size_t)EC_MAX_PARTS * 2 * a.cout, a.cout, lane, wave);
}

// dgamma = sum dz*xhat, dbeta = sum dz, m1/m2 = their means, and the conv-bias gradient
//   dbias = sum_ij dy_ij = gamma*rstd*((sum dz - cnt*m1) - m2 * sum xhat)
// which is analytically zero (a bias in front of a batch norm) and numerically round-off,
// exactly like the reference's autodiff value; computing it from the column sums avoids
// funnelling every workgroup through atomics on the same 64-128 addresses (that was 100 us).
static __global__ __launch_bounds__(BN_FIN_THREADS) void ec_bwd_finalize_kernel(
    int C, const double *__restrict__ partial, const double *__restrict__ partial3, int parts, double count,
    int training, const float *__restrict__ gamma, const float *__restrict__ save_var,
    float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dbias, float *__restrict__ m12,
    const double *__restrict__ gsums, double gcount)
{
    // gsums != nullptr (SyncBN): the two sums over all ranks' edges and their count; see bn_bwd_finalize_kernel
    const int c = bn_fin_channel(), pl = bn_fin_lane();
    double s, s2, s3;
    bn_reduce_partials(partial, parts, C, c, pl, s, s2, partial3, &s3);
    if (c >= C || pl != 0)
        return;
    if (dbeta != nullptr)
        dbeta[c] = (float)s;
    if (dgamma != nullptr)
        dgamma[c] = (float)s2;
    const float m1 = training ? (gsums != nullptr ? (float)(gsums[c] / gcount) : (float)(s / count)) : 0.0f;
    const float m2 = training ? (gsums != nullptr ? (float)(gsums[C + c] / gcount) : (float)(s2 / count)) : 0.0f;
    m12[c] = m1;
    m12[C + c] = m2;
    if (dbias != nullptr) {
        const double gr = (double)gamma[c] * (double)bn_rsqrt(save_var[c] + BN_EPS);
        dbias[c] = (float)(gr * ((s - count * (double)m1) - (double)m2 * s3));
    }
}

// Reverse neighbour lists of one cloud (one workgroup per cloud): for every point m the
// points i that have m among their k neighbours.  rev_off[N+1] (offsets into rev_src),
// rev_src[N*k] (source point i, cloud-local).  Counting sort in LDS; the order inside a
// list depends on LDS-atomic arrival order (so dQ sums are reproducible to fp32 round-off,
// not bitwise, exactly like the atomic scatter it replaces).
constexpr int EC_REV_MAX = 8;       // layers whose lists one launch can build
struct EcRevJobs {
    const int *nn_idx[EC_REV_MAX];
    int *rev[EC_REV_MAX];           // rev_off [b][N+1] followed by rev_src [b][N*k]
};

// grid (clouds, layers): the neighbour lists of every layer of the encoder exist once its forward pass
// is over, so the backward pass builds all their reverse lists with ONE launch (four ~11 us launches of
// 32 workgroups each otherwise).
constexpr int EC_REV_THREADS = 1024;    // (512: 14.2 us at [32, 1024, k = 10] x 4 layers, 135 us at [32, 4096, k = 20])
__global__ __launch_bounds__(EC_REV_THREADS) void ec_revlist_kernel(int B, int N, int k, EcRevJobs jobs, int sorted)
{
    extern __shared__ int cnt[];
    __shared__ int wsum[EC_REV_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int *nn_idx = jobs.nn_idx[blockIdx.y];
    int *rev_off = jobs.rev[blockIdx.y], *rev_src = rev_off + (size_t)B * (N + 1);
    const int *idx = nn_idx + (size_t)blockIdx.x * N * k;
    int *off = rev_off + (size_t)blockIdx.x * (N + 1);
    int *src = rev_src + (size_t)blockIdx.x * N * k;
    const int E = N * k;
    for (int m = t; m < N; m += EC_REV_THREADS)
        cnt[m] = 0;
    __syncthreads();
    // the index loads are issued eight at a time ahead of the LDS atomics that consume them: one global
    // round trip per batch instead of one per edge (a loop of load -> atomic pairs costs the latency
    // E / threads times over)
    constexpr int RU = 8;
    for (int e0 = t; e0 < E; e0 += EC_REV_THREADS * RU) {
        int v[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u)
            v[u] = e0 + EC_REV_THREADS * u < E ? idx[e0 + EC_REV_THREADS * u] : -1;
#pragma unroll
        for (int u = 0; u < RU; ++u)
            if (v[u] >= 0)
                atomicAdd(&cnt[v[u]], 1);
    }
    __syncthreads();
    const int per = (N + EC_REV_THREADS - 1) / EC_REV_THREADS;
    const int lo = min(N, t * per), hi = min(N, lo + per);
    int local = 0;
    for (int m = lo; m < hi; ++m)
        local += cnt[m];
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d)
            incl += o;
    }
    if (lane == 63)
        wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w)
        base += wsum[w];
    int run = base + incl - local;
    for (int m = lo; m < hi; ++m) {
        const int c = cnt[m];
        cnt[m] = run;      // becomes the fill cursor
        off[m] = run;
        run += c;
    }
    if (t == 0)
        off[N] = E;
    __syncthreads();
    for (int e0 = t; e0 < E; e0 += EC_REV_THREADS * RU) {
        int v[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u)
            v[u] = e0 + EC_REV_THREADS * u < E ? idx[e0 + EC_REV_THREADS * u] : -1;
#pragma unroll
        for (int u = 0; u < RU; ++u)
            if (v[u] >= 0) {
                const int pos = atomicAdd(&cnt[v[u]], 1);
                src[pos] = (e0 + EC_REV_THREADS * u) / k;
            }
    }
    if (sorted) {
        // deterministic mode: the slots of a list were handed out in arrival order; sorted by source point the list --
        // and with it the summation order of the backward gather -- is the same in every run
        __syncthreads();
        for (int mpt = t; mpt < N; mpt += EC_REV_THREADS) {
            const int a = off[mpt], e = (mpt + 1 < N) ? off[mpt + 1] : E;
            for (int i = a + 1; i < e; ++i) {
                const int v = src[i];
                int j = i - 1;
                while (j >= a && src[j] > v) {
                    src[j + 1] = src[j];
                    --j;
                }
                src[j + 1] = v;
            }
        }
    }
}

// dy_ij = gamma*rstd*((dz_ij - m1) - xhat_ij*m2)
//   dP'_m = S_m = sum_j dy_mj                     (own neighbours, forward direction)
//   dQ_m  = T_m - S_m,  T_m = sum_{(i,j): nbr(i,j)=m} dy_ij   (reverse list, gathered)
//   dbias += S_m
// No atomics on the big tensors and no zero-fill: every dpq element is written once.
template <int CPL, int KCAP, int POOL>
__global__ __launch_bounds__(64 * EC_WAVES) void ec_bwd_apply_kernel(
    EcArgs a, const float *__restrict__ m12, const int *__restrict__ rev_off, const int *__restrict__ rev_src,
    const float *__restrict__ fwd_out, int ldo, const float *__restrict__ ties, float *__restrict__ dpq,
    const float *__restrict__ edge_stats)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (the wave index in a scalar register: what follows from it -- point, cloud, row addresses -- is scalar arithmetic)
    float sc[CPL], sh[CPL], mean[CPL], rstd[CPL], gr[CPL], m1[CPL], m2[CPL], bias[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        const int c = lane + 64 * e;
        bn_scale_shift_of(a.gamma, a.beta, a.save_mean, a.save_var, c, sc[e], sh[e]);
        mean[e] = a.save_mean[c];
        rstd[e] = bn_rsqrt(a.save_var[c] + BN_EPS);
        gr[e] = a.gamma[c] * rstd[e];
        m1[e] = m12[c];
        m2[e] = m12[a.cout + c];
        bias[e] = a.bias[c];
    }
    const float invk = 1.0f;  // (mean pool divides by k exactly as the forward direction does, below)
    (void)invk;
    ec_for_each_point<EC_WAVES>(a, wave, [&](int pt) {
        const int cloud = pt / a.N, m = pt - cloud * a.N;
        // the reverse list's bounds and first 64 sources are fetched up front so that they are
        // in flight together with the forward-direction gathers (dependent stages cost ~20 us
        // each across the grid)
        const int *off = rev_off + (size_t)cloud * (a.N + 1);
        const int *src = rev_src + (size_t)cloud * a.N * a.k;
        const int beg = off[m], end = off[m + 1];
        int mine = (beg + lane < end) ? src[beg + lane] : 0;
        float S[CPL], T[CPL], Qm[CPL];
        if (POOL == 1 && edge_stats != nullptr) {
            // S_i = sum_j gr ((dz_ij - m1) - x_hat_ij m2) with dz_ij = mask_ij dout_i / k: from the point's
            // edge statistics of the forward pass, no neighbour is fetched
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                const int c = lane + 64 * e;
                const float *es = edge_stats + (size_t)pt * 3 * a.cout + c;
                const float gk = a.dout[(size_t)pt * a.lddo + c] / (float)a.k;
                S[e] = gr[e] * ((gk * es[0] - (float)a.k * m1[e]) - es[2 * a.cout] * m2[e]);
            }
        } else {
            EcPoint<CPL, KCAP, true> p;
            p.load(a, pt, lane);
            float dz[KCAP][CPL];
            ec_upstream<CPL, KCAP, POOL>(a, p, pt, lane, sc, sh, dz);
#pragma unroll
            for (int e = 0; e < CPL; ++e)
                S[e] = 0.0f;
#pragma unroll
            for (int j = 0; j < KCAP; ++j)
                if (j < a.k) {
#pragma unroll
                    for (int e = 0; e < CPL; ++e) {
                        const float xh = (p.y[j][e] - mean[e]) * rstd[e];
                        const float dy = gr[e] * ((dz[j][e] - m1[e]) - xh * m2[e]);
                        S[e] = S[e] + dy;
                    }
                }
        }
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            T[e] = 0.0f;
            Qm[e] = a.pq[(size_t)pt * a.ldpq + a.cout + lane + 64 * e];
        }
        constexpr int RU = CPL == 1 ? 8 : 4;   // sources gathered per batch
        for (int s0 = beg; s0 < end; s0 += 64) {
            const int cntc = min(64, end - s0);
            if (s0 != beg)
                mine = lane < cntc ? src[s0 + lane] : 0;
            for (int q0 = 0; q0 < cntc; q0 += RU) {
                float pi[RU][CPL], gi[RU][CPL], oi[RU][CPL], ti[RU][CPL];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const bool on = q0 + u < cntc;
                    const int i = cloud * a.N + __shfl(mine, on ? q0 + u : q0, 64);
                    const float *row = a.pq + (size_t)i * a.ldpq + lane;
#pragma unroll
                    for (int e = 0; e < CPL; ++e) {
                        pi[u][e] = row[64 * e];
                        gi[u][e] = on ? a.dout[(size_t)i * a.lddo + lane + 64 * e] : 0.0f;
                        if (POOL == 2) {
                            oi[u][e] = fwd_out[(size_t)i * ldo + lane + 64 * e];
                            ti[u][e] = ties[(size_t)i * a.cout + lane + 64 * e];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    if (q0 + u < cntc) {
#pragma unroll
                        for (int e = 0; e < CPL; ++e) {
                            const float uu = pi[u][e];      // U of the source point
                            const float y = uu + Qm[e];
                            const float z = fmaxf(y * sc[e] + sh[e], 0.0f);
                            float d = POOL == 2 ? (z == oi[u][e] ? gi[u][e] / ti[u][e] : 0.0f)
                                                : gi[u][e] / (float)a.k;
                            if (!(z > 0.0f))
                                d = 0.0f;
                            const float xh = (y - mean[e]) * rstd[e];
                            T[e] = T[e] + gr[e] * ((d - m1[e]) - xh * m2[e]);
                        }
                    }
                }
            }
        }
        float *mineo = dpq + (size_t)pt * a.ldpq + lane;
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            mineo[64 * e] = S[e];
            mineo[a.cout + 64 * e] = T[e] - S[e];
        }
    });
}

// The same pass for mean pooling in training (S from the forward pass's edge statistics), with the reverse list
// gathered SEVERAL SOURCE ROWS PER LOAD INSTRUCTION: a lane owns four consecutive channels (16-byte loads), so the
// 64 lanes cover 64 / (COUT / 4) = 4 (COUT = 64) or 2 (COUT = 128) source rows at once, and a batch of four such
// loads per array keeps 16 (8) rows = 8 KB of U and dOut in flight per wave.  ec_bwd_apply_kernel moves 256 bytes
// per load instruction (one row, four bytes per lane) and is bound by the latency of its dependent gathers.  The
// partial sums of the row groups meet in two (one) cross-lane steps; S is evaluated exactly as above.
template <int COUT, int FIX>
__global__ __launch_bounds__(64 * EC_WAVES) void ec_bwd_apply_mean4_kernel(
    EcArgs a, const float *__restrict__ m12, const int *__restrict__ rev_off, const int *__restrict__ rev_src,
    float *__restrict__ dpq, const float *__restrict__ edge_stats)
{
    constexpr int LPR = COUT / 4;          // lanes per row
    constexpr int SPI = 64 / LPR;          // source rows per load instruction
    constexpr int RU = 4;                  // load instructions per array in flight
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (the wave index in a scalar register: what follows from it -- point, cloud, row addresses -- is scalar arithmetic)
    const int grp = lane / LPR, c0 = 4 * (lane % LPR);
    float sc[4], sh[4], mean[4], rstd[4], gr[4], m1[4], m2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        bn_scale_shift_of(a.gamma, a.beta, a.save_mean, a.save_var, c, sc[e], sh[e]);
        mean[e] = a.save_mean[c];
        rstd[e] = bn_rsqrt(a.save_var[c] + BN_EPS);
        gr[e] = a.gamma[c] * rstd[e];
        m1[e] = m12[c];
        m2[e] = m12[COUT + c];
    }
    const float fk = (float)a.k, rk = 1.0f / fk;
    ec_for_each_point<EC_WAVES>(a, wave, [&](int pt) {
        const int cloud = pt / a.N, m = pt - cloud * a.N;
        const int *off = rev_off + (size_t)cloud * (a.N + 1);
        const int *src = rev_src + (size_t)cloud * a.N * a.k;
        const int beg = off[m], end = off[m + 1];
        int mine = (beg + lane < end) ? src[beg + lane] : 0;
        // own edges: S_i = sum_j gr ((dz_ij - m1) - x_hat_ij m2), dz_ij = mask_ij dout_i / k, from the edge statistics
        const float4v es0 = *reinterpret_cast<const float4v *>(edge_stats + (size_t)pt * 3 * COUT + c0);
        const float4v es2 = *reinterpret_cast<const float4v *>(edge_stats + (size_t)pt * 3 * COUT + 2 * COUT + c0);
        const float4v go = *reinterpret_cast<const float4v *>(a.dout + (size_t)pt * a.lddo + c0);
        const float4v qm = *reinterpret_cast<const float4v *>(a.pq + (size_t)pt * a.ldpq + COUT + c0);
        float S[4], T[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gk = ec_div_by<FIX>(go[e], fk, rk);
            S[e] = gr[e] * ((gk * es0[e] - fk * m1[e]) - es2[e] * m2[e]);
            T[e] = 0.0f;
        }
        for (int s0 = beg; s0 < end; s0 += 64) {
            const int cntc = min(64, end - s0);
            if (s0 != beg)
                mine = lane < cntc ? src[s0 + lane] : 0;
            for (int q0 = 0; q0 < cntc; q0 += SPI * RU) {
                float4v ui[RU], gi[RU];
                bool on[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int t = q0 + u * SPI + grp;       // this lane group's source of load u
                    on[u] = t < cntc;
                    const int i = cloud * a.N + __shfl(mine, on[u] ? t : q0, 64);
                    ui[u] = *reinterpret_cast<const float4v *>(a.pq + (size_t)i * a.ldpq + c0);
                    gi[u] = *reinterpret_cast<const float4v *>(a.dout + (size_t)i * a.lddo + c0);
                }
#pragma unroll
                for (int u = 0; u < RU; ++u)
                    if (on[u]) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float y = ui[u][e] + qm[e];   // U of the source point + Q of this one
                            const float z = y * sc[e] + sh[e];  // (ReLU passes the edge iff this is positive: no max needed to ask)
                            float d = ec_div_by<FIX>(gi[u][e], fk, rk);      // == gi / k, bit for bit
                            if (!(z > 0.0f))
                                d = 0.0f;
                            const float xh = (y - mean[e]) * rstd[e];
                            T[e] = T[e] + gr[e] * ((d - m1[e]) - xh * m2[e]);
                        }
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1)
                T[e] += __shfl_xor(T[e], o, 64);
        if (grp == 0) {
            float *mineo = dpq + (size_t)pt * a.ldpq + c0;
            *reinterpret_cast<float4v *>(mineo) = float4v{S[0], S[1], S[2], S[3]};
            *reinterpret_cast<float4v *>(mineo + COUT) = float4v{T[0] - S[0], T[1] - S[1], T[2] - S[2], T[3] - S[3]};
        }
    });
}

#endif   // EC_BWD

static int ec_stat_grid(int P)
{
    int g = ceil_div(ceil_div(P, EC_STAT_WAVES * 2), 8) * 8;   // multiple of 8: one share per XCD
    if (g > EC_MAX_PARTS)
        g = EC_MAX_PARTS;
    return g < 8 ? 8 : g;
}
static int ec_apply_grid(int P)
{
    int g = ceil_div(ceil_div(P, EC_WAVES * 2), 8) * 8;   // >= 2 points per wave, multiple of 8
    if (g > 4096)
        g = 4096;
    return g < 8 ? 8 : g;
}

// dispatch over (channels per lane, neighbour capacity, pool mode)
#define EC_DISPATCH(KERNEL_CALL)                                         \
    do {                                                                 \
        if (cpl == 1 && kcap == 10) { KERNEL_CALL(1, 10); }              \
        else if (cpl == 1 && kcap == 20) { KERNEL_CALL(1, 20); }         \
        else if (cpl == 1) { KERNEL_CALL(1, 32); }                       \
        else if (cpl == 2 && kcap == 10) { KERNEL_CALL(2, 10); }         \
        else if (cpl == 2 && kcap == 20) { KERNEL_CALL(2, 20); }         \
        else { KERNEL_CALL(2, 32); }                                     \
    } while (0)

} // namespace cloudaae

using namespace cloudaae;

// the block's dense products: fp32 operands, or rounded to bf16 on the way to the matrix cores; B and/or C
// may be the FOLDED view of the [2*cin, cout] kernel as [cin, 2*cout] = [W_centre | W_neighbour] (gemm.h)
static int ec_gemm(const char *name, int bf16, int ta, int tb, int M, int N, int K, const float *A, int lda,
                   const float *B, int ldb, float *C, int ldc, int accumulate, int fold_b, int fold_c,
                   cloudaae_stream_t stream)
{
    return bf16 ? gemm_bf16_launch(name, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, nullptr, accumulate, fold_b, fold_c,
                                   (hipStream_t)stream)
                : gemm_f32_launch(name, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, nullptr, accumulate, fold_b, fold_c,
                                  (hipStream_t)stream);
}

#if EC_BWD
static int ec_launch_revlists(const char *name, int count, int b, int n, int k, const int *const *nn_idx,
                              int *const *rev, hipStream_t s)
{
    CLOUDAAE_REQUIRE(count >= 1 && count <= EC_REV_MAX, name, "1 to 8 neighbour lists per launch");
    EcRevJobs jobs = {};
    for (int i = 0; i < count; ++i) {
        CLOUDAAE_REQUIRE(nn_idx[i] != nullptr && rev[i] != nullptr, name, "null argument");
        jobs.nn_idx[i] = nn_idx[i];
        jobs.rev[i] = rev[i];
    }
    const size_t lds = (size_t)n * sizeof(int);
    CLOUDAAE_REQUIRE(lds <= 150 * 1024, name, "cloud too large for the LDS counting sort");
    if (lds > 48 * 1024)
        CLOUDAAE_CHECK_HIP(hipFuncSetAttribute((const void *)ec_revlist_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), name);
    hipLaunchKernelGGL(ec_revlist_kernel, dim3(b, count), dim3(EC_REV_THREADS), lds, s, b, n, k, jobs,
                       CLOUDAAE_KNOB("CLOUDAAE_DETERMINISTIC", 0) != 0 ? 1 : 0);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_edgeconv_revlists(int count, int b, int n, int k, const int *const *nn_idx,
                                            int *const *rev_scratch, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_edgeconv_revlists";
    CLOUDAAE_REQUIRE(b > 0 && n > 0 && k > 0 && nn_idx && rev_scratch, name, "bad argument");
    return ec_launch_revlists(name, count, b, n, k, nn_idx, rev_scratch, (hipStream_t)stream);
}

// every float as the numerator of ec_div_by: count[0] = arguments whose quotient differs in its bits from x / d
// (two NaNs count as equal), count[1] = the largest magnitude among them (its bits)
template <int FIX>
__global__ __launch_bounds__(256) void ec_selftest_div_kernel(float d, unsigned long long *__restrict__ count)
{
    const float rd = 1.0f / d;
    unsigned long long bad = 0, largest = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += 256ull * gridDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const float want = x / d, got = ec_div_by<FIX>(x, d, rd);
        if (__float_as_uint(want) != __float_as_uint(got) && !(want != want && got != got)) {
            largest = max(largest, i & 0x7fffffffull);
            ++bad;
        }
    }
    if (bad) {
        atomicAdd(&count[0], bad);
        atomicMax(&count[1], largest);
    }
}

CLOUDAAE_API int cloudaae_selftest_div_by(float d, int corrections, unsigned long long *count, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_selftest_div_by";
    CLOUDAAE_REQUIRE(count != nullptr && d >= 1.0f && d <= 1048576.0f && corrections >= 0 && corrections <= 2, name, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (corrections == 0)       // what the launcher takes for a layer with d neighbours
        corrections = d == (float)(int)d ? ec_div_corrections((int)d) : 2;
    CLOUDAAE_CHECK_HIP(hipMemsetAsync(count, 0, 2 * sizeof(unsigned long long), s), name);
    if (corrections == 1)
        hipLaunchKernelGGL(ec_selftest_div_kernel<1>, dim3(4096), dim3(256), 0, s, d, count);
    else
        hipLaunchKernelGGL(ec_selftest_div_kernel<2>, dim3(4096), dim3(256), 0, s, d, count);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

#endif   // EC_BWD

#if EC_FWD
// [P' | Q] = X [W_centre | W_neighbour] for 64 input channels, streamed.  The general fp32 product (gemm.hip) runs these
// shapes -- 32768 x 64 times 64 x 128 -- at 2.0 - 2.3 TB/s: a workgroup is one chain load -> stage -> 32 MFMAs -> store and
// fetches its own copy of the weights (`notes_gemm_f32.md`).  Here a wave keeps its 32 (64) columns of the folded kernel in
// REGISTERS for the whole launch (32 (64) B operands of v_mfma_f32_32x32x2_f32), the workgroup walks 32-row tiles of X:
// global -> registers one tile ahead, registers -> LDS as [32 even channels | 32 odd | pad] (the layout the A operand reads
// with eight ds_read_b128: lane = row, half = parity), two LDS buffers so that one barrier per tile is enough.  The k order
// of the accumulation is the general kernel's (0, 1, ..., 63 in pairs), so the product has the same bits
// (tests/test_01_layers_gpu.py: test_edge_conv_streamed_product_equals_the_general_one).  11.5 -> 9.6 us (128 columns) and
// 19.1 -> 15.6 us (256) at 32768 rows, whatever the grid: a launch that moves 25 MB does not get much below that.
typedef float ec_f32x16 __attribute__((ext_vector_type(16)));
constexpr int EC_PQ_LD = 68;
template <int NOUT>
__global__ __launch_bounds__(256) void ec_pq_stream_kernel(int P, const float *__restrict__ x, int ldx,
                                                          const float *__restrict__ w, int cout, float *__restrict__ pq)
{
    constexpr int CIN = 64, TN = NOUT / 128;                 // 32 x 32 tiles per wave
    __shared__ __attribute__((aligned(16))) float lds[2][32 * EC_PQ_LD];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int col = lane & 31, half = lane >> 5;
    const int n0 = wave * (NOUT / 4);
    // B operands: step s, tile j: Bf[2 s + half][n0 + 32 j + col], Bf(kk, c) = w[((c / cout) * CIN + kk) * cout + c % cout]
    float b[32][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int c = n0 + 32 * j + col;
        const float *wc = w + ((size_t)(c / cout) * CIN) * cout + c % cout;
#pragma unroll
        for (int st = 0; st < 32; ++st)
            b[st][j] = wc[(size_t)(2 * st + half) * cout];
    }
    const int ntiles = P / 32;
    // staging share of a thread: two 16-byte pieces of the tile's 32 x 64 floats (16 lanes per row: coalesced)
    const int r0 = threadIdx.x >> 4, f = threadIdx.x & 15;     // rows r0 and r0 + 16, floats 4 f .. 4 f + 3
    float4v g0, g1;
    int t = blockIdx.x;
    if (t < ntiles) {
        g0 = *reinterpret_cast<const float4v *>(x + (size_t)(32 * t + r0) * ldx + 4 * f);
        g1 = *reinterpret_cast<const float4v *>(x + (size_t)(32 * t + r0 + 16) * ldx + 4 * f);
    }
    for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
        float *buf = lds[it & 1];
        *reinterpret_cast<float2 *>(buf + r0 * EC_PQ_LD + 2 * f) = float2{g0.x, g0.z};
        *reinterpret_cast<float2 *>(buf + r0 * EC_PQ_LD + 32 + 2 * f) = float2{g0.y, g0.w};
        *reinterpret_cast<float2 *>(buf + (r0 + 16) * EC_PQ_LD + 2 * f) = float2{g1.x, g1.z};
        *reinterpret_cast<float2 *>(buf + (r0 + 16) * EC_PQ_LD + 32 + 2 * f) = float2{g1.y, g1.w};
        __syncthreads();
        const int tn = t + (int)gridDim.x;
        if (tn < ntiles) {                                   // the next tile travels under this one's products
            g0 = *reinterpret_cast<const float4v *>(x + (size_t)(32 * tn + r0) * ldx + 4 * f);
            g1 = *reinterpret_cast<const float4v *>(x + (size_t)(32 * tn + r0 + 16) * ldx + 4 * f);
        }
        float4v a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            a[u] = *reinterpret_cast<const float4v *>(buf + col * EC_PQ_LD + 32 * half + 4 * u);
        ec_f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[j][r] = 0.0f;
#pragma unroll
        for (int st = 0; st < 32; ++st)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st >> 2][st & 3], b[st][j], acc[j], 0, 0, 0);
        // lane holds column (lane & 31), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        float *c0 = pq + (size_t)(32 * t + 4 * half) * NOUT + n0 + col;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                c0[(size_t)((r & 3) + 8 * (r >> 2)) * NOUT + 32 * j] = acc[j][r];
    }
}

CLOUDAAE_API long long cloudaae_edgeconv_workspace_bytes(int cout)
{
    return (long long)(ec_ws_doubles(cout) * sizeof(double));
}

#endif   // EC_FWD

static int ec_check(const char *name, int b, int n, int k, int cin, int cout, int pool_mode)
{
    CLOUDAAE_REQUIRE(b > 0 && n > 0 && cin > 0, name, "bad size");
    CLOUDAAE_REQUIRE(k >= 1 && k <= 32, name, "k must be in [1,32]");
    CLOUDAAE_REQUIRE(cout == 64 || cout == 128, name, "fused edge-conv supports 64 or 128 output channels");
    CLOUDAAE_REQUIRE(pool_mode == 1 || pool_mode == 2, name, "pool_mode must be 1 (mean) or 2 (max)");
    return 0;
}

#if EC_FWD
static int ec_forward_impl(const char *name, int b, int n, int k, int cin, int cout, const float *x, int ldx,
                           const int *nn_idx, const float *weights, const float *biases,
                           const float *gamma, const float *beta, int training,
                           const float *decay, float *ema_mean, float *ema_var,
                           int pool_mode, float *pq, float *save_mean, float *save_var,
                           float *out, int ldo, float *tie_count, float *edge_stats,
                           int gemm_bf16, void *workspace, const cloudaae_bn_sync *sync, cloudaae_stream_t stream,
                           void *out16 = nullptr, int ldo16 = 0)
{
    if (int rc = ec_check(name, b, n, k, cin, cout, pool_mode))
        return rc;
    CLOUDAAE_REQUIRE(training || (ema_mean && ema_var), name, "inference needs the EMA statistics");
    CLOUDAAE_REQUIRE(out16 == nullptr || ldo16 >= cout, name, "bfloat16 output rows too short");
    CLOUDAAE_REQUIRE(pool_mode != 2 || tie_count != nullptr, name, "max pool needs the tie_count output");
    hipStream_t s = (hipStream_t)stream;
    const int P = b * n;
    // [P' | Q] = X [W_centre | W_neighbour]: ONE product over the folded kernel (cout is 64 or 128)
    int rc = 0;
    if (!gemm_bf16 && cin == 64 && P % 32 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0) {
        const int grid = P / 32 < 512 ? P / 32 : 512;          // two workgroups per CU, tiles in turn (256 .. 2048: the same time)
        if (cout == 64)
            hipLaunchKernelGGL(ec_pq_stream_kernel<128>, dim3(grid), dim3(256), 0, s, P, x, ldx, weights, cout, pq);
        else
            hipLaunchKernelGGL(ec_pq_stream_kernel<256>, dim3(grid), dim3(256), 0, s, P, x, ldx, weights, cout, pq);
    } else {
        rc = ec_gemm(name, gemm_bf16, 0, 0, P, 2 * cout, cin, x, ldx, weights, cout, pq, 2 * cout, 0, cout, 0, stream);
    }
    if (rc)
        return rc;
    double *partial = (double *)workspace;
    float *scale_shift = (float *)(partial + (size_t)EC_MAX_PARTS * 4 * cout);
    EcArgs a = {};
    a.P = P; a.N = n; a.k = k; a.cout = cout; a.ldpq = 2 * cout;
    a.pq = pq; a.bias = biases; a.nn_idx = nn_idx; a.scale_shift = scale_shift;
    a.save_mean = save_mean; a.save_var = save_var;
    a.out16 = (unsigned short *)out16; a.ldo16 = ldo16;
    float *es = (training && pool_mode == 1) ? edge_stats : nullptr;
    const int cpl = cout / 64, kcap = k <= 10 ? 10 : (k <= 20 ? 20 : 32);
    const int grid = ec_stat_grid(P), agrid = ec_apply_grid(P);
    if (training) {
#define EC_STATS(CPL_, KC_)                                                                                                     \
    do {                                                                                                                         \
        if (k == KC_) hipLaunchKernelGGL((ec_stats_kernel<CPL_, KC_, true>), dim3(grid), dim3(64 * EC_STAT_WAVES), 0, s, a, partial); \
        else hipLaunchKernelGGL((ec_stats_kernel<CPL_, KC_, false>), dim3(grid), dim3(64 * EC_STAT_WAVES), 0, s, a, partial);   \
    } while (0)
        EC_DISPATCH(EC_STATS);
#undef EC_STATS
    }
    const double *sums = partial;
    int fin_parts = grid;
    double count = (double)P * (double)k;
    if (training && sync != nullptr) {      // SyncBN: the edges of every rank's clouds
        if (int rcs = bn_sync_exchange(name, sync, cout, partial, grid, s))
            return rcs;
        sums = sync->buf;
        fin_parts = 1;
        count *= (double)sync->world;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(cout, BN_FIN_CH)), dim3(BN_FIN_THREADS), 0, s, cout, sums, fin_parts,
                       count, training, decay, ema_mean, ema_var, gamma, beta, save_mean, save_var, scale_shift);
    if (pool_mode == 1) {
#define EC_APPLY(CPL_, KC_)                                                                                                     \
    do {                                                                                                                         \
        if (k == KC_ && es != nullptr) hipLaunchKernelGGL((ec_apply_kernel<CPL_, KC_, 1, true, true>), dim3(agrid), dim3(64 * EC_WAVES), 0, s, a, out, ldo, tie_count, es, pq); \
        else if (k == KC_) hipLaunchKernelGGL((ec_apply_kernel<CPL_, KC_, 1, true, false>), dim3(agrid), dim3(64 * EC_WAVES), 0, s, a, out, ldo, tie_count, es, pq); \
        else if (es != nullptr) hipLaunchKernelGGL((ec_apply_kernel<CPL_, KC_, 1, false, true>), dim3(agrid), dim3(64 * EC_WAVES), 0, s, a, out, ldo, tie_count, es, pq); \
        else hipLaunchKernelGGL((ec_apply_kernel<CPL_, KC_, 1, false, false>), dim3(agrid), dim3(64 * EC_WAVES), 0, s, a, out, ldo, tie_count, es, pq);   \
    } while (0)
        EC_DISPATCH(EC_APPLY);
#undef EC_APPLY
    } else {
#define EC_APPLY(CPL_, KC_)                                                                                                     \
    do {                                                                                                                         \
        if (k == KC_) hipLaunchKernelGGL((ec_apply_kernel<CPL_, KC_, 2, true, false>), dim3(agrid), dim3(64 * EC_WAVES), 0, s, a, out, ldo, tie_count, nullptr, pq); \
        else hipLaunchKernelGGL((ec_apply_kernel<CPL_, KC_, 2, false, false>), dim3(agrid), dim3(64 * EC_WAVES), 0, s, a, out, ldo, tie_count, nullptr, pq);   \
    } while (0)
        EC_DISPATCH(EC_APPLY);
#undef EC_APPLY
    }
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_edgeconv_forward(int b, int n, int k, int cin, int cout, const float *x, int ldx,
                                           const int *nn_idx, const float *weights, const float *biases,
                                           const float *gamma, const float *beta, int training,
                                           const float *decay, float *ema_mean, float *ema_var,
                                           int pool_mode, float *pq, float *save_mean, float *save_var,
                                           float *out, int ldo, float *tie_count, float *edge_stats,
                                           int gemm_bf16, void *workspace, cloudaae_stream_t stream)
{
    return ec_forward_impl("cloudaae_edgeconv_forward", b, n, k, cin, cout, x, ldx, nn_idx, weights, biases, gamma, beta,
                           training, decay, ema_mean, ema_var, pool_mode, pq, save_mean, save_var, out, ldo, tie_count,
                           edge_stats, gemm_bf16, workspace, nullptr, stream);
}

CLOUDAAE_API int cloudaae_edgeconv_forward_b16out(int b, int n, int k, int cin, int cout, const float *x, int ldx,
                                                  const int *nn_idx, const float *weights, const float *biases,
                                                  const float *gamma, const float *beta, int training,
                                                  const float *decay, float *ema_mean, float *ema_var,
                                                  int pool_mode, float *pq, float *save_mean, float *save_var,
                                                  float *out, int ldo, float *tie_count, float *edge_stats,
                                                  int gemm_bf16, void *workspace, void *out_bf16, int ldo_bf16,
                                                  cloudaae_stream_t stream)
{
    const char *name = "cloudaae_edgeconv_forward_b16out";
    CLOUDAAE_REQUIRE(out_bf16 != nullptr, name, "null argument");
    return ec_forward_impl(name, b, n, k, cin, cout, x, ldx, nn_idx, weights, biases, gamma, beta, training, decay, ema_mean,
                           ema_var, pool_mode, pq, save_mean, save_var, out, ldo, tie_count, edge_stats, gemm_bf16, workspace,
                           nullptr, stream, out_bf16, ldo_bf16);
}

CLOUDAAE_API int cloudaae_edgeconv_forward_sync(int b, int n, int k, int cin, int cout, const float *x, int ldx,
                                                const int *nn_idx, const float *weights, const float *biases,
                                                const float *gamma, const float *beta, int training,
                                                const float *decay, float *ema_mean, float *ema_var,
                                                int pool_mode, float *pq, float *save_mean, float *save_var,
                                                float *out, int ldo, float *tie_count, float *edge_stats,
                                                int gemm_bf16, void *workspace, const cloudaae_bn_sync *sync,
                                                cloudaae_stream_t stream)
{
    return ec_forward_impl("cloudaae_edgeconv_forward_sync", b, n, k, cin, cout, x, ldx, nn_idx, weights, biases, gamma,
                           beta, training, decay, ema_mean, ema_var, pool_mode, pq, save_mean, save_var, out, ldo,
                           tie_count, edge_stats, gemm_bf16, workspace, sync, stream);
}

#endif   // EC_FWD

#if EC_BWD
static int ec_backward_impl(const char *name, int b, int n, int k, int cin, int cout, const float *x, int ldx,
                            const int *nn_idx, const float *weights, const float *biases,
                            const float *gamma, const float *beta, int training,
                            int pool_mode, const float *pq, const float *save_mean,
                            const float *save_var, const float *out, int ldo,
                            const float *tie_count, const float *dout, int lddo, float *dpq,
                            int *rev_scratch, int rev_ready, float *dx, int lddx, int accumulate_dx,
                            float *dweights, int dweights_zeroed, float *dbiases, float *dgamma,
                            float *dbeta, const float *edge_stats, int gemm_bf16, void *workspace,
                            const cloudaae_bn_sync *sync, cloudaae_stream_t stream, cloudaae_stream_t side_stream)
{
    if (int rc = ec_check(name, b, n, k, cin, cout, pool_mode))
        return rc;
    CLOUDAAE_REQUIRE(dpq && dout && workspace && rev_scratch, name, "null argument");
    CLOUDAAE_REQUIRE(pool_mode != 2 || (out && tie_count), name, "max pool backward needs the forward output and tie count");
    CLOUDAAE_REQUIRE((size_t)n * sizeof(int) <= 150 * 1024, name, "cloud too large for the LDS counting sort");
    hipStream_t s = (hipStream_t)stream;
    // side stream (optional): the reverse neighbour lists depend on nothing this call computes, so they
    // are built there while the statistics pass runs here; the weight gradient, which nothing waits for
    // until the optimiser, follows them there.  The CALLER joins (cloudaae_stream_wait(stream, side)).
    const bool two = side_stream != nullptr && side_stream != stream;
    hipStream_t side = two ? (hipStream_t)side_stream : s;
    const int P = b * n;
    double *partial = (double *)workspace;
    float *scratch = (float *)(partial + (size_t)EC_MAX_PARTS * 4 * cout);
    float *m12 = scratch + 2 * (size_t)cout;
    int *rev_off = rev_scratch, *rev_src = rev_scratch + (size_t)b * (n + 1);
    const int *const idx1[1] = {nn_idx};
    int *const rev1[1] = {rev_scratch};
    if (two && !rev_ready) {
        if (int rc = cloudaae_stream_wait(side_stream, stream))
            return rc;
        if (int rc = ec_launch_revlists(name, 1, b, n, k, idx1, rev1, side))
            return rc;
    }
    EcArgs a = {};
    a.P = P; a.N = n; a.k = k; a.cout = cout; a.ldpq = 2 * cout;
    a.pq = pq; a.bias = biases; a.nn_idx = nn_idx; a.scale_shift = nullptr;   // backward kernels derive it per lane
    a.gamma = gamma; a.beta = beta; a.save_mean = save_mean; a.save_var = save_var; a.dout = dout; a.lddo = lddo;
    a.training = training;
    const int cpl = cout / 64, kcap = k <= 10 ? 10 : (k <= 20 ? 20 : 32);
    const int grid = ec_stat_grid(P), agrid = ec_apply_grid(P);
    if (pool_mode == 1 && edge_stats != nullptr && training) {
        if (cpl == 1)
            hipLaunchKernelGGL(ec_bwd_stats_pool_kernel<1>, dim3(grid), dim3(64 * EC_STAT_WAVES), 0, s, a, edge_stats,
                               partial);
        else
            hipLaunchKernelGGL(ec_bwd_stats_pool_kernel<2>, dim3(grid), dim3(64 * EC_STAT_WAVES), 0, s, a, edge_stats,
                               partial);
    } else if (pool_mode == 1) {
#define EC_BS(CPL_, KC_) hipLaunchKernelGGL((ec_bwd_stats_kernel<CPL_, KC_, 1>), dim3(grid), dim3(64 * EC_STAT_WAVES), 0, s, a, partial)
        EC_DISPATCH(EC_BS);
#undef EC_BS
    } else {
#define EC_BS(CPL_, KC_) hipLaunchKernelGGL((ec_bwd_stats_kernel<CPL_, KC_, 2>), dim3(grid), dim3(64 * EC_STAT_WAVES), 0, s, a, partial)
        EC_DISPATCH(EC_BS);
#undef EC_BS
    }
    const double *gsums = nullptr;
    double gcount = 0.0;
    if (training && sync != nullptr) {      // SyncBN: means of dz and dz * x_hat over every rank's edges
        if (int rcs = bn_sync_exchange(name, sync, cout, partial, grid, s))
            return rcs;
        gsums = sync->buf;
        gcount = (double)P * (double)k * (double)sync->world;
    }
    hipLaunchKernelGGL(ec_bwd_finalize_kernel, dim3(ceil_div(cout, BN_FIN_CH)), dim3(BN_FIN_THREADS), 0, s, cout, partial,
                       partial + (size_t)EC_MAX_PARTS * 2 * cout, grid, (double)P * (double)k, training, gamma,
                       save_var, dgamma, dbeta, dbiases, m12, gsums, gcount);
    if (rev_ready) {
        // (built for all layers at once: cloudaae_edgeconv_revlists)
    } else if (two) {
        if (int rc = cloudaae_stream_wait(stream, side_stream))    // the lists are ready before the apply pass
            return rc;
    } else if (int rc = ec_launch_revlists(name, 1, b, n, k, idx1, rev1, s)) {
        return rc;
    }
    const bool quads = pool_mode == 1 && training && edge_stats != nullptr && (cout == 64 || cout == 128) && lddo % 4 == 0 &&
                       (((uintptr_t)a.pq | (uintptr_t)a.dout | (uintptr_t)dpq | (uintptr_t)edge_stats) & 15) == 0;
    if (quads) {
        const bool one = ec_div_corrections(k) == 1;
#define EC_BA4(COUT_, FIX_) hipLaunchKernelGGL((ec_bwd_apply_mean4_kernel<COUT_, FIX_>), dim3(agrid), dim3(64 * EC_WAVES), 0, s, a, m12, rev_off, rev_src, dpq, edge_stats)
        if (cout == 64 && one) EC_BA4(64, 1);
        else if (cout == 64) EC_BA4(64, 2);
        else if (one) EC_BA4(128, 1);
        else EC_BA4(128, 2);
#undef EC_BA4
    } else if (pool_mode == 1) {
#define EC_BA(CPL_, KC_) hipLaunchKernelGGL((ec_bwd_apply_kernel<CPL_, KC_, 1>), dim3(agrid), dim3(64 * EC_WAVES), 0, s, a, m12, rev_off, rev_src, out, ldo, tie_count, dpq, training ? edge_stats : nullptr)
        EC_DISPATCH(EC_BA);
#undef EC_BA
    } else {
#define EC_BA(CPL_, KC_) hipLaunchKernelGGL((ec_bwd_apply_kernel<CPL_, KC_, 2>), dim3(agrid), dim3(64 * EC_WAVES), 0, s, a, m12, rev_off, rev_src, out, ldo, tie_count, dpq, nullptr)
        EC_DISPATCH(EC_BA);
#undef EC_BA
    }
    CLOUDAAE_CHECK_LAUNCH(name);
    // dX = [dP' | dQ] [W_c | W_n]^T ; [dW_c | dW_n] = X^T [dP' | dQ] : one product each over the folded kernel
    int rc = 0;
    if (dx != nullptr) {
        rc = ec_gemm(name, gemm_bf16, 0, 1, P, cin, 2 * cout, dpq, 2 * cout, weights, cout, dx, lddx, accumulate_dx,
                     cout, 0, stream);
        if (rc)
            return rc;
    }
    if (dweights != nullptr) {
        const int wacc = dweights_zeroed ? 2 : 0;     // 2: the caller cleared dweights already
        if (two)
            if (int rcw = cloudaae_stream_wait(side_stream, stream))   // dpq is complete
                return rcw;
        rc = ec_gemm(name, gemm_bf16, 1, 0, cin, 2 * cout, P, x, ldx, dpq, 2 * cout, dweights, cout, wacc, 0, cout,
                     (cloudaae_stream_t)side);
        if (rc)
            return rc;
    }
    return 0;
}

CLOUDAAE_API int cloudaae_edgeconv_backward(int b, int n, int k, int cin, int cout, const float *x, int ldx,
                                            const int *nn_idx, const float *weights, const float *biases,
                                            const float *gamma, const float *beta, int training,
                                            int pool_mode, const float *pq, const float *save_mean,
                                            const float *save_var, const float *out, int ldo,
                                            const float *tie_count, const float *dout, int lddo, float *dpq,
                                            int *rev_scratch, int rev_ready, float *dx, int lddx, int accumulate_dx,
                                            float *dweights, int dweights_zeroed, float *dbiases, float *dgamma,
                                            float *dbeta, const float *edge_stats, int gemm_bf16, void *workspace,
                                            cloudaae_stream_t stream, cloudaae_stream_t side_stream)
{
    return ec_backward_impl("cloudaae_edgeconv_backward", b, n, k, cin, cout, x, ldx, nn_idx, weights, biases, gamma, beta,
                            training, pool_mode, pq, save_mean, save_var, out, ldo, tie_count, dout, lddo, dpq,
                            rev_scratch, rev_ready, dx, lddx, accumulate_dx, dweights, dweights_zeroed, dbiases, dgamma,
                            dbeta, edge_stats, gemm_bf16, workspace, nullptr, stream, side_stream);
}

CLOUDAAE_API int cloudaae_edgeconv_backward_sync(int b, int n, int k, int cin, int cout, const float *x, int ldx,
                                                 const int *nn_idx, const float *weights, const float *biases,
                                                 const float *gamma, const float *beta, int training,
                                                 int pool_mode, const float *pq, const float *save_mean,
                                                 const float *save_var, const float *out, int ldo,
                                                 const float *tie_count, const float *dout, int lddo, float *dpq,
                                                 int *rev_scratch, int rev_ready, float *dx, int lddx,
                                                 int accumulate_dx, float *dweights, int dweights_zeroed,
                                                 float *dbiases, float *dgamma, float *dbeta, const float *edge_stats,
                                                 int gemm_bf16, void *workspace, const cloudaae_bn_sync *sync,
                                                 cloudaae_stream_t stream, cloudaae_stream_t side_stream)
{
    return ec_backward_impl("cloudaae_edgeconv_backward_sync", b, n, k, cin, cout, x, ldx, nn_idx, weights, biases, gamma,
                            beta, training, pool_mode, pq, save_mean, save_var, out, ldo, tie_count, dout, lddo, dpq,
                            rev_scratch, rev_ready, dx, lddx, accumulate_dx, dweights, dweights_zeroed, dbiases, dgamma,
                            dbeta, edge_stats, gemm_bf16, workspace, sync, stream, side_stream);
}

#endif   // EC_BWD
