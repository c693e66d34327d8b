// fc.hip -- the fully connected stack when the rows are the clouds of one GPU's batch (M <= 32).
//
// tf_util.fully_connected (reference utils/tf_util.py:321-365) is matmul + bias_add [+ batch norm
// + ReLU]; the decoder and the two pose heads (models/pointnet_ycb_23_decoder_4.py:413-455) are nine
// such layers.  With 32 rows the whole batch of a layer fits ONE 32-row MFMA tile, so
//   * the batch statistics of an output column live inside the workgroup that owns the column:
//     forward is ONE launch (product, bias, moments, EMA, normalise, ReLU) instead of two;
//   * backward is ONE launch instead of three or four: a workgroup owns a 128-column slice of the
//     layer's output, derives d(pre-BN) for it in LDS, and every wave then walks 32-row tiles of W:
//     dW[tile, slice] = X[:, tile]^T dY (complete -- the batch is the whole reduction) and
//     dX[:, tile] += dY W[tile, slice]^T (partial over the slice: fp32 atomics into a zeroed buffer).
// These products are bound by streaming W (and writing dW) once, not by the matrix pipe: the lanes
// read W rows as dwordx4 (512 contiguous bytes per half-wave) and the four components feed four
// v_mfma_f32_32x32x2_f32, i.e. lane l of an MFMA column index owns output columns 4l..4l+3.
#include "common.h"
#include "bn_common.h"
#include <stdlib.h>
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FC_M = 32;        // rows of one MFMA tile = the largest batch this path takes
constexpr int FC_TN = 128;      // output columns per workgroup (4 per MFMA column lane)
constexpr int FC_LD = FC_TN + 4;  // LDS row stride (floats), keeps rows 16-byte aligned

// Four consecutive floats at p[0..3], of which the first `valid` exist.  No lane ever branches around a
// load (a predicated load costs a branch each and serialises the batch): an address that does not
// exist is replaced by `safe`, a location that does, and whoever consumes the value ignores or
// zeroes it.  VEC: 16-byte aligned quads that exist whole or not at all.
template <bool VEC>
__device__ __forceinline__ float4v fc_load4(const float *__restrict__ p, int valid, const float *__restrict__ safe)
{
    float4v v;
    if (VEC) {
        v = *reinterpret_cast<const float4v *>(valid > 0 ? p : safe);
    } else {
        v.x = *(valid > 0 ? p : safe);
        v.y = *(valid > 1 ? p + 1 : safe);
        v.z = *(valid > 2 ? p + 2 : safe);
        v.w = *(valid > 3 ? p + 3 : safe);
        v.x = valid > 0 ? v.x : 0.0f;
        v.y = valid > 1 ? v.y : 0.0f;
        v.z = valid > 2 ? v.z : 0.0f;
        v.w = valid > 3 ? v.w : 0.0f;
    }
    return v;
}

__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ---- forward ------------------------------------------------------------------------------------

constexpr int FC_MAX_GROUP = 4;     // layers one launch can take

struct FcFwdArgs {
    int M, K, N, ldx, kslice, atomic, training, relu;
    int block0, tiles, splits, vec;     // this layer's workgroups: block0 .. block0 + tiles * splits - 1
    const float *x, *w, *bias;
    const float *gamma, *beta, *decay;      // gamma == nullptr: no batch norm
    float *ema_mean, *ema_var, *save_mean, *save_var;
    float *y, *out;
    int *tickets;       // a product cut over K that is finished by its last slice: one arrival counter per column tile
    float *partials;    // [tiles][splits][FC_M][FC_TN]: the slices' partial tiles, summed in slice order by the last to arrive
    const float *rowvec;    // no batch norm: y[r][c] += rowvec[r * rowvec_d + c % rowvec_d] (the "+ element_mean" of
    int rowvec_d;           // train_cloudAAE_ycbv.py:232-233 folded into the output layers); NULL: nothing
};

struct FcGroup {        // operands of eight k: lane half h holds k + 4h .. k + 4h + 3
    float4v a;          // X[row][k4 .. k4+3]
    float4v b[4];       // W[k4 + j][4 columns]
};
struct FcSet {          // sixteen k: what a wave keeps in flight behind its MFMAs
    FcGroup g[2];
    bool in[2];         // group lies inside the wave's run of k (else its X operand counts as zero)
};

struct FcFwdGroup {
    int count;
    FcFwdArgs p[FC_MAX_GROUP];
};

constexpr int FC_NW = 4;            // waves per workgroup

// One workgroup = (column tile, K slice) of one layer; the NW waves take contiguous runs of the slice's
// k, their partial 32 x 128 tiles meet in LDS, and the threads then finish one column each.
template <int NW, bool VEC>
__device__ __forceinline__ void fc_fwd_body(const FcFwdArgs &a, int tile_x, int slice, float *tile,
                                            double (*red)[NW / 2][FC_TN], int *last_flag)
{
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r32 = lane & 31, half = lane >> 5;
    const int n0 = tile_x * FC_TN;
    const int kb0 = slice * a.kslice, kb1 = min(a.K, kb0 + a.kslice);
    const int per = ((kb1 - kb0 + NW - 1) / NW + 7) & ~7;      // k per wave, whole groups of eight
    const int kw0 = kb0 + wv * per, kw1 = min(kb1, kw0 + per);
    const int colq = n0 + 4 * r32;
    // rows >= M and columns >= N are computed from existing data and never written
    const float *xrow = a.x + (size_t)min(r32, a.M - 1) * a.ldx;
    const float *wcol = a.w + (VEC ? min(colq, a.N - 4) : colq);
    const int wvalid = VEC ? 4 : a.N - colq;

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[c][r] = 0.0f;

    auto load = [&](FcSet &st, int k) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ka = k + 8 * u + 4 * half;
            // k beyond the run: read the run's first rows again and zero the X operand
            const bool in = k + 8 * u < kw1;
            const int kc = in ? ka : kb0;
            st.g[u].a = fc_load4<VEC>(xrow + kc, VEC ? 4 : kw1 - ka, a.x);
            st.in[u] = in;      // applied where the operand is consumed, not here: a select on the loaded
                                // value would make the wave wait for the load inside the batch
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kr = VEC ? kc + j : min(kc + j, a.K - 1);     // (a zeroed X column pairs with it)
                st.g[u].b[j] = fc_load4<VEC>(wcol + (size_t)kr * a.N, wvalid, a.w);
            }
        }
    };
    auto mma = [&](const FcSet &st) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float aj = st.in[u] ? st.g[u].a[j] : 0.0f;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(aj, st.g[u].b[j][c], acc[c], 0, 0, 0);
            }
    };

    // two operand sets: the loads of one are in flight behind the 32 MFMAs of the other
    if (kw0 < kw1) {
        FcSet s0, s1;
        load(s0, kw0);
        for (int k = kw0; k < kw1; k += 32) {
            // (the barriers keep each batch of ten loads AHEAD of the MFMAs it hides behind; left alone
            // the scheduler sinks every load next to its use and the wave has one load in flight)
            load(s1, k + 16);
            __builtin_amdgcn_sched_barrier(0);
            mma(s0);
            __builtin_amdgcn_sched_barrier(0);
            load(s0, k + 32);
            __builtin_amdgcn_sched_barrier(0);
            mma(s1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    float *mine = tile + (size_t)wv * FC_M * FC_LD;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float4v v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        *reinterpret_cast<float4v *>(mine + mfma_row(r, half) * FC_LD + 4 * r32) = v;
    }
    __syncthreads();

    constexpr int RG = NW / 2, RP = FC_M / RG;      // row groups, rows per thread
    const int col = threadIdx.x & (FC_TN - 1), rg = threadIdx.x >> 7;
    const int c = n0 + col;
    const bool ok = c < a.N;
    // the bias joins once: with the only slice, with slice 0 of a plain cut product, or (batch norm
    // over a cut product) when the last slice to arrive reads the finished sums back
    const float bias = (a.bias != nullptr && ok) ? a.bias[c] : 0.0f;
    const float bv = (!a.atomic || (a.tickets == nullptr && a.partials == nullptr && slice == 0)) ? bias : 0.0f;
    float v[RP];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int row = rg + RG * i;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w)
            s += tile[((size_t)w * FC_M + row) * FC_LD + col];
        v[i] = s + bv;
    }
    if (a.atomic && a.partials != nullptr) {
        // One K slice of several, combined in a FIXED order (bit-reproducible from run to run, whatever order the
        // slices finish in): every slice publishes its partial tile with agent-scope stores (write-through to
        // where all XCDs meet), every WAVE waits until its stores are acknowledged (s_waitcnt vmcnt(0); s_barrier
        // alone does not drain the counter), then a ticket is taken; the workgroup that draws the last one reads
        // all partial tiles back with agent-scope loads and sums them in slice order 0, 1, 2, ...  No cache
        // write-back or invalidate is involved (a __threadfence() here costs more than the whole product)
        // because no ordinary store takes part.  The counter returns to zero for the next launch.
        // (tests/test_capi_symbols.py checks the emitted ISA for the wait in front of the barrier.)
        float *slot = a.partials + (size_t)(tile_x * a.splits + slice) * (FC_M * FC_TN);
#pragma unroll
        for (int i = 0; i < RP; ++i)
            if (rg + RG * i < a.M)
                __hip_atomic_store(&slot[(rg + RG * i) * FC_TN + col], v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = __hip_atomic_fetch_add(&a.tickets[tile_x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *last_flag = t == a.splits - 1;
            if (t == a.splits - 1)
                __hip_atomic_store(&a.tickets[tile_x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!*last_flag)
            return;
        // thread t sums element pairs t, t + 256, ... of the 32 x 128 tile (8-byte loads, a row per wave and
        // instruction); four slices' loads are in flight together; the sums meet the column threads in LDS
        constexpr int PAIRS = FC_M * FC_TN / 2 / (NW * 64);
        const unsigned long long *base =
            reinterpret_cast<const unsigned long long *>(a.partials + (size_t)tile_x * a.splits * (FC_M * FC_TN));
        float2v sum[PAIRS];
#pragma unroll
        for (int q = 0; q < PAIRS; ++q)
            sum[q] = float2v{0.0f, 0.0f};
        constexpr int INFL = 4;     // (eight would need 280 registers: one workgroup per CU instead of two)
        for (int s0 = 0; s0 < a.splits; s0 += INFL) {
            unsigned long long raw[INFL][PAIRS];
#pragma unroll
            for (int u = 0; u < INFL; ++u) {
                const int s = min(s0 + u, a.splits - 1);        // (past the end: the last slice again, not added)
#pragma unroll
                for (int q = 0; q < PAIRS; ++q) {
                    const int e = (int)threadIdx.x + NW * 64 * q;
                    const int row = min(e / (FC_TN / 2), a.M - 1);      // rows >= M were never published
                    raw[u][q] = __hip_atomic_load(&base[((size_t)s * FC_M + row) * (FC_TN / 2) + e % (FC_TN / 2)],
                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int u = 0; u < INFL; ++u)
                if (s0 + u < a.splits)
#pragma unroll
                    for (int q = 0; q < PAIRS; ++q) {
                        sum[q].x += __uint_as_float((unsigned)raw[u][q]);
                        sum[q].y += __uint_as_float((unsigned)(raw[u][q] >> 32));
                    }
        }
        __syncthreads();        // (the waves' partial tiles in LDS have been consumed)
#pragma unroll
        for (int q = 0; q < PAIRS; ++q) {
            const int e = (int)threadIdx.x + NW * 64 * q;
            *reinterpret_cast<float2v *>(tile + (e / (FC_TN / 2)) * FC_LD + 2 * (e % (FC_TN / 2))) = sum[q];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RP; ++i)
            v[i] = tile[(rg + RG * i) * FC_LD + col] + bias;
    } else if (a.atomic) {     // one K slice of several, added with fp32 atomics: the output was cleared by the caller
        if (ok)
#pragma unroll
            for (int i = 0; i < RP; ++i)
                if (rg + RG * i < a.M)
                    __hip_atomic_fetch_add(&a.y[(size_t)(rg + RG * i) * a.N + c], v[i], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
        if (a.tickets == nullptr)
            return;
        // Batch norm needs the whole column: the same ticket protocol as above, with the sums added in arrival
        // order by agent-scope atomics (performed at the memory side, where all XCDs meet) and read back by the
        // last slice to arrive.  (Callers that pass no `partials` workspace.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = __hip_atomic_fetch_add(&a.tickets[tile_x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *last_flag = t == a.splits - 1;
            if (t == a.splits - 1)
                __hip_atomic_store(&a.tickets[tile_x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!*last_flag)
            return;
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const int row = min(rg + RG * i, a.M - 1);
            v[i] = __hip_atomic_load(&a.y[(size_t)row * a.N + min(c, a.N - 1)], __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) + bias;
        }
    }
    if (a.gamma == nullptr) {
        if (ok) {
            const int cd = a.rowvec != nullptr ? c % a.rowvec_d : 0;
#pragma unroll
            for (int i = 0; i < RP; ++i)
                if (rg + RG * i < a.M) {
                    float out = v[i];
                    if (a.rowvec != nullptr)
                        out = out + a.rowvec[(size_t)(rg + RG * i) * a.rowvec_d + cd];
                    a.y[(size_t)(rg + RG * i) * a.N + c] = out;
                }
        }
        return;
    }

    // batch norm of the column (same arithmetic as bn_small_fwd_kernel: fp64 sums, fp32 formulas)
    float mean, var;
    if (a.training) {
        double s = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < RP; ++i)
            if (rg + RG * i < a.M) {
                s += (double)v[i];
                s2 += (double)v[i] * (double)v[i];
            }
        red[0][rg][col] = s;
        red[1][rg][col] = s2;
        __syncthreads();
        double ts = 0.0, ts2 = 0.0;
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            ts += red[0][g][col];
            ts2 += red[1][g][col];
        }
        const double mu = ts / (double)a.M;
        double vv = ts2 / (double)a.M - mu * mu;
        vv = vv > 0.0 ? vv : 0.0;
        mean = (float)mu;
        var = (float)vv;
        if (ok && rg == 0 && a.ema_mean != nullptr) {
            const float om = 1.0f - a.decay[0];
            a.ema_mean[c] = a.ema_mean[c] - (a.ema_mean[c] - mean) * om;
            a.ema_var[c] = a.ema_var[c] - (a.ema_var[c] - var) * om;
        }
    } else {
        mean = ok ? a.ema_mean[c] : 0.0f;
        var = ok ? a.ema_var[c] : 1.0f;
    }
    if (!ok)
        return;
    if (rg == 0) {
        a.save_mean[c] = mean;
        a.save_var[c] = var;
    }
    const float inv = a.gamma[c] * bn_rsqrt(var + BN_EPS);
    const float sh = a.beta[c] - mean * inv;
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int row = rg + RG * i;
        if (row < a.M) {
            a.y[(size_t)row * a.N + c] = v[i];
            float z = v[i] * inv + sh;
            if (a.relu)
                z = fmaxf(z, 0.0f);
            a.out[(size_t)row * a.N + c] = z;
        }
    }
}

// which layer of the group a workgroup belongs to (block ranges are ascending)
template <typename G>
__device__ __forceinline__ int fc_group_member(const G &g)
{
    int p = 0;
    for (int i = 1; i < g.count; ++i)
        if ((int)blockIdx.x >= g.p[i].block0)
            p = i;
    return p;
}

__global__ __launch_bounds__(FC_NW * 64) void fc_fwd_kernel(FcFwdGroup g)
{
    __shared__ float4v tile4[FC_NW * FC_M * (FC_LD / 4)];
    __shared__ double red[2][FC_NW / 2][FC_TN];
    __shared__ int last;
    const FcFwdArgs a = g.p[fc_group_member(g)];
    const int local = (int)blockIdx.x - a.block0;
    const int tile_x = local % a.tiles, slice = local / a.tiles;
    if (a.vec)
        fc_fwd_body<FC_NW, true>(a, tile_x, slice, reinterpret_cast<float *>(tile4), red, &last);
    else
        fc_fwd_body<FC_NW, false>(a, tile_x, slice, reinterpret_cast<float *>(tile4), red, &last);
}

// ---- backward -----------------------------------------------------------------------------------

struct FcBwdArgs {
    int M, K, N, ldx, lddo, lddx, tiles_per_block, acc_dw, acc_pg, training, relu;
    int block0, slices, parts, vec;     // this layer's workgroups: block0 .. block0 + slices * groups - 1
    const float *x, *w, *y, *gamma, *beta, *save_mean, *save_var, *dout;
    float *dx, *dw, *dgamma, *dbeta, *dbias;
};

// dW[kr0.., NC columns per lane from column c0] of one 32-row tile of W: 16 MFMA steps over the batch
template <bool VEC, int NC>
__device__ __forceinline__ void fc_bwd_dw(const FcBwdArgs &a, const float *dyl, int kr0, int c0, int lcol, int r32,
                                          int half)
{
    const int krc = min(kr0 + r32, a.K - 1);    // rows of W past K: an existing one, result not written
    float xa[16];
#pragma unroll
    for (int s = 0; s < 16; ++s)                // rows past M meet zero rows of dY
        xa[s] = a.x[(size_t)min(2 * s + half, a.M - 1) * a.ldx + krc];
    f32x16 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[c][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        float bq[NC];
        const float *src = dyl + (2 * s + half) * FC_LD + lcol;
        if (NC == 4) {
            const float4v t = *reinterpret_cast<const float4v *>(src);
            bq[0] = t.x; bq[1] = t.y; bq[NC - 2] = t.z; bq[NC - 1] = t.w;
        } else {
            const float2v t = *reinterpret_cast<const float2v *>(src);
            bq[0] = t.x; bq[1] = t.y;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c)
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[s], bq[c], acc[c], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = kr0 + mfma_row(r, half);
        if (row >= a.K)
            continue;
        float *dst = a.dw + (size_t)row * a.N + c0;
        if (VEC) {      // N % 4 == 0: a lane's NC columns exist together
            if (c0 < a.N) {
                if (NC == 4) {
                    float4v v = {acc[0][r], acc[1][r], acc[NC - 2][r], acc[NC - 1][r]};
                    if (a.acc_dw)
                        v += *reinterpret_cast<const float4v *>(dst);
                    *reinterpret_cast<float4v *>(dst) = v;
                } else {
                    float2v v = {acc[0][r], acc[1][r]};
                    if (a.acc_dw)
                        v += *reinterpret_cast<const float2v *>(dst);
                    *reinterpret_cast<float2v *>(dst) = v;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (c0 + c < a.N)
                    dst[c] = (a.acc_dw ? dst[c] : 0.0f) + acc[c][r];
        }
    }
}

// dX[:, kr0..kr0+31] += dY[:, 8 q0 .. 8 (q0+NQ)) W[tile, same columns]^T: 4 NQ MFMA steps
template <bool VEC, int NQ>
__device__ __forceinline__ void fc_bwd_dx(const FcBwdArgs &a, const float *dyl, int kr0, int n0, int q0, int r32,
                                          int half)
{
    const int kr = kr0 + r32;
    const float *wrow = a.w + (size_t)min(kr, a.K - 1) * a.N;
    float4v wq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        // columns past N meet zeros of dY (VEC: the row's last quad again; else zero-filled)
        const int cq = n0 + 4 * half + 8 * (q0 + q);
        wq[q] = fc_load4<VEC>(wrow + (VEC ? min(cq, a.N - 4) : cq), VEC ? 4 : a.N - cq, a.w);
    }
    f32x16 d;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        d[r] = 0.0f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4v aq = *reinterpret_cast<const float4v *>(dyl + r32 * FC_LD + 8 * (q0 + q) + 4 * half);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[j], wq[q][j], d, 0, 0, 0);
    }
    if (kr < a.K)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = mfma_row(r, half);
            if (b < a.M)
                atomicAdd(&a.dx[(size_t)b * a.lddx + kr], d[r]);
        }
}

// The same product for the wide output layer (VEC only), W read the way it lies in memory.  Above, a lane
// follows ONE row of W, so a load instruction touches 32 rows x 32 bytes: every 128-byte line is
// requested by four instructions and crosses the L2 -> L1 path four times (2.1 TB/s on the 50 MB of the
// output layer).  Here the wave reads 64-column halves of the tile row by row (half-wave = 256
// contiguous bytes), parks them in its own LDS patch and takes the MFMA operand (lane = row) from
// there.  Both halves are requested before the first is consumed.
constexpr int FC_WLD = 64 + 4;      // LDS row stride of the half tile (floats)

__device__ __forceinline__ void fc_bwd_dx_staged(const FcBwdArgs &a, const float *dyl, float *patch, int kr0,
                                                 int n0, int r32, int half, int lane)
{
    // load h, i: rows 4 i + (lane >> 4), columns 64 h + 4 (lane & 15)
    const int lrow = lane >> 4, lcol = 4 * (lane & 15);
    float4v wq[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = min(kr0 + 4 * i + lrow, a.K - 1);       // past K: an existing row, never written
            const int col = min(n0 + 64 * h + lcol, a.N - 4);       // past N: meets zeros of dY
            wq[h][i] = *reinterpret_cast<const float4v *>(a.w + (size_t)row * a.N + col);
        }
    f32x16 d;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        d[r] = 0.0f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            *reinterpret_cast<float4v *>(patch + (4 * i + lrow) * FC_WLD + lcol) = wq[h][i];
        // (wave-private patch: program order and the LDS counter are all the synchronisation needed)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4v aq = *reinterpret_cast<const float4v *>(dyl + r32 * FC_LD + 64 * h + 8 * q + 4 * half);
            const float4v bq = *reinterpret_cast<const float4v *>(patch + r32 * FC_WLD + 8 * q + 4 * half);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[j], bq[j], d, 0, 0, 0);
        }
    }
    const int kr = kr0 + r32;
    if (kr < a.K)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = mfma_row(r, half);
            if (b < a.M)
                atomicAdd(&a.dx[(size_t)b * a.lddx + kr], d[r]);
        }
}

struct FcBwdGroup {
    int count;
    FcBwdArgs p[FC_MAX_GROUP];
};

// One workgroup (four waves) = (128-column slice of the output, group of 32-row tiles of W).
// parts = 1: a wave does both products of a tile (fewest atomics: the wide output layer);
// parts = 4: the four waves share a tile -- dW columns 0..63 | 64..127, dX over columns 0..63 | 64..127 --
//            for the layers whose whole backward is a few hundred tiles (32 MFMAs per wave, one tile per
//            workgroup, every workgroup resident at once).
template <bool VEC>
__device__ __forceinline__ void fc_bwd_body(const FcBwdArgs &a, int slice_x, int group_y, float *dyl,
                                            double (*red)[2][FC_TN], float *patches)
{
    constexpr int HF = 2, RP = FC_M / HF;       // row groups of the first phase, rows per thread
    const int n0 = slice_x * FC_TN;
    {   // d(pre-BN output) of this slice (bn_small_bwd_kernel's arithmetic)
        const int col = threadIdx.x & (FC_TN - 1), hf = threadIdx.x >> 7;
        const int c = n0 + col;
        const bool ok = c < a.N;
        const bool bn = a.gamma != nullptr;
        const bool writer = ok && hf == 0 && group_y == 0;
        const int cc = min(c, a.N - 1);     // columns past N: an existing one, results zeroed or not written
        float mean = 0.0f, rstd = 1.0f, g = 0.0f, inv = 0.0f, sh = 0.0f;
        if (bn) {
            mean = a.save_mean[cc];
            const float var = a.save_var[cc];
            g = a.gamma[cc];
            const float b = a.beta[cc];
            rstd = bn_rsqrt(var + BN_EPS);
            inv = g * rstd;
            sh = b - mean * inv;
        }
        float xh[RP], dz[RP], yv[RP];
        // all loads first, from addresses that exist (no branch around any of them)
#pragma unroll
        for (int i = 0; i < RP; ++i)
            dz[i] = a.dout[(size_t)min(hf * RP + i, a.M - 1) * a.lddo + cc];
#pragma unroll
        for (int i = 0; i < RP; ++i)
            yv[i] = 0.0f;
        if (bn)
#pragma unroll
            for (int i = 0; i < RP; ++i)
                yv[i] = a.y[(size_t)min(hf * RP + i, a.M - 1) * a.N + cc];
        double s = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const int r = hf * RP + i;
            const bool in = ok && r < a.M;
            float d = in ? dz[i] : 0.0f;
            xh[i] = 0.0f;
            if (bn) {
                const float v = in ? yv[i] : 0.0f;
                float z = v * inv + sh;
                if (a.relu)
                    z = fmaxf(z, 0.0f);
                if (a.relu && !(z > 0.0f))
                    d = 0.0f;
                xh[i] = (v - mean) * rstd;
            }
            dz[i] = d;
            if (in) {
                s += (double)d;
                s2 += (double)d * (double)xh[i];
            }
        }
        red[0][hf][col] = s;
        red[1][hf][col] = s2;
        __syncthreads();
        double ts = 0.0, ts2 = 0.0;
#pragma unroll
        for (int h = 0; h < HF; ++h) {
            ts += red[0][h][col];
            ts2 += red[1][h][col];
        }
        if (!bn) {
            if (writer && a.dbias != nullptr)
                a.dbias[c] = (a.acc_pg ? a.dbias[c] : 0.0f) + (float)ts;
#pragma unroll
            for (int i = 0; i < RP; ++i)
                dyl[(hf * RP + i) * FC_LD + col] = dz[i];
        } else {
            if (writer) {
                if (a.dbeta != nullptr)
                    a.dbeta[c] = (a.acc_pg ? a.dbeta[c] : 0.0f) + (float)ts;
                if (a.dgamma != nullptr)
                    a.dgamma[c] = (a.acc_pg ? a.dgamma[c] : 0.0f) + (float)ts2;
            }
            const float m1 = a.training ? (float)(ts / (double)a.M) : 0.0f;
            const float m2 = a.training ? (float)(ts2 / (double)a.M) : 0.0f;
            const float gr = g * rstd;
            double sdy = 0.0;
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                const int r = hf * RP + i;
                float v = 0.0f;
                if (ok && r < a.M) {
                    v = gr * ((dz[i] - m1) - xh[i] * m2);
                    sdy += (double)v;
                }
                dyl[r * FC_LD + col] = v;
            }
            if (a.dbias != nullptr && group_y == 0) {   // uniform per workgroup
                __syncthreads();
                red[0][hf][col] = sdy;
                __syncthreads();
                if (writer) {
                    double t = 0.0;
#pragma unroll
                    for (int h = 0; h < HF; ++h)
                        t += red[0][h][col];
                    a.dbias[c] = (a.acc_pg ? a.dbias[c] : 0.0f) + (float)t;
                }
            }
        }
        __syncthreads();
    }

    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r32 = lane & 31, half = lane >> 5;
    const int ktiles = (a.K + 31) / 32;
    const int t_end = min(ktiles, (group_y + 1) * a.tiles_per_block);
    if (a.parts == 1) {
        for (int t = group_y * a.tiles_per_block + wv; t < t_end; t += 4) {
            if (a.dw != nullptr)
                fc_bwd_dw<VEC, 4>(a, dyl, t * 32, n0 + 4 * r32, 4 * r32, r32, half);
            if (a.dx != nullptr) {
                if (VEC)
                    fc_bwd_dx_staged(a, dyl, patches + wv * (FC_M * FC_WLD), t * 32, n0, r32, half, lane);
                else
                    fc_bwd_dx<VEC, 16>(a, dyl, t * 32, n0, 0, r32, half);
            }
        }
    } else {
        const int side = wv & 1;
        for (int t = group_y * a.tiles_per_block; t < t_end; ++t) {
            if (wv < 2) {
                if (a.dw != nullptr)
                    fc_bwd_dw<VEC, 2>(a, dyl, t * 32, n0 + 64 * side + 2 * r32, 64 * side + 2 * r32, r32, half);
            } else if (a.dx != nullptr) {
                fc_bwd_dx<VEC, 8>(a, dyl, t * 32, n0, 8 * side, r32, half);
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void fc_bwd_kernel(FcBwdGroup g)
{
    __shared__ float4v dy4[FC_M * (FC_LD / 4)];
    __shared__ double red[2][2][FC_TN];
    __shared__ float4v patch4[4 * FC_M * (FC_WLD / 4)];
    const FcBwdArgs a = g.p[fc_group_member(g)];
    const int local = (int)blockIdx.x - a.block0;
    const int slice_x = local % a.slices, group_y = local / a.slices;
    if (a.vec)
        fc_bwd_body<true>(a, slice_x, group_y, reinterpret_cast<float *>(dy4), red, reinterpret_cast<float *>(patch4));
    else
        fc_bwd_body<false>(a, slice_x, group_y, reinterpret_cast<float *>(dy4), red, reinterpret_cast<float *>(patch4));
}

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }


// Column tiles and K slices of one forward layer.  Four waves per workgroup, each with at least sixteen k; K is
// cut into slices until the chip is covered.  These products are short chains of load -> MFMA: what they need
// is every load of the layer in flight at once, i.e. many workgroups.  
static void fc_fwd_plan(int K, int N, bool bn, bool whole_k, int &tiles, int &splits, int &kslice)
{
    const int want_bn = CLOUDAAE_KNOB("CLOUDAAE_FC_FWD_BLOCKS", 128), want_plain = CLOUDAAE_KNOB("CLOUDAAE_FC_FWD_BLOCKS", 192);
    const int forced = CLOUDAAE_KNOB("CLOUDAAE_FC_FWD_SPLITS", 0);
    tiles = ceil_div(N, FC_TN);
    splits = (bn ? want_bn : want_plain) / tiles;
    const int most = K / (16 * FC_NW);
    splits = splits > most ? most : splits;
    splits = splits < 1 ? 1 : splits;
    if (forced)
        splits = forced;
    if (whole_k)
        splits = 1;
    kslice = ceil_div(ceil_div(K, splits), 8) * 8;
    splits = ceil_div(K, kslice);
}

} // namespace cloudaae

using namespace cloudaae;

CLOUDAAE_API int cloudaae_fc_max_rows(void) { return FC_M; }
CLOUDAAE_API int cloudaae_fc_max_group(void) { return FC_MAX_GROUP; }
CLOUDAAE_API int cloudaae_fc_forward_tickets(int N) { return N > 0 ? ceil_div(N, FC_TN) : 0; }
CLOUDAAE_API long long cloudaae_fc_forward_partials(int K, int N, int batch_norm)
{
    if (K <= 0 || N <= 0)
        return 0;
    int tiles, splits, kslice;
    fc_fwd_plan(K, N, batch_norm != 0, false, tiles, splits, kslice);
    return splits > 1 ? (long long)tiles * splits * FC_M * FC_TN : 0;
}

CLOUDAAE_API int cloudaae_fc_forward_group(int M, int count, const cloudaae_fc_layer *layers, int training,
                                           const float *decay, int y_zeroed, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_fc_forward_group";
    CLOUDAAE_REQUIRE(M > 0 && M <= FC_M, name, "bad size (rows must be <= 32)");
    CLOUDAAE_REQUIRE(count > 0 && count <= FC_MAX_GROUP && layers, name, "1 to 4 layers per call");
    hipStream_t s = (hipStream_t)stream;
    FcFwdGroup g;
    g.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const cloudaae_fc_layer &l = layers[i];
        CLOUDAAE_REQUIRE(l.K > 0 && l.N > 0 && l.ldx >= l.K && l.x && l.w && l.y, name, "bad layer");
        const bool bn = l.gamma != nullptr;
        if (bn) {
            CLOUDAAE_REQUIRE(l.beta && l.save_mean && l.save_var && l.out, name,
                             "batch norm needs beta, saved moments and out");
            CLOUDAAE_REQUIRE(training || (l.ema_mean && l.ema_var), name, "inference needs the EMA statistics");
            CLOUDAAE_REQUIRE(!training || !l.ema_mean || decay, name, "EMA update needs the decay scalar");
        }
        // a layer with batch norm can only be cut over K when the caller provides the arrival counters
        int tiles, splits, kslice;
        fc_fwd_plan(l.K, l.N, bn, bn && l.tickets == nullptr, tiles, splits, kslice);
        FcFwdArgs &a = g.p[i];
        a.M = M; a.K = l.K; a.N = l.N; a.ldx = l.ldx; a.kslice = kslice; a.atomic = splits > 1;
        a.training = training; a.relu = l.relu;
        a.block0 = blocks; a.tiles = tiles; a.splits = splits;
        a.vec = l.K % 8 == 0 && l.ldx % 4 == 0 && l.N % 4 == 0 && aligned16(l.x) && aligned16(l.w);
        a.x = l.x; a.w = l.w; a.bias = l.bias; a.gamma = l.gamma; a.beta = l.beta; a.decay = decay;
        a.ema_mean = l.ema_mean; a.ema_var = l.ema_var; a.save_mean = l.save_mean; a.save_var = l.save_var;
        a.y = l.y; a.out = l.out;
        CLOUDAAE_REQUIRE(l.out_rowvec == nullptr || (!bn && l.out_rowvec_d > 0), name,
                         "a row vector can only be added to the output of a layer without batch norm");
        a.rowvec = l.out_rowvec; a.rowvec_d = l.out_rowvec_d;
        // cut over K: with the partial-tile workspace the slices are summed in a fixed order by the last one to
        // arrive (y is plainly stored); without it they add into y with atomics (y cleared first)
        a.partials = (a.atomic && l.tickets != nullptr) ? l.partials : nullptr;
        // (the cut is derived again at every launch, also from development knobs: a buffer sized by an earlier query
        //  must still cover it)
        CLOUDAAE_REQUIRE(a.partials == nullptr || l.partials_floats >= (long long)tiles * splits * FC_M * FC_TN, name,
                         "partials_floats is smaller than this launch's partial tiles (cloudaae_fc_forward_partials; did a "
                         "split knob change since the query?)");
        a.tickets = (a.atomic && (bn || a.partials != nullptr)) ? l.tickets : nullptr;
        CLOUDAAE_REQUIRE(a.rowvec == nullptr || !a.atomic || a.partials != nullptr, name,
                         "adding a row vector to a product cut over K needs the tickets and the partial-tile scratch");
        if (a.atomic && a.partials == nullptr && !y_zeroed)
            CLOUDAAE_CHECK_HIP(hipMemsetAsync(l.y, 0, sizeof(float) * (size_t)M * l.N, s), name);
        blocks += tiles * splits;
    }
    hipLaunchKernelGGL(fc_fwd_kernel, dim3(blocks), dim3(FC_NW * 64), 0, s, g);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_fc_backward_group(int M, int count, const cloudaae_fc_layer *layers, int training,
                                            cloudaae_stream_t stream)
{
    const char *name = "cloudaae_fc_backward_group";
    CLOUDAAE_REQUIRE(M > 0 && M <= FC_M, name, "bad size (rows must be <= 32)");
    CLOUDAAE_REQUIRE(count > 0 && count <= FC_MAX_GROUP && layers, name, "1 to 4 layers per call");
    hipStream_t s = (hipStream_t)stream;
    FcBwdGroup g;
    g.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const cloudaae_fc_layer &l = layers[i];
        CLOUDAAE_REQUIRE(l.K > 0 && l.N > 0 && l.ldx >= l.K && l.lddo >= l.N && l.x && l.w && l.dout, name,
                         "bad layer");
        CLOUDAAE_REQUIRE(l.dx == nullptr || l.lddx >= l.K, name, "bad dx stride");
        if (l.gamma != nullptr)
            CLOUDAAE_REQUIRE(l.y && l.beta && l.save_mean && l.save_var, name,
                             "batch norm needs y, beta and the saved moments");
        const int slices = ceil_div(l.N, FC_TN), ktiles = ceil_div(l.K, 32);
        // Few tiles (every layer but the wide output one): one tile per workgroup, its four waves share
        // it, everything resident at once.  Many tiles: a wave per tile, workgroups for the resident set.
        const bool fine = (long long)slices * ktiles <= CLOUDAAE_KNOB("CLOUDAAE_FC_BWD_FINE", 1024);
        int by;
        if (fine) {
            by = ktiles;
        } else {
            const int want = CLOUDAAE_KNOB("CLOUDAAE_FC_BWD_BLOCKS", 384);
            by = ceil_div(want, slices);
            const int most = ceil_div(ktiles, 4);
            by = by > most ? most : by;
            by = by < 1 ? 1 : by;
        }
        const int tpb = ceil_div(ktiles, by);
        by = ceil_div(ktiles, tpb);
        FcBwdArgs &a = g.p[i];
        a.M = M; a.K = l.K; a.N = l.N; a.ldx = l.ldx; a.lddo = l.lddo; a.lddx = l.lddx; a.tiles_per_block = tpb;
        a.acc_dw = l.accumulate_dw; a.acc_pg = l.accumulate_param_grads; a.training = training; a.relu = l.relu;
        a.block0 = blocks; a.slices = slices; a.parts = fine ? 4 : 1;
        a.vec = l.N % 4 == 0 && aligned16(l.w) && (l.dw == nullptr || aligned16(l.dw));
        a.x = l.x; a.w = l.w; a.y = l.y; a.gamma = l.gamma; a.beta = l.beta; a.save_mean = l.save_mean;
        a.save_var = l.save_var; a.dout = l.dout; a.dx = l.dx; a.dw = l.dw; a.dgamma = l.dgamma;
        a.dbeta = l.dbeta; a.dbias = l.dbias;
        blocks += slices * by;
    }
    hipLaunchKernelGGL(fc_bwd_kernel, dim3(blocks), dim3(256), 0, s, g);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_fc_forward(int M, int K, int N, const float *x, int ldx, const float *w,
                                     const float *bias, const float *gamma, const float *beta, int training,
                                     const float *decay, float *ema_mean, float *ema_var, float *save_mean,
                                     float *save_var, int relu, float *y, float *out, int y_zeroed, int *tickets,
                                     float *partials, long long partials_floats, cloudaae_stream_t stream)
{
    cloudaae_fc_layer l = {};
    l.K = K; l.N = N; l.x = x; l.ldx = ldx; l.w = w; l.bias = bias; l.gamma = gamma; l.beta = beta;
    l.ema_mean = ema_mean; l.ema_var = ema_var; l.save_mean = save_mean; l.save_var = save_var; l.relu = relu;
    l.y = y; l.out = out; l.tickets = tickets; l.partials = partials; l.partials_floats = partials_floats;
    return cloudaae_fc_forward_group(M, 1, &l, training, decay, y_zeroed, stream);
}

CLOUDAAE_API int cloudaae_fc_backward(int M, int K, int N, const float *x, int ldx, const float *w, const float *y,
                                      const float *gamma, const float *beta, const float *save_mean,
                                      const float *save_var, int training, int relu, const float *dout, int lddo,
                                      float *dx, int lddx, float *dw, int accumulate_dw, float *dgamma,
                                      float *dbeta, float *dbias, int accumulate_param_grads,
                                      cloudaae_stream_t stream)
{
    cloudaae_fc_layer l = {};
    l.K = K; l.N = N; l.x = x; l.ldx = ldx; l.w = w; l.gamma = gamma; l.beta = beta;
    l.save_mean = (float *)save_mean; l.save_var = (float *)save_var; l.relu = relu; l.y = (float *)y;
    l.dout = dout; l.lddo = lddo; l.dx = dx; l.lddx = lddx; l.dw = dw; l.accumulate_dw = accumulate_dw;
    l.dgamma = dgamma; l.dbeta = dbeta; l.dbias = dbias; l.accumulate_param_grads = accumulate_param_grads;
    return cloudaae_fc_backward_group(M, 1, &l, training, stream);
}
