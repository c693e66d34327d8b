// fc.hip -- the fully connected stack when the rows are the clouds of one GPU's batch (M <= 128).
//
// tf_util.fully_connected (reference utils/tf_util.py:321-365) is matmul + bias_add [+ batch norm
// + ReLU]; the decoder and the two pose heads (models/pointnet_ycb_23_decoder_4.py:413-455) are nine
// such layers whose rows are the clouds of the batch: 32 rows (BASELINE configs[1]) are ONE 32-row MFMA
// tile, 128 rows (the per-GPU shape of configs[3]) are four.
//   * forward is ONE launch per depth (product, bias, moments, EMA, normalise, ReLU): a workgroup is
//     (column tile, row tile, K slice); slices -- and, under batch norm, the row tiles of a column tile --
//     publish their partial tiles, take a ticket, and the last to arrive sums them in a FIXED order and
//     finishes the column tile (the batch statistics of a column never leave that workgroup);
//   * backward is ONE launch per depth: a workgroup owns a 128-column slice of the layer's output,
//     derives d(pre-BN) for all rows of it in LDS, and its waves then walk 32-row tiles of W:
//     dW[tile, slice] = X[:, tile]^T dY (complete -- the batch is the whole reduction) and
//     dX[:, tile] += dY W[tile, slice]^T (partial over the slice: fp32 atomics into a zeroed buffer).
// At 32 rows these products are bound by streaming W (and writing dW) once; at 128 rows by the fp32
// matrix pipe (2 M K N flops at 157 TFLOP/s: 20 us forward for the 1024 -> 12288 layer).  The lanes
// read W rows as dwordx4 / dwordx2 (512 / 256 contiguous bytes per half-wave) and the components feed
// one v_mfma_f32_32x32x2_f32 each, i.e. lane l of an MFMA column index owns output columns CQ l .. CQ l + CQ - 1.
#include "common.h"
#include "bn_common.h"
#include <stdlib.h>
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FC_ROWS = 32;     // rows of one MFMA tile
constexpr int FC_MAX_RT = 4;    // row tiles of a layer: batches of up to 128 clouds
constexpr int FC_TN = 128;      // backward: output columns per workgroup (4 per MFMA column lane)
constexpr int FC_LD = FC_TN + 4;  // LDS row stride (floats), keeps rows 16-byte aligned

// Development builds only (make prof: -DCLOUDAAE_FC_PROFILE, a second library that tools/dev/fc_phases.py loads): thread 0 of
// every workgroup leaves the 100 MHz wall clock at a few points of the kernel.
#ifdef CLOUDAAE_FC_PROFILE
__device__ unsigned long long g_fc_prof[8192 * 8];
#define FC_STAMP(slot)                                                         \
    do {                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 8192)                             \
            g_fc_prof[blockIdx.x * 8 + (slot)] = wall_clock64();               \
    } while (0)
#else
#define FC_STAMP(slot) do { } while (0)
#endif

template <int CQ> struct fcvec;
template <> struct fcvec<4> { typedef float4v type; };
template <> struct fcvec<2> { typedef float2v type; };

// CQ consecutive floats at p[0..CQ-1], of which the first `valid` exist.  No lane ever branches around a
// load (a predicated load costs a branch each and serialises the batch): an address that does not
// exist is replaced by `safe`, a location that does, and whoever consumes the value ignores or
// zeroes it.  VEC: aligned groups that exist whole or not at all.
template <int CQ, bool VEC>
__device__ __forceinline__ typename fcvec<CQ>::type fc_loadq(const float *__restrict__ p, int valid,
                                                            const float *__restrict__ safe)
{
    typedef typename fcvec<CQ>::type vq;
    vq v;
    if (VEC) {
        v = *reinterpret_cast<const vq *>(valid > 0 ? p : safe);
    } else {
#pragma unroll
        for (int j = 0; j < CQ; ++j)
            v[j] = *(valid > j ? p + j : safe);
#pragma unroll
        for (int j = 0; j < CQ; ++j)
            v[j] = valid > j ? v[j] : 0.0f;
    }
    return v;
}

template <bool VEC>
__device__ __forceinline__ float4v fc_load4(const float *__restrict__ p, int valid, const float *__restrict__ safe)
{
    return fc_loadq<4, VEC>(p, valid, safe);
}

__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ---- forward ------------------------------------------------------------------------------------

constexpr int FC_MAX_GROUP = 4;     // layers one launch can take
constexpr int FC_NW = 4;            // waves per workgroup

struct FcFwdArgs {
    int M, K, N, ldx, kslice, training, relu;
    int block0, tiles, splits, rts, units, cq, vec;     // this layer's workgroups start at block0 (a multiple of 8)
    int combine;        // partial tiles meet in the scratch: K is cut, or batch norm spans several row tiles
    const float *x, *w, *bias;
    const float *gamma, *beta, *decay;      // gamma == nullptr: no batch norm
    float *ema_mean, *ema_var, *save_mean, *save_var;
    float *y, *out;
    int *tickets;       // arrival counters: one per column tile (batch norm) or per (column tile, row tile)
    float *partials;    // [tiles][rts][splits][32][32 CQ]: partial tiles, summed in a fixed order by the last to arrive
    const float *rowvec;    // no batch norm: y[r][c] += rowvec[r * rowvec_d + c % rowvec_d] (the "+ element_mean" of
    int rowvec_d;           // train_cloudAAE_ycbv.py:232-233 folded into the output layers); NULL: nothing
};

struct FcFwdGroup {
    int count;
    FcFwdArgs p[FC_MAX_GROUP];
};

// floats of LDS one workgroup of the forward kernel needs with CQ columns per lane: the four waves' partial tiles
// (later: the row tiles of the finished column tile), the fp64 column sums of the batch norm, the "I am last" word
template <int CQ> constexpr int fc_fwd_lds_floats() { return FC_NW * FC_ROWS * (32 * CQ + 4) + 2 * 2 * FC_NW * 64 + 4; }

// One workgroup = (column tile of 32 CQ columns, row tile of 32 rows, K slice) of one layer; the four waves take
// contiguous runs of the slice's k, their partial 32 x 32 CQ tiles meet in LDS, and the threads then finish
// CQ / 8 .. columns each.  CQ = 4: lanes read W as dwordx4, 128 columns per workgroup (fewest instructions per byte:
// the wide output layer); CQ = 2: dwordx2, 64 columns (half the partial-tile bytes per column tile: the last arrival
// of a layer with batch norm sums them in one round of loads).
template <int CQ, bool VEC>
__device__ __forceinline__ void fc_fwd_body(const FcFwdArgs &a, int tile_x, int rt, int slice, float *smem)
{
    constexpr int NW = FC_NW, TN = 32 * CQ, LD = TN + 4;
    constexpr int RG = NW * 64 / TN, RP = FC_ROWS / RG;      // row groups of the column threads, rows per thread and row tile
    typedef typename fcvec<CQ>::type vq;
    float *tile = smem;
    double (*red)[RG][TN] = reinterpret_cast<double (*)[RG][TN]>(smem + NW * FC_ROWS * LD);
    int *last_flag = reinterpret_cast<int *>(smem + NW * FC_ROWS * LD + 2 * 2 * NW * 64);

    FC_STAMP(0);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r32 = lane & 31, half = lane >> 5;
    const int n0 = tile_x * TN, row0 = rt * FC_ROWS;
    const int kb0 = slice * a.kslice, kb1 = min(a.K, kb0 + a.kslice);
    const int per = ((kb1 - kb0 + NW - 1) / NW + 7) & ~7;      // k per wave, whole groups of eight
    const int kw0 = kb0 + wv * per, kw1 = min(kb1, kw0 + per);
    const int colq = n0 + CQ * r32;
    // rows >= M and columns >= N are computed from existing data and never written
    const float *xrow = a.x + (size_t)min(row0 + r32, a.M - 1) * a.ldx;
    const float *wcol = a.w + (VEC ? min(colq, a.N - CQ) : colq);
    const int wvalid = VEC ? CQ : a.N - colq;

    struct Grp {            // operands of eight k: lane half h holds k + 4h .. k + 4h + 3
        float4v a;          // X[row][k4 .. k4+3]
        vq b[4];            // W[k4 + j][CQ columns]
    };
    struct Set {            // sixteen k: what a wave keeps in flight behind its MFMAs
        Grp g[2];
        bool in[2];         // group lies inside the wave's run of k (else its X operand counts as zero)
    };

    f32x16 acc[CQ];
#pragma unroll
    for (int c = 0; c < CQ; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[c][r] = 0.0f;

    auto load = [&](Set &st, int k) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ka = k + 8 * u + 4 * half;
            // k beyond the run: read the run's first rows again and zero the X operand
            const bool in = k + 8 * u < kw1;
            const int kc = in ? ka : kb0;
            st.g[u].a = fc_loadq<4, VEC>(xrow + kc, VEC ? 4 : kw1 - ka, a.x);
            st.in[u] = in;      // applied where the operand is consumed, not here: a select on the loaded
                                // value would make the wave wait for the load inside the batch
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kr = VEC ? kc + j : min(kc + j, a.K - 1);     // (a zeroed X column pairs with it)
                st.g[u].b[j] = fc_loadq<CQ, VEC>(wcol + (size_t)kr * a.N, wvalid, a.w);
            }
        }
    };
    auto mma = [&](const Set &st) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float aj = st.in[u] ? st.g[u].a[j] : 0.0f;
#pragma unroll
                for (int c = 0; c < CQ; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(aj, st.g[u].b[j][c], acc[c], 0, 0, 0);
            }
    };

    // two operand sets: the loads of one are in flight behind the MFMAs of the other
    if (VEC && kw0 < kw1) {
        // The X operand in WHOLE LINES.  The fragment-shaped load above (a lane: 16 bytes of ITS row) touches 64 lines per
        // instruction for 1 KB of operand -- eight times the bytes cross the L2 -> L1 path, and at 128 rows that is more
        // traffic than W itself.  Here a wave brings 32 k of its 32 rows as four loads of 8 rows x 128 bytes, parks them in
        // its own LDS patch (its slot of the partial-tile area, which is only written after the loop: wave-private, program
        // order and the LDS counter are all the synchronisation needed) and takes the MFMA operand from there.
        constexpr int XLD = 32 + 4;                  // floats per patch row
        float *patch = tile + (size_t)wv * FC_ROWS * LD;
        const int xr = lane >> 3, xc = 4 * (lane & 7);
        const float *xline = a.x + (size_t)min(row0 + xr, a.M - 1) * a.ldx;
        const size_t xstep = (size_t)8 * a.ldx;      // (rows past M: the last row again, never written)
        const int rows_left = a.M - 1 - min(row0 + xr, a.M - 1);
        auto xload = [&](float4v (&r)[4], int k) {
            const int kq = min(k + xc, kw1 - 4);     // (a quad past the run: the run's last one, zeroed where it is consumed)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                r[i] = *reinterpret_cast<const float4v *>(xline + (8 * i <= rows_left ? i * xstep : (size_t)rows_left * a.ldx) + kq);
        };
        auto xpark = [&](const float4v (&r)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4v *>(patch + (xr + 8 * i) * XLD + xc) = r[i];
        };
        auto loadw = [&](Set &st, int k) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ka = k + 8 * u + 4 * half;
                const bool in = k + 8 * u < kw1;
                const int kc = in ? ka : kb0;
                st.in[u] = in;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    st.g[u].b[j] = fc_loadq<CQ, VEC>(wcol + (size_t)(kc + j) * a.N, wvalid, a.w);
            }
        };
        auto frag = [&](Set &st, int klocal) {       // klocal: 0 or 16 inside the parked slab
#pragma unroll
            for (int u = 0; u < 2; ++u)
                st.g[u].a = *reinterpret_cast<const float4v *>(patch + r32 * XLD + klocal + 8 * u + 4 * half);
        };
        Set s0, s1;
        float4v x0[4], x1[4];
        xload(x0, kw0);
        loadw(s0, kw0);
        for (int k = kw0; k < kw1; k += 32) {
            loadw(s1, k + 16);
            xload(x1, k + 32);
            __builtin_amdgcn_sched_barrier(0);
            xpark(x0);
            frag(s0, 0);
            frag(s1, 16);
            __builtin_amdgcn_sched_barrier(0);
            mma(s0);
            __builtin_amdgcn_sched_barrier(0);
            loadw(s0, k + 32);
            __builtin_amdgcn_sched_barrier(0);
            mma(s1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                x0[i] = x1[i];
        }
    } else if (kw0 < kw1) {
        Set s0, s1;
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

    FC_STAMP(1);
    float *mine = tile + (size_t)wv * FC_ROWS * LD;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        vq v;
#pragma unroll
        for (int c = 0; c < CQ; ++c)
            v[c] = acc[c][r];
        *reinterpret_cast<vq *>(mine + mfma_row(r, half) * LD + CQ * r32) = v;
    }
    __syncthreads();

    const int col = threadIdx.x & (TN - 1), rg = (int)threadIdx.x / TN;
    const int c = n0 + col;
    const bool ok = c < a.N;
    const bool bn = a.gamma != nullptr;
    const float bias = (a.bias != nullptr && ok) ? a.bias[c] : 0.0f;
    // (what the batch norm at the end reads per column is requested now: three dependent trips to memory less for the
    // workgroup that finishes the column tile)
    float p_gamma = 1.0f, p_beta = 0.0f, p_em = 0.0f, p_ev = 1.0f, p_decay = 0.0f;
    if (bn && ok) {
        p_gamma = a.gamma[c];
        p_beta = a.beta[c];
        if (a.ema_mean != nullptr) {
            p_em = a.ema_mean[c];
            p_ev = a.ema_var[c];
        }
        if (a.training && a.ema_mean != nullptr)
            p_decay = a.decay[0];
    }
    // vals[r][i]: row (r_lo + r) * 32 + rg + RG i of column c
    float vals[FC_MAX_RT][RP];
    int r_lo = rt, r_n = 1;
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int row = rg + RG * i;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w)
            s += tile[((size_t)w * FC_ROWS + row) * LD + col];
        vals[0][i] = s;
    }
    if (a.combine) {
        // Partial tiles combined in a FIXED order (bit-reproducible from run to run, whatever order the workgroups
        // finish in): every workgroup publishes its partial tile with agent-scope stores (write-through to
        // where all XCDs meet), every WAVE waits until its stores are acknowledged (s_waitcnt vmcnt(0); s_barrier
        // alone does not drain the counter), then a ticket is taken; the workgroup that draws the last one reads
        // the partial tiles back with agent-scope loads and sums them in slice order 0, 1, 2, ...  No cache
        // write-back or invalidate is involved (a __threadfence() here costs more than the whole product)
        // because no ordinary store takes part.  The counter returns to zero for the next launch.
        // (tests/test_capi_symbols.py checks the emitted ISA for the wait in front of the barrier.)
        // Without batch norm a row tile is finished by the last of ITS slices; with batch norm the column tile
        // -- all row tiles, the whole batch -- by the last of all of them.
        float *slot = a.partials + ((size_t)(tile_x * a.rts + rt) * a.splits + slice) * (FC_ROWS * TN);
#pragma unroll
        for (int i = 0; i < RP; ++i)
            if (row0 + rg + RG * i < a.M)
                __hip_atomic_store(&slot[(rg + RG * i) * TN + col], vals[0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        FC_STAMP(2);
        int *ctr = a.tickets + (bn ? tile_x : tile_x * a.rts + rt);
        const int expect = bn ? a.splits * a.rts : a.splits;
        if (threadIdx.x == 0) {
            const int t = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *last_flag = t == expect - 1;
            if (t == expect - 1)
                __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        FC_STAMP(3);
        if (!*last_flag)
            return;
        if (bn) {
            r_lo = 0;
            r_n = a.rts;
        }
        // thread t owns element quads t, t + 256, ... of every 32 x TN tile (16-byte agent-scope loads through a buffer
        // descriptor, aux = sc1: half a row per wave and instruction); INFL partial tiles' loads are in flight
        // together; the running sums live in LDS (each thread adds to its own words, in list order: row tile by row
        // tile, slice 0, 1, 2, ...)
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        constexpr int QUADS = FC_ROWS * TN / 4 / (NW * 64), INFL = 16 / QUADS;
        const int total = r_n * a.splits;
        const unsigned slab = FC_ROWS * TN * sizeof(float);
        // (every slab of this column tile: descriptor inputs are workgroup-uniform)
        const float *tile_base = a.partials + (size_t)tile_x * a.rts * a.splits * (FC_ROWS * TN);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(tile_base), 0, (int)(slab * (unsigned)(a.rts * a.splits)), 0x00020000);
        for (int l0 = 0; l0 < total; l0 += INFL) {
            u32x4 raw[INFL][QUADS];
#pragma unroll
            for (int u = 0; u < INFL; ++u) {
                const int l = min(l0 + u, total - 1);       // (past the end: the last one again, not added)
                const int r = l / a.splits, s = l - r * a.splits;
                const int rows_r = min(FC_ROWS, a.M - (r_lo + r) * FC_ROWS);       // rows >= M were never published
                const unsigned soff = (unsigned)((r_lo + r) * a.splits + s) * slab;
#pragma unroll
                for (int q = 0; q < QUADS; ++q) {
                    const int e = (int)threadIdx.x + NW * 64 * q;
                    const int row = min(e / (TN / 4), rows_r - 1);
                    raw[u][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)(row * TN + 4 * (e % (TN / 4))) * 4u, soff, 16);
                }
            }
#pragma unroll
            for (int u = 0; u < INFL; ++u)
                if (l0 + u < total) {
                    const int r = (l0 + u) / a.splits, s = (l0 + u) - r * a.splits;
#pragma unroll
                    for (int q = 0; q < QUADS; ++q) {
                        const int e = (int)threadIdx.x + NW * 64 * q;
                        float4v *p = reinterpret_cast<float4v *>(tile + (size_t)(r * FC_ROWS + e / (TN / 4)) * LD + 4 * (e % (TN / 4)));
                        float4v cur = s == 0 ? float4v{0.0f, 0.0f, 0.0f, 0.0f} : *p;
                        cur.x += __uint_as_float(raw[u][q].x);
                        cur.y += __uint_as_float(raw[u][q].y);
                        cur.z += __uint_as_float(raw[u][q].z);
                        cur.w += __uint_as_float(raw[u][q].w);
                        *p = cur;
                    }
                }
        }
        __syncthreads();
        FC_STAMP(4);
#pragma unroll
        for (int r = 0; r < FC_MAX_RT; ++r)
            if (r < r_n)
#pragma unroll
                for (int i = 0; i < RP; ++i)
                    vals[r][i] = tile[(size_t)(r * FC_ROWS + rg + RG * i) * LD + col];
    }
#pragma unroll
    for (int r = 0; r < FC_MAX_RT; ++r)
        if (r < r_n)
#pragma unroll
            for (int i = 0; i < RP; ++i)
                vals[r][i] += bias;

    if (!bn) {
        if (ok) {
            const int cd = a.rowvec != nullptr ? c % a.rowvec_d : 0;
#pragma unroll
            for (int r = 0; r < FC_MAX_RT; ++r)
                if (r < r_n)
#pragma unroll
                    for (int i = 0; i < RP; ++i) {
                        const int row = (r_lo + r) * FC_ROWS + rg + RG * i;
                        if (row < a.M) {
                            float out = vals[r][i];
                            if (a.rowvec != nullptr)
                                out = out + a.rowvec[(size_t)row * a.rowvec_d + cd];
                            a.y[(size_t)row * a.N + c] = out;
                        }
                    }
        }
        return;
    }

    // batch norm of the column (same arithmetic as bn_small_fwd_kernel: fp64 sums, fp32 formulas); r_lo = 0 and
    // the r_n row tiles are the whole batch here
    float mean, var;
    if (a.training) {
        double s = 0.0, s2 = 0.0;
#pragma unroll
        for (int r = 0; r < FC_MAX_RT; ++r)
            if (r < r_n)
#pragma unroll
                for (int i = 0; i < RP; ++i)
                    if (r * FC_ROWS + rg + RG * i < a.M) {
                        s += (double)vals[r][i];
                        s2 += (double)vals[r][i] * (double)vals[r][i];
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
            const float om = 1.0f - p_decay;
            a.ema_mean[c] = p_em - (p_em - mean) * om;
            a.ema_var[c] = p_ev - (p_ev - var) * om;
        }
    } else {
        mean = ok ? p_em : 0.0f;
        var = ok ? p_ev : 1.0f;
    }
    if (!ok)
        return;
    if (rg == 0) {
        a.save_mean[c] = mean;
        a.save_var[c] = var;
    }
    const float inv = p_gamma * bn_rsqrt(var + BN_EPS);
    const float sh = p_beta - mean * inv;
#pragma unroll
    for (int r = 0; r < FC_MAX_RT; ++r)
        if (r < r_n)
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                const int row = r * FC_ROWS + rg + RG * i;
                if (row < a.M) {
                    a.y[(size_t)row * a.N + c] = vals[r][i];
                    float z = vals[r][i] * inv + sh;
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

// CQMAX: the widest column tile among the launch's layers (LDS is sized for it).  Workgroup b runs on XCD b % 8
// (MI355X_MICROARCH.md) and every layer's range starts at a multiple of 8: the row tiles of one (column tile, slice)
// are consecutive workgroups of ONE XCD -- they stream the same slab of W at the same time, once from HBM and the rest
// from that XCD's L2.  (A different placement would change speed, never results.)
template <int CQMAX>
__global__ __launch_bounds__(FC_NW * 64) void fc_fwd_kernel(FcFwdGroup g)
{
    __shared__ float4v smem4[(fc_fwd_lds_floats<CQMAX>() + 3) / 4];
    float *smem = reinterpret_cast<float *>(smem4);
    const FcFwdArgs a = g.p[fc_group_member(g)];
    const int local = (int)blockIdx.x - a.block0;
    const int xcd = local & 7, j = local >> 3;
    const int unit = (j / a.rts) * 8 + xcd, rt = j % a.rts;
    if (unit >= a.units)
        return;
    const int tile_x = unit % a.tiles, slice = unit / a.tiles;
    if (CQMAX == 4 && a.cq == 4) {
        if (a.vec)
            fc_fwd_body<4, true>(a, tile_x, rt, slice, smem);
        else
            fc_fwd_body<4, false>(a, tile_x, rt, slice, smem);
    } else {
        if (a.vec)
            fc_fwd_body<2, true>(a, tile_x, rt, slice, smem);
        else
            fc_fwd_body<2, false>(a, tile_x, rt, slice, smem);
    }
#ifdef CLOUDAAE_FC_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FC_STAMP(5);
#endif
}

// ---- backward -----------------------------------------------------------------------------------

struct FcBwdArgs {
    int M, K, N, ldx, lddo, lddx, tiles_per_block, acc_dw, acc_pg, training, relu;
    int block0, slices, parts, vec;     // this layer's workgroups: block0 .. block0 + slices * groups - 1
    const float *x, *w, *y, *gamma, *beta, *save_mean, *save_var, *dout;
    float *dx, *dw, *dgamma, *dbeta, *dbias;
};

// the X operand of dW for one 32-row tile of W: lane = row of W, 16 RT batch-row pairs (rows past M meet zero rows of dY)
template <int RT>
__device__ __forceinline__ void fc_bwd_dw_load(const FcBwdArgs &a, int kr0, int r32, int half, float (&xa)[16 * RT])
{
    const int krc = min(kr0 + r32, a.K - 1);    // rows of W past K: an existing one, result not written
    // (a running row offset the optimiser cannot hoist out of the tile loop as 16 RT separate registers)
    unsigned off = (unsigned)half * (unsigned)a.ldx;
    const unsigned step = 2u * (unsigned)a.ldx, last = (unsigned)(a.M - 1) * (unsigned)a.ldx;
    asm volatile("" : "+v"(off));
    const float *xk = a.x + krc;
#pragma unroll
    for (int s = 0; s < 16 * RT; ++s) {
        xa[s] = xk[min(off, last)];
        off += step;
        if (RT > 1 && (s & 15) == 15)
            __builtin_amdgcn_sched_barrier(0);     // (or every address of the batch is formed before the first load)
    }
}

// dW[kr0.., NC columns per lane from column c0] of one 32-row tile of W: 16 RT MFMA steps over the batch
template <bool VEC, int NC, int RT>
__device__ __forceinline__ void fc_bwd_dw(const FcBwdArgs &a, const float *dyl, int kr0, int c0, int lcol, int half,
                                          const float (&xa)[16 * RT])
{
    f32x16 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[c][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 16 * RT; ++s) {
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
        // (several row tiles: left alone the scheduler requests all 16 RT rows of dY from LDS ahead of the first
        // MFMA -- 256 registers at 128 rows -- and spills)
        if (RT > 1 && (s & 7) == 7)
            __builtin_amdgcn_sched_barrier(0);
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

// the batch rows' gradients of one 32-row tile of W leave with fp32 atomics (the slices of the output meet in dx).
// The 16 RT row addresses are formed by a running offset the optimiser cannot see through: left to itself it hoists
// all of them (64-bit pairs, and a lane mask per row) out of the tile loop -- 128 registers at four row tiles, spilled.
template <int RT>
__device__ __forceinline__ void fc_bwd_dx_add(const FcBwdArgs &a, const f32x16 (&d)[RT], int kr, int half)
{
    if (kr >= a.K)
        return;
    unsigned off = (unsigned)(4 * half) * (unsigned)a.lddx + (unsigned)kr;      // row mfma_row(0, half), column kr
    const unsigned step = (unsigned)a.lddx;
    asm volatile("" : "+v"(off));
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const bool whole = (rt + 1) * FC_ROWS <= a.M;       // (uniform: no lane mask per row)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (whole || rt * FC_ROWS + mfma_row(r, half) < a.M)
                atomicAdd(a.dx + off, d[rt][r]);
            off += (r & 3) == 3 ? 5 * step : step;          // rows (r & 3) + 8 (r >> 2) + 4 half
        }
    }
}

// dX[:, kr0..kr0+31] += dY[:, 8 q0 .. 8 (q0+NQ)) W[tile, same columns]^T: 4 NQ MFMA steps per row tile, the W operand
// (lane = row of W) shared by the row tiles
template <bool VEC, int NQ, int RT>
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
        wq[q] = fc_loadq<4, VEC>(wrow + (VEC ? min(cq, a.N - 4) : cq), VEC ? 4 : a.N - cq, a.w);
    }
    f32x16 d[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            d[rt][r] = 0.0f;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const float4v aq = *reinterpret_cast<const float4v *>(dyl + (rt * FC_ROWS + r32) * FC_LD + 8 * (q0 + q) + 4 * half);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                d[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[j], wq[q][j], d[rt], 0, 0, 0);
            if (RT > 1)
                __builtin_amdgcn_sched_barrier(0);     // (as in fc_bwd_dw)
        }
    fc_bwd_dx_add<RT>(a, d, kr, half);
}

// The same product for the wide output layer at one row tile (VEC only), W read the way it lies in memory.  Above, a
// lane follows ONE row of W, so a load instruction touches 32 rows x 32 bytes: every 128-byte line is
// requested by four instructions and crosses the L2 -> L1 path four times (2.1 TB/s on the 50 MB of the
// output layer).  Here the wave reads 64-column halves of the tile row by row (half-wave = 256
// contiguous bytes), parks them in its own LDS patch and takes the MFMA operand (lane = row) from
// there.  The loads of a tile are issued by fc_bwd_w_load -- before the workgroup derives dY, so that the first
// trip to memory of both products overlaps that phase.
constexpr int FC_WLD = 64 + 4;      // LDS row stride of the half tile (floats)

__device__ __forceinline__ void fc_bwd_w_load(const FcBwdArgs &a, int kr0, int n0, int lane, float4v (&wq)[2][8])
{
    // load h, i: rows 4 i + (lane >> 4), columns 64 h + 4 (lane & 15)
    const int lrow = lane >> 4, lcol = 4 * (lane & 15);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = min(kr0 + 4 * i + lrow, a.K - 1);       // past K: an existing row, never written
            const int col = min(n0 + 64 * h + lcol, a.N - 4);       // past N: meets zeros of dY
            wq[h][i] = *reinterpret_cast<const float4v *>(a.w + (size_t)row * a.N + col);
        }
}

__device__ __forceinline__ void fc_bwd_dx_staged(const FcBwdArgs &a, const float *dyl, float *patch, int kr0,
                                                 int r32, int half, int lane, const float4v (&wq)[2][8])
{
    const int lrow = lane >> 4, lcol = 4 * (lane & 15);
    f32x16 d[1];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        d[0][r] = 0.0f;
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
                d[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[j], bq[j], d[0], 0, 0, 0);
        }
    }
    fc_bwd_dx_add<1>(a, d, kr0 + r32, half);
}

struct FcBwdGroup {
    int count;
    FcBwdArgs p[FC_MAX_GROUP];
};

// One workgroup (four waves) = (128-column slice of the output, group of 32-row tiles of W), all RT row tiles of the batch.
// parts = 1: a wave does both products of a tile (the wide output layer);
// parts = 2: the same tiles, but waves 0, 1 write dW and waves 2, 3 gather dX: the two streams (W read + atomics, dW
//            written) run side by side on different SIMDs instead of one after the other in every wave;
// parts = 4: the four waves share a tile -- dW columns 0..63 | 64..127, dX over columns 0..63 | 64..127 --
//            for the layers whose whole backward is a few hundred tiles (one tile per workgroup, every
//            workgroup resident at once).
template <bool VEC, int RT>
__device__ __forceinline__ void fc_bwd_body(const FcBwdArgs &a, int slice_x, int group_y, float *dyl,
                                            double (*red)[2][FC_TN], float *patches)
{
    constexpr int HF = 2, RP = FC_ROWS * RT / HF;       // row groups of the first phase, rows per thread
    FC_STAMP(0);
    const int n0 = slice_x * FC_TN;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int r32 = lane & 31, half = lane >> 5;
    const int ktiles = (a.K + 31) / 32;
    const int t_end = min(ktiles, (group_y + 1) * a.tiles_per_block);
    // one row tile, a tile per wave, W in whole quads: the first tile's operands are requested NOW
    constexpr bool STAGED = VEC && RT == 1;
    const bool ahead = STAGED && a.parts <= 2;
    // parts = 2: waves 0, 1 write dW and waves 2, 3 gather dX, each pair over the workgroup's tiles alternately
    const int role = a.parts == 2 ? wv >> 1 : -1;       // -1: both products
    const int t_first = group_y * a.tiles_per_block + (a.parts == 2 ? (wv & 1) : wv);
    const int t_step = a.parts == 2 ? 2 : 4;
    const bool do_dw = a.dw != nullptr && role != 1, do_dx = a.dx != nullptr && role != 0;
    float4v wq[STAGED ? 2 : 1][8];
    float xa0[16 * RT];
    if constexpr (STAGED) {
        if (ahead && t_first < t_end) {
            if (do_dx)
                fc_bwd_w_load(a, t_first * 32, n0, lane, wq);
            if (do_dw)
                fc_bwd_dw_load<RT>(a, t_first * 32, r32, half, xa0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    {   // d(pre-BN output) of this slice (bn_small_bwd_kernel's arithmetic)
        const int col = threadIdx.x & (FC_TN - 1), hf = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 7);
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
        float dz[RP], yv[RP];
        // all loads first, from addresses that exist (no branch around any of them)
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            dz[i] = a.dout[(size_t)min(hf * RP + i, a.M - 1) * a.lddo + cc];
            if (RT > 1 && (i & 15) == 15)
                __builtin_amdgcn_sched_barrier(0);     // (or every address of the batch is formed before the first load)
        }
#pragma unroll
        for (int i = 0; i < RP; ++i)
            yv[i] = 0.0f;
        if (bn)
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                yv[i] = a.y[(size_t)min(hf * RP + i, a.M - 1) * a.N + cc];
                if (RT > 1 && (i & 15) == 15)
                    __builtin_amdgcn_sched_barrier(0);
            }
        double s = 0.0, s2 = 0.0;
        // (selects, no branches: without batch norm mean = 0, rstd = 1, inv = sh = 0 and y counts as zero)
        const bool mask = bn && a.relu;
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const int r = hf * RP + i;
            const bool in = ok && r < a.M;
            float d = in ? dz[i] : 0.0f;
            const float v = in ? yv[i] : 0.0f;
            float z = v * inv + sh;
            z = a.relu ? fmaxf(z, 0.0f) : z;
            d = (mask && !(z > 0.0f)) ? 0.0f : d;
            const float xh = (v - mean) * rstd;
            yv[i] = xh;             // (from here on the normalised value; zero without batch norm)
            dz[i] = d;
            s += (double)d;
            s2 += (double)d * (double)xh;
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
                    v = gr * ((dz[i] - m1) - yv[i] * m2);
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
    FC_STAMP(1);

    if (a.parts <= 2) {
        for (int t = t_first; t < t_end; t += t_step) {
            if constexpr (STAGED) {
                if (t != t_first) {
                    if (do_dx)
                        fc_bwd_w_load(a, t * 32, n0, lane, wq);
                    if (do_dw)
                        fc_bwd_dw_load<RT>(a, t * 32, r32, half, xa0);
                }
                if (do_dw)
                    fc_bwd_dw<VEC, 4, RT>(a, dyl, t * 32, n0 + 4 * r32, 4 * r32, half, xa0);
                if (do_dx)
                    fc_bwd_dx_staged(a, dyl, patches + wv * (FC_ROWS * FC_WLD), t * 32, r32, half, lane, wq);
            } else {
                if (do_dw) {
                    fc_bwd_dw_load<RT>(a, t * 32, r32, half, xa0);
                    fc_bwd_dw<VEC, 4, RT>(a, dyl, t * 32, n0 + 4 * r32, 4 * r32, half, xa0);
                }
                // (several row tiles: the W operand of dX is not requested behind dW's MFMAs -- 64 more live registers,
                // and the second workgroup of the CU hides the trip to memory better than they would)
                if (RT > 1)
                    __builtin_amdgcn_sched_barrier(0);
                if (do_dx)
                    fc_bwd_dx<VEC, 16, RT>(a, dyl, t * 32, n0, 0, r32, half);
                if (RT > 1)
                    __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        const int side = wv & 1;
        for (int t = group_y * a.tiles_per_block; t < t_end; ++t) {
            if (wv < 2) {
                if (a.dw != nullptr) {
                    fc_bwd_dw_load<RT>(a, t * 32, r32, half, xa0);
                    fc_bwd_dw<VEC, 2, RT>(a, dyl, t * 32, n0 + 64 * side + 2 * r32, 64 * side + 2 * r32, half, xa0);
                }
            } else if (a.dx != nullptr) {
                fc_bwd_dx<VEC, 8, RT>(a, dyl, t * 32, n0, 8 * side, r32, half);
            }
        }
    }
}

// floats of LDS of the backward kernel at RT row tiles: dY of the slice, the fp64 column sums, (one row tile) the
// waves' W patches
template <int RT> constexpr int fc_bwd_lds_floats()
{
    return FC_ROWS * RT * FC_LD + 2 * 2 * 2 * FC_TN + (RT == 1 ? 4 * FC_ROWS * FC_WLD : 0);
}

template <int RT>
__global__ __launch_bounds__(256, 2) void fc_bwd_kernel(FcBwdGroup g)
{
    __shared__ float4v smem4[fc_bwd_lds_floats<RT>() / 4];
    float *dyl = reinterpret_cast<float *>(smem4);
    double (*red)[2][FC_TN] = reinterpret_cast<double (*)[2][FC_TN]>(dyl + FC_ROWS * RT * FC_LD);
    float *patches = dyl + FC_ROWS * RT * FC_LD + 2 * 2 * 2 * FC_TN;
    const FcBwdArgs a = g.p[fc_group_member(g)];
    const int local = (int)blockIdx.x - a.block0;
    const int slice_x = local % a.slices, group_y = local / a.slices;
    if (a.vec)
        fc_bwd_body<true, RT>(a, slice_x, group_y, dyl, red, patches);
    else
        fc_bwd_body<false, RT>(a, slice_x, group_y, dyl, red, patches);
#ifdef CLOUDAAE_FC_PROFILE
    FC_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FC_STAMP(3);
#endif
}

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }


// Column tiles, row tiles and K slices of one forward layer.  Four waves per workgroup, each with at least sixteen
// k.  These products are short chains of load -> MFMA: what they need is every load of the layer in flight at once,
// i.e. many workgroups -- but every slice costs its column tile a publish / ticket / read-back episode of 5-7 us
// (profiles/r05_fc_phases_first.log) whose length grows with the slices, so K is cut sparingly: a layer with batch
// norm (whose last arrival finishes ALL rows of the column tile) takes 64-column tiles and about 64 workgroups, the
// wide output layer 128-column tiles and about 256 (measured: profiles/r05_fc_fwd_block_sweep.log).
struct FcFwdPlan {
    int cq, tiles, rts, splits, kslice, units, blocks, combine;
};
static FcFwdPlan fc_fwd_plan(int M, int K, int N, bool bn, bool no_scratch)
{
    FcFwdPlan p;
    p.rts = ceil_div(M, FC_ROWS);
    // a layer without batch norm (the 12288-column output layer): at ONE row tile 64-column tiles with K whole -- 192 workgroups,
    // no publish / ticket / read-back episode: 25.7 -> 20.7 us in the B = 32 step (round 6; 128-column tiles cut over K twice
    // were 208 workgroups and an episode of 5-7 us) --, at several row tiles 128-column tiles as before
    p.cq = bn ? 2 : (p.rts == 1 ? 2 : 4);
    const int want = bn ? 64 : (p.rts == 1 ? 192 : 256);
    p.tiles = ceil_div(N, 32 * p.cq);
    int splits = want / (p.tiles * p.rts);
    const int most = K / (16 * FC_NW);
    splits = splits > most ? most : splits;
    splits = splits < 1 ? 1 : splits;
    if (no_scratch)
        splits = 1;
    p.kslice = ceil_div(ceil_div(K, splits), 8) * 8;
    p.splits = ceil_div(K, p.kslice);
    p.units = p.tiles * p.splits;
    p.blocks = ceil_div(p.units, 8) * 8 * p.rts;
    p.combine = p.splits > 1 || (bn && p.rts > 1);
    return p;
}

} // namespace cloudaae

using namespace cloudaae;

#ifdef CLOUDAAE_FC_PROFILE
CLOUDAAE_API int cloudaae_fc_profile_read(unsigned long long *host, int clear)
{
    if (host && hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fc_prof), sizeof(unsigned long long) * 8192 * 8) != hipSuccess)
        return 1;
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fc_prof)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned long long) * 8192 * 8) != hipSuccess)
            return 1;
    }
    return 0;
}
#endif

CLOUDAAE_API int cloudaae_fc_max_rows(void) { return FC_ROWS * FC_MAX_RT; }
CLOUDAAE_API int cloudaae_fc_max_group(void) { return FC_MAX_GROUP; }
CLOUDAAE_API int cloudaae_fc_forward_tickets(int M, int N)
{
    return (M > 0 && N > 0) ? ceil_div(N, 64) * ceil_div(M, FC_ROWS) : 0;      // (the narrowest column tile)
}
CLOUDAAE_API long long cloudaae_fc_forward_partials(int M, int K, int N, int batch_norm)
{
    if (M <= 0 || M > FC_ROWS * FC_MAX_RT || K <= 0 || N <= 0)
        return 0;
    const FcFwdPlan p = fc_fwd_plan(M, K, N, batch_norm != 0, false);
    return p.combine ? (long long)p.tiles * p.rts * p.splits * FC_ROWS * (32 * p.cq) : 0;
}

CLOUDAAE_API int cloudaae_fc_forward_group(int M, int count, const cloudaae_fc_layer *layers, int training,
                                           const float *decay, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_fc_forward_group";
    CLOUDAAE_REQUIRE(M > 0 && M <= FC_ROWS * FC_MAX_RT, name, "bad size (rows must be <= 128)");
    CLOUDAAE_REQUIRE(count > 0 && count <= FC_MAX_GROUP && layers, name, "1 to 4 layers per call");
    hipStream_t s = (hipStream_t)stream;
    FcFwdGroup g;
    g.count = count;
    int blocks = 0, cqmax = 2;
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
        // without the arrival counters and the partial-tile scratch a layer keeps K whole in one workgroup per
        // column tile (slower)
        const bool no_scratch = l.tickets == nullptr || l.partials == nullptr;
        const FcFwdPlan p = fc_fwd_plan(M, l.K, l.N, bn, no_scratch);
        CLOUDAAE_REQUIRE(!p.combine || !no_scratch, name,
                         "batch norm over more than 32 rows needs the tickets and the partial-tile scratch");
        // (the cut is derived again at every launch, also from development knobs: a buffer sized by an earlier query
        //  must still cover it)
        CLOUDAAE_REQUIRE(!p.combine || l.partials_floats >= (long long)p.tiles * p.rts * p.splits * FC_ROWS * (32 * p.cq),
                         name, "partials_floats is smaller than this launch's partial tiles (cloudaae_fc_forward_partials; "
                         "did a split knob change since the query?)");
        FcFwdArgs &a = g.p[i];
        a.M = M; a.K = l.K; a.N = l.N; a.ldx = l.ldx; a.kslice = p.kslice; a.combine = p.combine;
        a.training = training; a.relu = l.relu;
        a.block0 = blocks; a.tiles = p.tiles; a.splits = p.splits; a.rts = p.rts; a.units = p.units; a.cq = p.cq;
        a.vec = l.K % 8 == 0 && l.ldx % 4 == 0 && l.N % 4 == 0 && aligned16(l.x) && aligned16(l.w);
        a.x = l.x; a.w = l.w; a.bias = l.bias; a.gamma = l.gamma; a.beta = l.beta; a.decay = decay;
        a.ema_mean = l.ema_mean; a.ema_var = l.ema_var; a.save_mean = l.save_mean; a.save_var = l.save_var;
        a.y = l.y; a.out = l.out;
        CLOUDAAE_REQUIRE(l.out_rowvec == nullptr || (!bn && l.out_rowvec_d > 0), name,
                         "a row vector can only be added to the output of a layer without batch norm");
        a.rowvec = l.out_rowvec; a.rowvec_d = l.out_rowvec_d;
        a.partials = p.combine ? l.partials : nullptr;
        a.tickets = p.combine ? l.tickets : nullptr;
        cqmax = p.cq > cqmax ? p.cq : cqmax;
        blocks += p.blocks;
    }
    if (cqmax == 4)
        hipLaunchKernelGGL(fc_fwd_kernel<4>, dim3(blocks), dim3(FC_NW * 64), 0, s, g);
    else
        hipLaunchKernelGGL(fc_fwd_kernel<2>, dim3(blocks), dim3(FC_NW * 64), 0, s, g);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_fc_backward_group(int M, int count, const cloudaae_fc_layer *layers, int training,
                                            cloudaae_stream_t stream)
{
    const char *name = "cloudaae_fc_backward_group";
    CLOUDAAE_REQUIRE(M > 0 && M <= FC_ROWS * FC_MAX_RT, name, "bad size (rows must be <= 128)");
    CLOUDAAE_REQUIRE(count > 0 && count <= FC_MAX_GROUP && layers, name, "1 to 4 layers per call");
    hipStream_t s = (hipStream_t)stream;
    const int rts = ceil_div(M, FC_ROWS);
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
        const bool fine = (long long)slices * ktiles <= 1024;
        int by;
        if (fine) {
            // (several row tiles: every workgroup of a slice derives dY for all rows of the slice again -- 128 KB of
            // L2 reads at 128 rows -- so a workgroup keeps two tiles)
            by = ceil_div(ktiles, rts > 1 ? 2 : 1);
        } else {
            const int want = 384;
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
        a.block0 = blocks; a.slices = slices; a.parts = fine ? 4 : 2;
        a.vec = l.N % 4 == 0 && aligned16(l.w) && (l.dw == nullptr || aligned16(l.dw));
        a.x = l.x; a.w = l.w; a.y = l.y; a.gamma = l.gamma; a.beta = l.beta; a.save_mean = l.save_mean;
        a.save_var = l.save_var; a.dout = l.dout; a.dx = l.dx; a.dw = l.dw; a.dgamma = l.dgamma;
        a.dbeta = l.dbeta; a.dbias = l.dbias;
        blocks += slices * by;
    }
    if (rts == 1)
        hipLaunchKernelGGL(fc_bwd_kernel<1>, dim3(blocks), dim3(256), 0, s, g);
    else if (rts == 2)
        hipLaunchKernelGGL(fc_bwd_kernel<2>, dim3(blocks), dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL(fc_bwd_kernel<4>, dim3(blocks), dim3(256), 0, s, g);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_fc_forward(int M, int K, int N, const float *x, int ldx, const float *w,
                                     const float *bias, const float *gamma, const float *beta, int training,
                                     const float *decay, float *ema_mean, float *ema_var, float *save_mean,
                                     float *save_var, int relu, float *y, float *out, int *tickets,
                                     float *partials, long long partials_floats, cloudaae_stream_t stream)
{
    cloudaae_fc_layer l = {};
    l.K = K; l.N = N; l.x = x; l.ldx = ldx; l.w = w; l.bias = bias; l.gamma = gamma; l.beta = beta;
    l.ema_mean = ema_mean; l.ema_var = ema_var; l.save_mean = save_mean; l.save_var = save_var; l.relu = relu;
    l.y = y; l.out = out; l.tickets = tickets; l.partials = partials; l.partials_floats = partials_floats;
    return cloudaae_fc_forward_group(M, 1, &l, training, decay, stream);
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
