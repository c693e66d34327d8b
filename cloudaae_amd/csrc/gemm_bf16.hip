// gemm_bf16.hip -- the dense layers with bf16 operands on the gfx950 matrix cores
// (v_mfma_f32_32x32x16_bf16), fp32 accumulate, fp32 in and out.
//
// BASELINE configs[2] ("bf16 MLPs + fp32 Chamfer"): every conv1x1 / fully connected product of the
// CloudAAE path (reference utils/tf_util.py:161-166, :349-352) and both of its gradient products
//     y  = x W + b,   dx = dy W^T,   dW = x^T dy
// with the operands rounded to bfloat16 (round to nearest even, v_cvt_pk_bf16_f32) as they are
// staged into LDS; tensors stay fp32 in HBM, so nothing else in the step changes.  The bf16 MFMA
// runs at 16x the fp32 MFMA rate, which moves these products from the matrix pipe to HBM: the
// dgcnn_agg forward product writes 134 MB of fp32 output.
//
// Same decomposition as gemm.hip: a workgroup of 4 waves owns a BM x BN tile as 32x32 accumulator
// tiles, K walked in slabs of 32 (two MFMA k-steps), next slab prefetched through registers, K
// split across grid.z with fp32 atomics when the output has too few tiles.  One MFMA consumes 8
// bf16 per operand per lane (row/column = lane & 31, k-block = lane >> 5), so both operands are
// staged k-contiguous, rows padded to 40 bf16 (80 B: conflict-free ds_read_b128); an operand whose
// memory layout is [k][outer] is transposed on the way in by packing two k-rows per 32-bit store.
#include <cstdlib>
#include "common.h"
#include "gemm.h"
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int GB_BK = 32;            // k per slab
constexpr int GB_LDK = 40;           // bf16 per staged row (32 + 8 pad)
constexpr int GB_THREADS = 256;

enum { GB_STORE = 0, GB_ACCUM = 1, GB_ATOMIC = 2 };

// folded row-major matrix (see gemm.hip): logical (r, c) -> row (c >> shift) * rows + r, column c & mask
struct FoldB {
    int shift, rows;
};
__device__ __forceinline__ size_t foldb_off(int r, int c, int ld, FoldB f)
{
    if (f.shift < 0)
        return (size_t)r * ld + c;
    return (size_t)((c >> f.shift) * f.rows + r) * ld + (c & ((1 << f.shift) - 1));
}

// two floats -> two bfloat16 (round to nearest even) in one 32-bit word, lo first
__device__ __forceinline__ unsigned pack2(float lo, float hi)
{
    const bf16x2 p = {(__bf16)lo, (__bf16)hi};
    unsigned w;
    __builtin_memcpy(&w, &p, 4);
    asm volatile("" : "+v"(w));      // keep the pair packed (the compiler otherwise selects halves and re-packs)
    return w;
}

// One operand slab: ROWS outer indices x 32 k.  KC: memory is [outer][k]; else [k][outer].
template <int ROWS, bool KC>
struct SlabB {
    // KC : float4 = 4 consecutive k of one outer row      -> one 64-bit LDS store
    // !KC: two float4 = 4 consecutive outer at k, k + 1   -> four 32-bit LDS stores
    static constexpr int ITEMS = KC ? ROWS * (GB_BK / 4) : (GB_BK / 2) * (ROWS / 4);
    static constexpr int PER = (ITEMS + GB_THREADS - 1) / GB_THREADS;
    float4v r0[PER], r1[KC ? 1 : PER];

    // !KC items = 16 k-pairs x ROWS/4 groups of four outer indices.  A 32-bit LDS store is served per half wave over
    // 32 banks, and the rows of one item are 4 * 20 dwords apart (a multiple of 16 banks): with the lanes of a half wave
    // on 32 consecutive groups of ONE k-pair (the obvious, fully coalesced order) every store hit 2 banks, a 16-way
    // conflict that cost more LDS cycles than the slab's MFMAs (the W operand of every forward product, both operands
    // of every weight-gradient product).  Instead a half wave covers 4 k-pairs x 8 groups (still whole 128-byte
    // lines of 8 consecutive groups per k row), and lane by lane the four rows are written in a rotated order
    // ((oq >> 1) & 3): bank = 16 (oq & 1) + 20 ((i + rot) & 3) + kp mod 32 takes all 32 values.
    static_assert(KC || ROWS % 32 == 0, "groups of 8 x 4 outer indices");
    static __device__ __forceinline__ int item_kp(int it) { return ((it >> 5) & 3) * 4 + (it & 3); }
    static __device__ __forceinline__ int item_oq(int it) { return (it >> 7) * 8 + ((it >> 2) & 7); }

    __device__ __forceinline__ float4v fetch4(const float *__restrict__ src, int avail, bool vec) const
    {
        float4v v = {0.f, 0.f, 0.f, 0.f};
        if (avail >= 4 && vec) {
            v = *reinterpret_cast<const float4v *>(src);
        } else {
            if (avail > 0) v.x = src[0];
            if (avail > 1) v.y = src[1];
            if (avail > 2) v.z = src[2];
            if (avail > 3) v.w = src[3];
        }
        return v;
    }

    __device__ __forceinline__ void load(const float *__restrict__ P, int ld, int outer0, int nouter, int k0,
                                         int kend, bool vec, FoldB fold)
    {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = u * GB_THREADS + (int)threadIdx.x;
            float4v a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            if (ITEMS % GB_THREADS == 0 || it < ITEMS) {
                if (KC) {
                    const int o = it / (GB_BK / 4), kq = it % (GB_BK / 4);
                    const int go = outer0 + o, gk = k0 + 4 * kq;
                    if (go < nouter && gk < kend)
                        a = fetch4(P + foldb_off(go, gk, ld, fold), kend - gk, vec);
                } else {
                    const int kp = item_kp(it), oq = item_oq(it);
                    const int gk = k0 + 2 * kp, go = outer0 + 4 * oq;
                    if (go < nouter) {
                        if (gk < kend)
                            a = fetch4(P + foldb_off(gk, go, ld, fold), nouter - go, vec);
                        if (gk + 1 < kend)
                            b = fetch4(P + foldb_off(gk + 1, go, ld, fold), nouter - go, vec);
                    }
                }
            }
            r0[u] = a;
            if (!KC)
                r1[u] = b;
        }
    }

    // FAST path (whole tiles, whole slabs, 16-byte aligned rows: decided at launch): the slab is addressed
    // as ONE wave-uniform pointer that advances per slab plus per-thread byte offsets computed once -- a load is a
    // single global_load_dwordx4 with no bounds checks.  (The generic path spends ~400 scalar and vector instructions
    // per slab on predicates and 64-bit addresses next to 8 MFMAs: with bf16 operands the matrix pipe is 16 x faster
    // than with fp32 ones, and those instructions were what the dgcnn_agg products waited on.)
    unsigned boff[PER];
    // fold (the B operand only; see FoldB): a k-contiguous operand folds over k and its slabs lie inside one fold block
    // (checked at launch), so the fold only moves the slab origin (fast_origin); a [k][outer] operand folds over the outer
    // index, a per-thread constant -- its offsets carry the tile origin outer0 and the slab origin is row k alone.
    __device__ __forceinline__ void fast_init(int ld, FoldB fold, int outer0)
    {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = u * GB_THREADS + (int)threadIdx.x;
            if (KC)
                boff[u] = 4u * (unsigned)((it / (GB_BK / 4)) * ld + 4 * (it % (GB_BK / 4)));
            else
                boff[u] = 4u * (unsigned)foldb_off(2 * item_kp(it), outer0 + 4 * item_oq(it), ld, fold);
        }
    }
    static __device__ __forceinline__ const float *fast_origin(const float *__restrict__ P, int ld, FoldB fold, int outer0,
                                                               int k)
    {
        return KC ? P + foldb_off(outer0, k, ld, fold) : P + (size_t)k * ld;
    }
    __device__ __forceinline__ void fast_load(const float *__restrict__ P0, int ld)
    {
        const char *base = reinterpret_cast<const char *>(P0);
        const char *base1 = reinterpret_cast<const char *>(P0 + ld);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = u * GB_THREADS + (int)threadIdx.x;
            if (ITEMS % GB_THREADS == 0 || it < ITEMS) {
                r0[u] = *reinterpret_cast<const float4v *>(base + boff[u]);
                if (!KC)
                    r1[u] = *reinterpret_cast<const float4v *>(base1 + boff[u]);
            }
        }
    }

    __device__ __forceinline__ void stage(__bf16 *__restrict__ lds) const
    {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = u * GB_THREADS + (int)threadIdx.x;
            if (ITEMS % GB_THREADS == 0 || it < ITEMS) {
                if (KC) {
                    const int o = it / (GB_BK / 4), kq = it % (GB_BK / 4);
                    bf16x4 p = {(__bf16)r0[u].x, (__bf16)r0[u].y, (__bf16)r0[u].z, (__bf16)r0[u].w};
                    *reinterpret_cast<bf16x4 *>(lds + o * GB_LDK + 4 * kq) = p;
                } else {
                    const int kp = item_kp(it), oq = item_oq(it);
                    // (packed first, rotated as 32-bit words: eight selects per item)
                    const unsigned p0 = pack2(r0[u].x, r1[u].x), p1 = pack2(r0[u].y, r1[u].y),
                                   p2 = pack2(r0[u].z, r1[u].z), p3 = pack2(r0[u].w, r1[u].w);
                    // the four rows 4 oq .. 4 oq + 3 are written in the order (i + rot) & 3: see item_kp
                    const int rot = (oq >> 1) & 3;
                    const unsigned t0 = (rot & 1) ? p1 : p0, t1 = (rot & 1) ? p2 : p1, t2 = (rot & 1) ? p3 : p2,
                                   t3 = (rot & 1) ? p0 : p3;
                    const unsigned q0 = (rot & 2) ? t2 : t0, q1 = (rot & 2) ? t3 : t1, q2 = (rot & 2) ? t0 : t2,
                                   q3 = (rot & 2) ? t1 : t3;
                    __bf16 *dst = lds + (4 * oq) * GB_LDK + 2 * kp;
                    *reinterpret_cast<unsigned *>(dst + ((0 + rot) & 3) * GB_LDK) = q0;
                    *reinterpret_cast<unsigned *>(dst + ((1 + rot) & 3) * GB_LDK) = q1;
                    *reinterpret_cast<unsigned *>(dst + ((2 + rot) & 3) * GB_LDK) = q2;
                    *reinterpret_cast<unsigned *>(dst + ((3 + rot) & 3) * GB_LDK) = q3;
                }
            }
        }
    }
};

// C[M,N] (+)= bf16(op(A))[M,K] * bf16(op(B))[K,N] (+ bias[N]), fp32 accumulate
template <int BM, int BN, int WM, int WN, bool TA, bool TB, bool FAST>
__global__ __launch_bounds__(GB_THREADS) void gemm_bf16_kernel(int M, int N, int K, const float *__restrict__ A,
                                                               int lda, const float *__restrict__ B, int ldb,
                                                               float *__restrict__ C, int ldc,
                                                               const float *__restrict__ bias, int epilogue,
                                                               int kchunk, int vecA, int vecB, FoldB foldB,
                                                               FoldB foldC, double *__restrict__ colstats,
                                                               long long cslice)
{
    static_assert(WM * WN * 64 == GB_THREADS, "4 waves");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    typedef SlabB<BM, !TA> SA;
    typedef SlabB<BN, TB> SB;
    __shared__ __attribute__((aligned(16))) __bf16 ldsA[BM * GB_LDK];
    __shared__ __attribute__((aligned(16))) __bf16 ldsB[BN * GB_LDK];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles = gridDim.x * gridDim.y;
    int vid, slice;
    if (gridDim.z > 1 && (gridDim.z & 7) == 0) {
        // split K, slices a multiple of 8: all tiles of a slice on ONE XCD (see gemm_f32_kernel: the transposed
        // product of dgcnn_agg fetched 4.4 x its algorithmic bytes with the per-slice tile order)
        const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const int u = lin >> 3;
        slice = (lin & 7) + 8 * (u / tiles);
        vid = u % tiles;
    } else {
        vid = xcd_contiguous(blockIdx.y * gridDim.x + blockIdx.x, tiles);
        slice = blockIdx.z;
    }
    const int m0 = (vid / (int)gridDim.x) * BM, n0 = (vid % (int)gridDim.x) * BN;
    const int kbeg = slice * kchunk;
    const int kend = min(K, kbeg + kchunk);
    C += (size_t)slice * (size_t)cslice;        // != 0: every K slice stores its own copy (summed in slice order afterwards)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.0f;

    const int fr = lane & 31, fk = lane >> 5;
    // the slab in LDS times the accumulators: two MFMA k-steps
    auto multiply = [&]() {
#pragma unroll
        for (int s = 0; s < GB_BK / 16; ++s) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const bf16x8 *>(ldsA + ((wm * TM + i) * 32 + fr) * GB_LDK + 16 * s + 8 * fk);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const bf16x8 *>(ldsB + ((wn * TN + j) * 32 + fr) * GB_LDK + 16 * s + 8 * fk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    if (FAST) {
        // wave-uniform slab origins, advanced by one slab per iteration
        const FoldB nofold = {-1, 0};
        const float *pa = SA::fast_origin(A, lda, nofold, m0, kbeg);
        const float *pb = SB::fast_origin(B, ldb, foldB, n0, kbeg);
        const size_t stepa = TA ? (size_t)GB_BK * lda : (size_t)GB_BK;
        SA sa;
        SB sb;
        sa.fast_init(lda, nofold, m0);
        sb.fast_init(ldb, foldB, n0);
        sa.fast_load(pa, lda);
        sb.fast_load(pb, ldb);
        for (int k0 = kbeg; k0 < kend; k0 += GB_BK) {
            __syncthreads();
            sa.stage(ldsA);
            sb.stage(ldsB);
            __syncthreads();
            if (k0 + GB_BK < kend) {
                pa += stepa;
                pb = SB::fast_origin(B, ldb, foldB, n0, k0 + GB_BK);
                sa.fast_load(pa, lda);
                sb.fast_load(pb, ldb);
            }
            multiply();
        }
    } else {
        SA sa;
        SB sb;
        const FoldB nofold = {-1, 0};
        sa.load(A, lda, m0, M, kbeg, kend, vecA != 0, nofold);
        sb.load(B, ldb, n0, N, kbeg, kend, vecB != 0, foldB);
        for (int k0 = kbeg; k0 < kend; k0 += GB_BK) {
            __syncthreads();
            sa.stage(ldsA);
            sb.stage(ldsB);
            __syncthreads();
            if (k0 + GB_BK < kend) {
                sa.load(A, lda, m0, M, k0 + GB_BK, kend, vecA != 0, nofold);
                sb.load(B, ldb, n0, N, k0 + GB_BK, kend, vecB != 0, foldB);
            }
            multiply();
        }
    }

    // epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool add_bias = bias != nullptr && (epilogue != GB_ATOMIC || slice == 0);
    if (colstats != nullptr) {
        // column sums / sums of squares of this tile in fp64 for the batch norm that consumes C (as in
        // gemm_f32_kernel: colstats[tile row][0 | 1][col]; fixed order of summation)
        __shared__ double cs[2][WM][BN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cl = (wn * TN + j) * 32 + fr;
            const float bv = (add_bias && n0 + cl < N) ? bias[n0 + cl] : 0.0f;
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    if (FAST || row < M) {
                        const double v = (double)(acc[i][j][r] + bv);
                        s1 += v;
                        s2 += v * v;
                    }
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (fk == 0) {
                cs[0][wm][cl] = s1;
                cs[1][wm][cl] = s2;
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * BN; t += GB_THREADS) {
            const int which = t / BN, cl = t % BN;
            if (n0 + cl < N) {
                double v = cs[which][0][cl];
#pragma unroll
                for (int w = 1; w < WM; ++w)
                    v += cs[which][w][cl];
                colstats[((size_t)(m0 / BM) * 2 + which) * N + n0 + cl] = v;
            }
        }
    }
    if (FAST) {
        // whole tile: one per-lane pointer per column, wave-uniform row offsets
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + fr;
            // (a folded output keeps the rows of one column ldc apart: the fold only moves the column's origin)
            float *c0 = C + foldb_off(m0 + wm * TM * 32 + 4 * fk, col, ldc, foldC);
            const float bv = add_bias ? bias[col] : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (epilogue == GB_ACCUM) {
                    // the sixteen old values first, then the sums (see gemm.hip: per element every load waits behind the
                    // previous store)
                    float old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        old[r] = c0[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        c0[(size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc] = old[r] + (acc[i][j][r] + bv);
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *dst = c0 + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc;
                    const float v = acc[i][j][r] + bv;
                    if (epilogue == GB_STORE)
                        *dst = v;
                    else
                        atomicAdd(dst, v);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + fr;
        if (col >= N)
            continue;
        const float bv = add_bias ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (row < M) {
                    float *dst = C + foldb_off(row, col, ldc, foldC);
                    const float v = acc[i][j][r] + bv;
                    if (epilogue == GB_STORE)
                        *dst = v;
                    else if (epilogue == GB_ACCUM)
                        *dst = *dst + v;
                    else
                        atomicAdd(dst, v);
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool FAST>
static void launch_bf16_as(bool ta, bool tb, dim3 grid, hipStream_t s, int M, int N, int K, const float *A, int lda,
                           const float *B, int ldb, float *C, int ldc, const float *bias, int epi, int kchunk,
                           int vecA, int vecB, FoldB fb, FoldB fc, double *cs, long long cslice)
{
    dim3 block(GB_THREADS);
    if (!ta && !tb)
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, false, false, FAST>), grid, block, 0, s, M, N, K, A, lda,
                           B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb, fc, cs, cslice);
    else if (!ta && tb)
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, false, true, FAST>), grid, block, 0, s, M, N, K, A, lda,
                           B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb, fc, cs, cslice);
    else if (ta && !tb)
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, true, false, FAST>), grid, block, 0, s, M, N, K, A, lda,
                           B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb, fc, cs, cslice);
    else
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, true, true, FAST>), grid, block, 0, s, M, N, K, A, lda,
                           B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb, fc, cs, cslice);
}

template <int BM, int BN, int WM, int WN>
static void launch_bf16(bool ta, bool tb, dim3 grid, hipStream_t s, int M, int N, int K, const float *A, int lda,
                        const float *B, int ldb, float *C, int ldc, const float *bias, int epi, int kchunk,
                        int vecA, int vecB, FoldB fb, FoldB fc, double *cs, long long cslice)
{
    // whole tiles, whole slabs in every K slice, 16-byte aligned rows: the lean loop
    // (a folded k-contiguous B: every 32-wide slab inside one fold block)
    const bool fold_ok = fb.shift < 0 || !tb || (1 << fb.shift) % GB_BK == 0;
    const bool fast = M % BM == 0 && N % BN == 0 && K % GB_BK == 0 && vecA && vecB && fold_ok;
    if (fast)
        launch_bf16_as<BM, BN, WM, WN, true>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, vecA,
                                             vecB, fb, fc, cs, cslice);
    else
        launch_bf16_as<BM, BN, WM, WN, false>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, vecA,
                                              vecB, fb, fc, cs, cslice);
}

// tile shape and K slices (same policy as gemm.hip's gemm_plan, slabs of 32)
static void gemm_bf16_plan(int M, int N, int K, int &BM, int &BN, int &splits, bool ordered = false)
{
    // With bf16 operands these products are bound by what the CUs pull out of L2, so the side that is a multiple of
    // 160 but not of 128 (the 320 concat channels of dgcnn_agg: N of dX, M of dW) takes 160-wide tiles of five 32 x 32
    // accumulators per wave: the big operand (dY, 1.07 GB at B=256) is then re-read twice instead of five times
    // (measured, dW / dX: B=256 987 -> 716 / 563 -> 542 us, B=128 494 -> 372 / 288 -> 276 us, B=32 139 -> 152 / 79 -> 64 us:
    // the transposed product keeps 64-row tiles below 65536 rows of K).
    if (M <= 32) {
        BM = 32;
        BN = 128;
    } else if (N % 160 == 0 && N % 128 != 0 && M >= 1024) {
        BM = 128;
        BN = 160;
    } else if (M % 160 == 0 && M % 128 != 0 && N % 128 == 0 && K >= 65536) {
        BM = 160;
        BN = 128;
    } else if (N <= 64 || (N % 128 != 0 && N % 64 == 0)) {
        BM = 128;
        BN = 64;
    } else if (M % 128 != 0 && M % 64 == 0) {
        BM = 64;
        BN = 128;
    } else {
        BM = 128;
        BN = 128;
    }
    const long long tiles = (long long)ceil_div(M, BM) * ceil_div(N, BN);
    splits = 1;
    const int resident = 256 * (BM == 32 ? 2 : 4);
    if (tiles < 256 && K >= 256) {
        splits = (int)((tiles <= 4 ? 256 : resident) / tiles);
        const int max_splits = K / 128 > 0 ? K / 128 : 1;
        if (splits > max_splits)
            splits = max_splits;
        if (splits < 1)
            splits = 1;
        if (splits > 8)
            splits = splits / 8 * 8;      // whole slices per XCD
    }
    if (!ordered && CLOUDAAE_KNOB("CLOUDAAE_DETERMINISTIC", 0) != 0)     // deterministic mode: see gemm.hip
        splits = 1;
}

} // namespace cloudaae

using namespace cloudaae;

CLOUDAAE_API int cloudaae_gemm_bf16_splits(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0)
        return 1;
    int BM, BN, splits;
    gemm_bf16_plan(M, N, K, BM, BN, splits);
    const int kchunk = ceil_div(ceil_div(K, splits), GB_BK) * GB_BK;
    return ceil_div(K, kchunk);
}

int cloudaae::gemm_bf16_launch(const char *name, int trans_a, int trans_b, int M, int N, int K, const float *A,
                               int lda, const float *B, int ldb, float *C, int ldc, const float *bias, int accumulate,
                               int fold_b, int fold_c, hipStream_t s, double *colstats, float *ordered_ws)
{
    CLOUDAAE_REQUIRE(M >= 0 && N >= 0 && K >= 0, name, "negative size");
    if (M == 0 || N == 0)
        return 0;
    CLOUDAAE_REQUIRE(lda >= (trans_a ? M : K), name, "leading dimension too small");
    CLOUDAAE_REQUIRE(fold_b ? ldb == fold_b : ldb >= (trans_b ? K : N), name, "leading dimension too small");
    CLOUDAAE_REQUIRE(fold_c ? ldc == fold_c : ldc >= N, name, "leading dimension too small");
    CLOUDAAE_REQUIRE((fold_b & (fold_b - 1)) == 0 && (fold_c & (fold_c - 1)) == 0 && fold_b % 4 == 0 &&
                         fold_c % 4 == 0, name, "fold width must be a power of two >= 4");
    FoldB fb = {-1, 0}, fc = {-1, 0};
    if (fold_b) {
        fb.shift = __builtin_ctz((unsigned)fold_b);
        fb.rows = trans_b ? N : K;
    }
    if (fold_c) {
        fc.shift = __builtin_ctz((unsigned)fold_c);
        fc.rows = M;
    }
    int BM, BN, splits;
    gemm_bf16_plan(M, N, K, BM, BN, splits, ordered_ws != nullptr);
    const int tm = ceil_div(M, BM), tn = ceil_div(N, BN);
    CLOUDAAE_REQUIRE(tm <= 65535, name, "M too large");
    CLOUDAAE_REQUIRE(colstats == nullptr || (splits == 1 && accumulate == 0 && !fold_c), name,
                     "column statistics need an unsplit, overwriting product");
    int kchunk = K > 0 ? ceil_div(ceil_div(K, splits), GB_BK) * GB_BK : GB_BK;
    splits = K > 0 ? ceil_div(K, kchunk) : 1;
    int epi = accumulate == 1 ? GB_ACCUM : GB_STORE;
    // ordered_ws: slices kept apart and summed in slice order by a second kernel (see gemm_f32_launch)
    const bool ordered = ordered_ws != nullptr && splits > 1;
    CLOUDAAE_REQUIRE(ordered_ws == nullptr || (accumulate == 0 && !fold_c && colstats == nullptr), name,
                     "slice-ordered products overwrite an unfolded output");
    float *const Cout = C;
    const int ldc_out = ldc;
    const float *const bias_out = bias;
    long long cslice = 0;
    if (ordered) {
        C = ordered_ws;
        ldc = N;
        bias = nullptr;
        cslice = (long long)M * N;
    } else if (splits > 1) {
        epi = GB_ATOMIC;
        if (!accumulate) {
            if (fold_c)
                CLOUDAAE_CHECK_HIP(hipMemsetAsync(C, 0, sizeof(float) * (size_t)M * (size_t)N, s), name);
            else
                CLOUDAAE_CHECK_HIP(hipMemset2DAsync(C, sizeof(float) * (size_t)ldc, 0, sizeof(float) * (size_t)N,
                                                    (size_t)M, s), name);
        }
    }
    const int vecA = (((uintptr_t)A & 15) == 0 && lda % 4 == 0) ? 1 : 0;
    const int vecB = (((uintptr_t)B & 15) == 0 && ldb % 4 == 0) ? 1 : 0;
    dim3 grid(tn, tm, splits);
    const bool ta = trans_a != 0, tb = trans_b != 0;
    if (BM == 32)
        launch_bf16<32, 128, 1, 4>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb,
                                   fc, colstats, cslice);
    else if (BN == 160)
        launch_bf16<128, 160, 4, 1>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb,
                                    fc, colstats, cslice);
    else if (BM == 160)
        launch_bf16<160, 128, 1, 4>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb,
                                    fc, colstats, cslice);
    else if (BN == 64)
        launch_bf16<128, 64, 4, 1>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb,
                                   fc, colstats, cslice);
    else if (BM == 64)
        launch_bf16<64, 128, 2, 2>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb,
                                   fc, colstats, cslice);
    else
        launch_bf16<128, 128, 2, 2>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB,
                                    fb, fc, colstats, cslice);
    CLOUDAAE_CHECK_LAUNCH(name);
    if (ordered) {
        const int rc = gemm_slices_sum(name, M, N, splits, ordered_ws, Cout, ldc_out, bias_out, s);
        if (rc != 0)
            return rc;
    }
    return 0;
}

CLOUDAAE_API int cloudaae_gemm_bf16(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                                    const float *B, int ldb, float *C, int ldc, const float *bias, int accumulate,
                                    cloudaae_stream_t stream)
{
    return gemm_bf16_launch("cloudaae_gemm_bf16", trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate,
                            0, 0, (hipStream_t)stream);
}

CLOUDAAE_API long long cloudaae_gemm_bf16_ordered_workspace(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0)
        return 0;
    int BM, BN, splits;
    gemm_bf16_plan(M, N, K, BM, BN, splits, true);
    const int kchunk = ceil_div(ceil_div(K, splits), GB_BK) * GB_BK;
    splits = ceil_div(K, kchunk);
    return splits > 1 ? (long long)splits * M * N : 0;
}

CLOUDAAE_API int cloudaae_gemm_bf16_ordered(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                                            const float *B, int ldb, float *C, int ldc, const float *bias,
                                            float *workspace, long long workspace_floats, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_gemm_bf16_ordered";
    CLOUDAAE_REQUIRE(workspace != nullptr ? workspace_floats >= cloudaae_gemm_bf16_ordered_workspace(M, N, K)
                                          : cloudaae_gemm_bf16_ordered_workspace(M, N, K) == 0,
                     name, "this product is cut over K: workspace missing or smaller than cloudaae_gemm_bf16_ordered_workspace");
    static float dummy_ws;      // (non-NULL selects the ordered plan; a product that stays whole never touches it)
    return gemm_bf16_launch(name, trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, 0, 0, 0, (hipStream_t)stream,
                            nullptr, workspace != nullptr ? workspace : &dummy_ws);
}

CLOUDAAE_API int cloudaae_gemm_bf16_colstats_parts(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0)
        return 0;
    int BM, BN, splits;
    gemm_bf16_plan(M, N, K, BM, BN, splits);
    return splits == 1 ? ceil_div(M, BM) : 0;      // one row of sums per tile row
}

CLOUDAAE_API int cloudaae_gemm_bf16_colstats(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                                             const float *B, int ldb, float *C, int ldc, const float *bias,
                                             double *colstats, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_gemm_bf16_colstats";
    CLOUDAAE_REQUIRE(colstats != nullptr, name, "null argument");
    return gemm_bf16_launch(name, trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, 0, 0, 0,
                            (hipStream_t)stream, colstats);
}
