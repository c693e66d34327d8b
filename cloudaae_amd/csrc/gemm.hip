// gemm.hip -- fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Serves every dense layer of the CloudAAE path: the 1x1 convolutions
// (reference utils/tf_util.py:161-166: tf.nn.conv2d + bias_add on [B,N,k,C] rows)
// and the fully connected layers (tf_util.py:349-352: tf.matmul + bias_add),
// forward and both backward products:
//     y  = x W + b            op(A) = A      op(B) = B
//     dx = dy W^T             op(A) = A      op(B) = B^T
//     dW = x^T dy             op(A) = A^T    op(B) = B
// fp32 in, fp32 accumulate: the f32 MFMA is bitwise a k-ordered fmaf chain
// (MI355X_MICROARCH.md, matrix cores), so without split-K the result is exactly
//     c = fma(a[k], b[k], c), k ascending, from +0.
//
// Tiling is for wave64 / 32x32 MFMA tiles, not a warp-shaped port: a workgroup of
// 4 waves owns a BM x BN tile, each wave a grid of 32x32 accumulator tiles (16
// VGPRs each).  One MFMA consumes ONE fp32 per operand per lane (A[i=l&31][k=l>>5],
// B[k=l>>5][j=l&31]) and takes 64 cycles, so operand traffic is tiny next to the
// matrix pipe: operands sit in LDS in their natural global layout (k-contiguous
// rows padded to an odd stride, or row-contiguous panels) and are fetched with
// conflict-free ds_read_b32; no transposition pass is needed for any of the three
// products.  The next K-slab's global loads are issued before the current slab's
// MFMAs (register prefetch).  K can be split across workgroups (grid.z) when the
// output has too few tiles to fill 256 CUs (dW products, tiny-batch FC layers);
// slices then combine with hardware fp32 atomics.
#include <cstdlib>
#include "common.h"
#include "gemm.h"
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_BK = 16;
constexpr int GEMM_THREADS = 256;
#define GEMM_ATTR

enum { EPI_STORE = 0, EPI_ACCUM = 1, EPI_ATOMIC = 2 };

// A row-major matrix whose logical columns are FOLDED into stacked row blocks: logical (r, c) lives
// at physical row (c >> shift) * rows + r, column c & (width - 1).  That is how the edge convolution's
// [2*cin, cout] kernel looks when it is used as [cin, 2*cout] = [W_centre | W_neighbour]: with it the
// P/Q products, their dX and their dW are ONE product each instead of two (edgeconv.hip).
// shift < 0: no folding.
struct Fold {
    int shift, rows;
};
__device__ __forceinline__ size_t fold_off(int r, int c, int ld, Fold f)
{
    if (f.shift < 0)
        return (size_t)r * ld + c;
    return (size_t)((c >> f.shift) * f.rows + r) * ld + (c & ((1 << f.shift) - 1));
}

// One operand panel: ROWS "outer" indices (m for A, n for B) x BK k-values.
// KC = true : memory is [outer][k] (k contiguous)   -> LDS [ROWS][BK+1]
// KC = false: memory is [k][outer] (outer contiguous)-> LDS [BK][ROWS]
template <int ROWS, bool KC, int BK = GEMM_BK>
struct Panel {
    static constexpr int LDS_FLOATS = KC ? ROWS * (BK + 1) : BK * ROWS;
    static constexpr int VECS = ROWS * BK / 4;           // float4 per slab
    static constexpr int PER_THREAD = (VECS + GEMM_THREADS - 1) / GEMM_THREADS;

    float4v reg[PER_THREAD];

    // global -> registers for the slab starting at k0; outer0 = first outer index.
    // FAST: the whole slab is inside the matrix and 16-byte aligned (checked once at launch):
    // straight global_load_dwordx4, no per-element predicates in the loop.
    template <bool FAST>
    __device__ __forceinline__ void load(const float *__restrict__ P, int ld, int outer0, int nouter,
                                         int k0, int kend, bool vec_ok, Fold fold)
    {
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            const int v = it * GEMM_THREADS + (int)threadIdx.x;
            float4v r = {0.f, 0.f, 0.f, 0.f};
            if (FAST) {
                if (VECS % GEMM_THREADS != 0 && v >= VECS)
                    break;
                const size_t off = KC ? fold_off(outer0 + v / (BK / 4), k0 + (v % (BK / 4)) * 4, ld, fold)
                                      : fold_off(k0 + v / (ROWS / 4), outer0 + (v % (ROWS / 4)) * 4, ld, fold);
                reg[it] = *reinterpret_cast<const float4v *>(P + off);
                continue;
            } else if (VECS % GEMM_THREADS == 0 || v < VECS) {
                if (KC) {
                    const int o = v / (BK / 4), kq = v % (BK / 4);
                    const int go = outer0 + o, gk = k0 + kq * 4;
                    if (go < nouter) {
                        const float *src = P + fold_off(go, gk, ld, fold);
                        if (vec_ok && gk + 3 < kend) {
                            r = *reinterpret_cast<const float4v *>(src);
                        } else {
                            if (gk + 0 < kend) r.x = src[0];
                            if (gk + 1 < kend) r.y = src[1];
                            if (gk + 2 < kend) r.z = src[2];
                            if (gk + 3 < kend) r.w = src[3];
                        }
                    }
                } else {
                    const int kk = v / (ROWS / 4), oq = v % (ROWS / 4);
                    const int gk = k0 + kk, go = outer0 + oq * 4;
                    if (gk < kend) {
                        const float *src = P + fold_off(gk, go, ld, fold);
                        if (vec_ok && go + 3 < nouter) {
                            r = *reinterpret_cast<const float4v *>(src);
                        } else {
                            if (go + 0 < nouter) r.x = src[0];
                            if (go + 1 < nouter) r.y = src[1];
                            if (go + 2 < nouter) r.z = src[2];
                            if (go + 3 < nouter) r.w = src[3];
                        }
                    }
                }
            }
            reg[it] = r;
        }
    }

    // registers -> LDS
    __device__ __forceinline__ void stage(float *__restrict__ lds) const
    {
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            const int v = it * GEMM_THREADS + (int)threadIdx.x;
            if (VECS % GEMM_THREADS == 0 || v < VECS) {
                if (KC) {
                    const int o = v / (BK / 4), kq = v % (BK / 4);
                    float *dst = lds + o * (BK + 1) + kq * 4;
                    dst[0] = reg[it].x;
                    dst[1] = reg[it].y;
                    dst[2] = reg[it].z;
                    dst[3] = reg[it].w;
                } else {
                    const int kk = v / (ROWS / 4), oq = v % (ROWS / 4);
                    *reinterpret_cast<float4v *>(lds + kk * ROWS + oq * 4) = reg[it];
                }
            }
        }
    }

    // MFMA operand of this lane: element (outer, kk) of the staged slab
    static __device__ __forceinline__ float frag(const float *__restrict__ lds, int outer, int kk)
    {
        return KC ? lds[outer * (BK + 1) + kk] : lds[kk * ROWS + outer];
    }
};

// C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias[N])
// TA: A is stored [K][M] (lda >= M); else [M][K].  TB: B is stored [N][K]; else [K][N].
// The tile (m0, n0) of K slice `slice` (what one workgroup computes).
template <int BM, int BN, int WM, int WN, bool TA, bool TB, bool FAST, int BK = GEMM_BK>
__device__ __forceinline__ void gemm_f32_tile(
    int M, int N, int K, const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
    float *__restrict__ C, int ldc, const float *__restrict__ bias, int epilogue, int kchunk,
    int vecA, int vecB, Fold foldB, Fold foldC, double *__restrict__ colstats, int m0, int n0, int slice)
{
    static_assert(WM * WN * 64 == GEMM_THREADS, "4 waves");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;   // 32x32 tiles per wave
    typedef Panel<BM, !TA, BK> PA;
    typedef Panel<BN, TB, BK> PB;
    __shared__ float ldsA[PA::LDS_FLOATS];
    __shared__ float ldsB[PB::LDS_FLOATS];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int kbeg = slice * kchunk;
    const int kend = min(K, kbeg + kchunk);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.0f;

    PA pa;
    PB pb;
    const Fold nofold = {-1, 0};
    pa.template load<FAST>(A, lda, m0, M, kbeg, kend, vecA != 0, nofold);
    pb.template load<FAST>(B, ldb, n0, N, kbeg, kend, vecB != 0, foldB);

    const int fr = lane & 31, fk = lane >> 5;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();            // previous slab fully consumed
        pa.stage(ldsA);
        pb.stage(ldsB);
        __syncthreads();
        if (k0 + BK < kend) {  // prefetch the next slab behind the MFMAs
            pa.template load<FAST>(A, lda, m0, M, k0 + BK, kend, vecA != 0, nofold);
            pb.template load<FAST>(B, ldb, n0, N, k0 + BK, kend, vecB != 0, foldB);
        }
        // operand fragments are read one k-step ahead of the MFMAs that consume them
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a[0][i] = PA::frag(ldsA, (wm * TM + i) * 32 + fr, fk);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            b[0][j] = PB::frag(ldsB, (wn * TN + j) * 32 + fr, fk);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int c = s & 1, n = c ^ 1;
            if (s + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[n][i] = PA::frag(ldsA, (wm * TM + i) * 32 + fr, 2 * (s + 1) + fk);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[n][j] = PB::frag(ldsB, (wn * TN + j) * 32 + fr, 2 * (s + 1) + fk);
            }
            // keep the ds_reads of step s+1 AHEAD of the MFMAs of step s (the scheduler otherwise
            // sinks them next to their use and every fourth MFMA waits a full LDS round trip)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool add_bias = bias != nullptr && (epilogue != EPI_ATOMIC || slice == 0);
    if (colstats != nullptr) {
        // Column sums of this tile (values as stored, bias included) for the batch norm that consumes C:
        // colstats[tile row][0][col] = sum, [1][col] = sum of squares, in fp64 -- what bn_colsum_kernel
        // would recompute by reading all of C again (134 MB for dgcnn_agg).  Fixed order: rows of a lane,
        // the two lane halves, then the WM waves stacked along M.
        __shared__ double cs[2][WM][BN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cl = (wn * TN + j) * 32 + fr;
            const float bv = (add_bias && (FAST || n0 + cl < N)) ? bias[n0 + cl] : 0.0f;
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
        for (int t = threadIdx.x; t < 2 * BN; t += GEMM_THREADS) {
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
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + fr;
        if (!FAST && col >= N)
            continue;
        const float bv = add_bias ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (epilogue == EPI_ACCUM) {
                // all sixteen old values first, then the sums: written as "*dst = *dst + v" per element every load waits
                // behind the previous store (the compiler cannot tell the rows apart), sixteen memory round trips in a row --
                // 7 of the 20 us of the edge convolution's dX product into the concat gradient
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    old[r] = (FAST || row < M) ? C[fold_off(row, col, ldc, foldC)] : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    if (FAST || row < M)
                        C[fold_off(row, col, ldc, foldC)] = old[r] + (acc[i][j][r] + bv);
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (FAST || row < M) {
                    float *dst = C + fold_off(row, col, ldc, foldC);
                    const float v = acc[i][j][r] + bv;
                    if (epilogue == EPI_STORE)
                        *dst = v;
                    else
                        atomicAdd(dst, v);
                }
            }
        }
    }
}

// C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias[N])
// TA: A is stored [K][M] (lda >= M); else [M][K].  TB: B is stored [N][K]; else [K][N].
template <int BM, int BN, int WM, int WN, bool TA, bool TB, bool FAST, int BK = GEMM_BK>
__global__ __launch_bounds__(GEMM_THREADS) GEMM_ATTR void gemm_f32_kernel(
    int M, int N, int K, const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb,
    float *__restrict__ C, int ldc, const float *__restrict__ bias, int epilogue, int kchunk,
    int vecA, int vecB, Fold foldB, Fold foldC, double *__restrict__ colstats, long long cslice)
{
    // XCD-aware tile order: workgroups that share an A row-panel (same tile row) get
    // consecutive virtual ids inside one XCD, so the panel is fetched from HBM once per XCD
    // pass instead of once per column tile (PMC: 8x over-fetch of A without this)
    const int tiles = gridDim.x * gridDim.y;
    int vid, slice;
    if (gridDim.z > 1 && (gridDim.z & 7) == 0) {
        // split K, slices a multiple of 8: ALL tiles of a K slice go to ONE XCD (slice s -> XCD s % 8), in
        // dispatch order, so the slice's A and B panels (1.3 + 4 MB for the dgcnn_agg weight gradient) are
        // fetched from HBM once by the tiles that walk them in step instead of once per tile row / column:
        // the transposed product of dgcnn_agg fetched 784 MB against 176 MB algorithmic with the per-slice
        // tile order below (each XCD saw 5 of the 40 tiles of EVERY slice)
        const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        const int u = lin >> 3;
        slice = (lin & 7) + 8 * (u / tiles);
        vid = u % tiles;
    } else {
        vid = xcd_contiguous(blockIdx.y * gridDim.x + blockIdx.x, tiles);
        slice = blockIdx.z;
    }
    // cslice != 0: every K slice stores its own copy of C (summed in slice order by gemm_slices_sum_kernel)
    gemm_f32_tile<BM, BN, WM, WN, TA, TB, FAST, BK>(M, N, K, A, lda, B, ldb, C + (size_t)slice * (size_t)cslice, ldc, bias, epilogue, kchunk, vecA, vecB, foldB,
                                                foldC, colstats, (vid / (int)gridDim.x) * BM, (vid % (int)gridDim.x) * BN,
                                                slice);
}

// ---- several independent weight-gradient products C_j = A_j^T B_j in ONE launch ------------------------------
// The four edge-convolution layers each end their backward pass with dW = X^T [dP' | dQ]: 32768 rows reduced
// into a 24..64 x 128..256 output, split over K into ~256 slices of one 64 x 128 tile.  Alone, such a launch
// is a single wave of short workgroups -- 17-22 us of mostly latency for 4-7 us of memory traffic -- and nothing
// waits for the result before the optimiser, so the four run as one launch at the end of backward.
constexpr int GEMM_GROUP_MAX = 8;
struct GemmGroupJob {
    int M, N, K, lda, ldb, ldc, kchunk, vecA, vecB, block0, tiles_x, tiles, splits;
    const float *A, *B;
    float *C;
    Fold foldC;
};
struct GemmGroup {
    int count;
    GemmGroupJob job[GEMM_GROUP_MAX];
};

__global__ __launch_bounds__(GEMM_THREADS) void gemm_f32_tn_group_kernel(GemmGroup g)
{
    int p = 0;
    for (int i = 1; i < g.count; ++i)
        if ((int)blockIdx.x >= g.job[i].block0)
            p = i;
    const GemmGroupJob &j = g.job[p];
    const int local = (int)blockIdx.x - j.block0;
    if (local >= j.tiles * j.splits)
        return;                         // (padding up to the next multiple of 8)
    int vid, slice;
    if ((j.splits & 7) == 0) {          // (block0 is a multiple of 8: local & 7 is the XCD the workgroup runs on)
        const int u = local >> 3;
        slice = (local & 7) + 8 * (u / j.tiles);
        vid = u % j.tiles;
    } else {
        slice = local / j.tiles;
        vid = local % j.tiles;
    }
    const Fold nofold = {-1, 0};
    gemm_f32_tile<64, 128, 2, 2, true, false, false>(j.M, j.N, j.K, j.A, j.lda, j.B, j.ldb, j.C, j.ldc, nullptr, EPI_ATOMIC,
                                                     j.kchunk, j.vecA, j.vecB, nofold, j.foldC, nullptr,
                                                     (vid / j.tiles_x) * 64, (vid % j.tiles_x) * 128, slice);
}

template <int BM, int BN, int WM, int WN, bool FAST>
static void launch_fast(bool ta, bool tb, dim3 grid, hipStream_t s, int M, int N, int K, const float *A,
                        int lda, const float *B, int ldb, float *C, int ldc, const float *bias, int epi,
                        int kchunk, int vecA, int vecB, Fold fb, Fold fc, double *cs, long long cslice)
{
    dim3 block(GEMM_THREADS);
    if (!ta && !tb)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, false, FAST>), grid, block, 0, s, M, N, K,
                           A, lda, B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb, fc, cs, cslice);
    else if (!ta && tb)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, false, true, FAST>), grid, block, 0, s, M, N, K,
                           A, lda, B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb, fc, cs, cslice);
    else if (ta && !tb)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, false, FAST>), grid, block, 0, s, M, N, K,
                           A, lda, B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb, fc, cs, cslice);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, true, true, FAST>), grid, block, 0, s, M, N, K,
                           A, lda, B, ldb, C, ldc, bias, epi, kchunk, vecA, vecB, fb, fc, cs, cslice);
}

template <int BM, int BN, int WM, int WN>
static void launch_cfg(bool ta, bool tb, dim3 grid, hipStream_t s, int M, int N, int K, const float *A,
                       int lda, const float *B, int ldb, float *C, int ldc, const float *bias, int epi,
                       int kchunk, int vecA, int vecB, Fold fb, Fold fc, double *cs, long long cslice)
{
    // every tile and every K-slab whole, both operands float4-loadable: the predicate-free kernel
    const bool fast = M % BM == 0 && N % BN == 0 && K % kchunk == 0 && kchunk % GEMM_BK == 0 && vecA && vecB;
    if (fast)
        launch_fast<BM, BN, WM, WN, true>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk,
                                          vecA, vecB, fb, fc, cs, cslice);
    else
        launch_fast<BM, BN, WM, WN, false>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk,
                                           vecA, vecB, fb, fc, cs, cslice);
}

} // namespace cloudaae

namespace cloudaae {

// C[r][c] = ((ws[0][r][c] + ws[1][r][c]) + ... ) + bias[c]: the slices of a product cut over K, in slice order.
__global__ __launch_bounds__(256) void gemm_slices_sum_kernel(long long total, int N, int splits, const float *__restrict__ ws,
                                                              float *__restrict__ C, int ldc, const float *__restrict__ bias,
                                                              Fold fold)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total)
        return;
    float v = ws[i];
    for (int s = 1; s < splits; ++s)
        v += ws[(size_t)s * total + i];
    const int r = (int)(i / N), c = (int)(i % N);
    if (bias != nullptr)
        v += bias[c];
    C[fold_off(r, c, ldc, fold)] = v;
}

int gemm_slices_sum(const char *name, int M, int N, int splits, const float *ws, float *C, int ldc, const float *bias,
                    hipStream_t s, int fold_shift, int fold_rows)
{
    const long long total = (long long)M * N;
    hipLaunchKernelGGL(gemm_slices_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, N, splits, ws,
                       C, ldc, bias, Fold{fold_shift, fold_rows});
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

// Tile shape and split-K slice count of a product (shared by the launcher and the query below).
static void gemm_plan(int M, int N, int K, int &BM, int &BN, int &splits, bool ordered = false)
{
    // tile shape: small-batch FC rows -> 32-row tiles; narrow outputs -> 64 columns; a dimension
    // that is a multiple of 64 but not of 128 (the 320 concat channels of dgcnn_agg: dX has N = 320,
    // dW has M = 320) takes 64-wide tiles on that side instead of a half-empty 128 one
    if (M <= 32) {
        BM = 32;
        BN = 128;
    } else if (M <= 128) {
        // the fully connected stack at 33..128 clouds per GPU (the fused kernels of fc.hip take at most 32 rows): one or
        // two row tiles, so the parallelism has to come from N and K -- 64 x 64 tiles (six workgroups per CU) instead of
        // 128 x 128: [128 x 1024] x [1024 x 12288] 69 -> 48 us, [128 x 1024] x [1024 x 1024] 15.6 -> 10.7 us
        BM = 64;
        BN = 64;
    } else if (N % 160 == 0 && N % 128 != 0 && M >= 1024 && K >= 512) {
        // dX of dgcnn_agg (N = 320): 160-wide tiles, five 32 x 32 accumulators per wave -- 1.2 LDS fragment reads per
        // MFMA instead of 1.5 and 512 workgroups instead of 1280: 214 -> 208 us (B=32), 830 -> 807 us (B=128)
        BM = 128;
        BN = 160;
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
    // tall, short-K products (the edge convolution's: 32768 rows, K = 24..256, 64..256 columns) are
    // bound by streaming A and C, not by the matrix pipe: with 128-row tiles they make one workgroup per
    // CU, each a handful of slabs long, and nothing hides the load latency (0.9-1.8 TB/s measured).
    // Halve the tile height until there are two workgroups per CU.
    // (not the 160-wide tiles: they exist as 128 x 160 only -- a 64-row grid over them would start half the
    //  workgroups past the end of the matrix)
    if (BM == 128 && BN != 160 && M >= 1024 && K <= 512 && (long long)ceil_div(M, 128) * ceil_div(N, BN) < 512)
        BM = 64;
    // (32-row tiles for these shapes, four workgroups per CU: 11.5 vs 11.9 us at [32768 x 64] x [64 x 128] -- not worth a
    //  second rule)
    const int tm = ceil_div(M, BM), tn = ceil_div(N, BN);
    // split K until one full wave of workgroups exists (tiles * splits ~ the workgroups the chip
    // holds at once for this tile shape: registers allow 3 per CU for 128x128, 5 for 64x128, 6 for
    // 128x64), keeping >= 64 k per slice; outputs of <= 4 tiles get at most 256
    // slices: every slice adds to the SAME few thousand addresses with atomics.
    // Measured on dgcnn_agg dW (40 tiles of 64x128, K = 32768): 12 slices 247 us, 32 slices 196 us,
    // 64 slices 203 us, 128 slices 219 us.
    splits = 1;
    const long long tiles = (long long)tm * tn;
    const int resident = 256 * (BM == 32 ? 2 : (BN == 64 || BM == 64) ? (BN == 64 ? 6 : 5) : 3);   // (32-row tiles: 2 measured best)
    if (tiles < 256 && K >= 128) {
        splits = (int)((tiles <= 4 ? 256 : resident) / tiles);
        const int max_splits = K / 64 > 0 ? K / 64 : 1;
        if (splits > max_splits)
            splits = max_splits;
        if (splits > 1024)
            splits = 1024;
        if (splits < 1)
            splits = 1;
        if (splits > 8)
            splits = splits / 8 * 8;      // whole slices per XCD (the kernel then keeps a slice's tiles on one XCD)
    }
    // deterministic mode (cloudaae_set_knob("CLOUDAAE_DETERMINISTIC", 1)): a product that would add its K slices with
    // atomics stays whole (and pays with idle CUs); the slice-ordered variant (cloudaae_gemm_f32_ordered) keeps its cut
    if (!ordered && CLOUDAAE_KNOB("CLOUDAAE_DETERMINISTIC", 0) != 0)
        splits = 1;
}

} // namespace cloudaae

using namespace cloudaae;

CLOUDAAE_API int cloudaae_gemm_f32_splits(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0)
        return 1;
    int BM, BN, splits;
    gemm_plan(M, N, K, BM, BN, splits);
    const int kchunk = ceil_div(ceil_div(K, splits), GEMM_BK) * GEMM_BK;
    return ceil_div(K, kchunk);
}

// The launcher behind cloudaae_gemm_f32 and the folded products of edgeconv.hip.
// fold_b / fold_c: 0, or the power-of-two width at which B's / C's logical columns fold into stacked
// row blocks (see Fold); the folded matrix has leading dimension == width.
int cloudaae::gemm_f32_launch(const char *name, int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                              const float *B, int ldb, float *C, int ldc, const float *bias, int accumulate,
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
    Fold fb = {-1, 0}, fc = {-1, 0};
    if (fold_b) {           // B's folded index: n for [K][N] storage, k for [N][K] storage
        fb.shift = __builtin_ctz((unsigned)fold_b);
        fb.rows = trans_b ? N : K;
    }
    if (fold_c) {
        fc.shift = __builtin_ctz((unsigned)fold_c);
        fc.rows = M;
    }

    int BM, BN, splits;
    gemm_plan(M, N, K, BM, BN, splits, ordered_ws != nullptr);
    const int tm = ceil_div(M, BM), tn = ceil_div(N, BN);
    CLOUDAAE_REQUIRE(tm <= 65535, name, "M too large");
    CLOUDAAE_REQUIRE(colstats == nullptr || (splits == 1 && accumulate == 0 && !fold_c), name,
                     "column statistics need an unsplit, overwriting product");
    int kchunk = K > 0 ? ceil_div(ceil_div(K, splits), GEMM_BK) * GEMM_BK : GEMM_BK;
    splits = K > 0 ? ceil_div(K, kchunk) : 1;
    // accumulate: 0 = overwrite C, 1 = add to C, 2 = C is known to hold zeros (the caller cleared
    // a whole gradient buffer once): plain stores when K is not split, atomics WITHOUT the clear
    // pass when it is
    int epi = accumulate == 1 ? EPI_ACCUM : EPI_STORE;
    // ordered_ws: a product cut over K keeps its slices apart -- slice s stores its [M, N] result at
    // ordered_ws + s M N -- and a second kernel sums them in slice order (bit-reproducible, unlike the atomics)
    const bool ordered = ordered_ws != nullptr && splits > 1;
    CLOUDAAE_REQUIRE(ordered_ws == nullptr || (accumulate == 0 && colstats == nullptr), name,
                     "slice-ordered products overwrite their output");
    float *const Cout = C;
    const int ldc_out = ldc;
    const float *const bias_out = bias;
    long long cslice = 0;
    const Fold fc_out = fc;
    if (ordered) {
        C = ordered_ws;
        ldc = N;
        bias = nullptr;
        cslice = (long long)M * N;
        fc = Fold{-1, 0};               // the slices are plain [M, N] blocks; the sum kernel folds the output
    } else if (splits > 1) {
        epi = EPI_ATOMIC;
        if (!accumulate) {  // slices add into a zeroed output
            if (fold_c)     // the folded output is one contiguous [N/width * M][width] block
                CLOUDAAE_CHECK_HIP(hipMemsetAsync(C, 0, sizeof(float) * (size_t)M * (size_t)N, s), name);
            else
                CLOUDAAE_CHECK_HIP(hipMemset2DAsync(C, sizeof(float) * (size_t)ldc, 0, sizeof(float) * (size_t)N,
                                                    (size_t)M, s), name);
        }
    }
    const bool a16 = ((uintptr_t)A & 15) == 0 && lda % 4 == 0;
    const bool b16 = ((uintptr_t)B & 15) == 0 && ldb % 4 == 0;
    const int vecA = a16 ? 1 : 0, vecB = b16 ? 1 : 0;
    dim3 grid(tn, tm, splits);
    const bool ta = trans_a != 0, tb = trans_b != 0;
    if (BM == 32)
        launch_cfg<32, 128, 1, 4>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk,
                                  vecA, vecB, fb, fc, colstats, cslice);
    else if (BN == 160)
        launch_cfg<128, 160, 4, 1>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk,
                                   vecA, vecB, fb, fc, colstats, cslice);
    else if (BM == 64 && BN == 64)
        launch_cfg<64, 64, 2, 2>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk,
                                 vecA, vecB, fb, fc, colstats, cslice);
    else if (BN == 64)
        launch_cfg<128, 64, 4, 1>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk,
                                  vecA, vecB, fb, fc, colstats, cslice);
    else if (BM == 64)
        launch_cfg<64, 128, 2, 2>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk,
                                  vecA, vecB, fb, fc, colstats, cslice);
    else
        launch_cfg<128, 128, 2, 2>(ta, tb, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk,
                                   vecA, vecB, fb, fc, colstats, cslice);
    CLOUDAAE_CHECK_LAUNCH(name);
    if (ordered) {
        const int rc = gemm_slices_sum(name, M, N, splits, ordered_ws, Cout, ldc_out, bias_out, s, fc_out.shift, fc_out.rows);
        if (rc != 0)
            return rc;
    }
    return 0;
}

CLOUDAAE_API int cloudaae_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float *A,
                                   int lda, const float *B, int ldb, float *C, int ldc,
                                   const float *bias, int accumulate, cloudaae_stream_t stream)
{
    return gemm_f32_launch("cloudaae_gemm_f32", trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate,
                           0, 0, (hipStream_t)stream);
}

CLOUDAAE_API long long cloudaae_gemm_f32_ordered_workspace(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0)
        return 0;
    int BM, BN, splits;
    gemm_plan(M, N, K, BM, BN, splits, true);
    const int kchunk = ceil_div(ceil_div(K, splits), GEMM_BK) * GEMM_BK;
    splits = ceil_div(K, kchunk);
    return splits > 1 ? (long long)splits * M * N : 0;
}

CLOUDAAE_API int cloudaae_gemm_f32_ordered(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                                           const float *B, int ldb, float *C, int ldc, const float *bias,
                                           float *workspace, long long workspace_floats, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_gemm_f32_ordered";
    // (the cut is derived again at every launch, also from development knobs: a buffer sized by an earlier query must
    //  still cover it)
    CLOUDAAE_REQUIRE(workspace != nullptr ? workspace_floats >= cloudaae_gemm_f32_ordered_workspace(M, N, K)
                                          : cloudaae_gemm_f32_ordered_workspace(M, N, K) == 0,
                     name, "this product is cut over K: workspace missing or smaller than cloudaae_gemm_f32_ordered_workspace");
    static float dummy_ws;      // (a product that stays whole never touches it; non-NULL selects the ordered plan)
    return gemm_f32_launch(name, trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, 0, 0, 0, (hipStream_t)stream,
                           nullptr, workspace != nullptr ? workspace : &dummy_ws);
}

// the same with the output's logical columns folded into stacked row blocks of width fold_c (0: none; the edge convolution's
// [2*cin, cout] kernel addressed as [cin, 2*cout], see gemm.h): the deterministic mode's weight-gradient products
CLOUDAAE_API int cloudaae_gemm_f32_ordered_fold(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                                                const float *B, int ldb, float *C, int ldc, int fold_c, float *workspace,
                                                long long workspace_floats, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_gemm_f32_ordered_fold";
    CLOUDAAE_REQUIRE(workspace != nullptr ? workspace_floats >= cloudaae_gemm_f32_ordered_workspace(M, N, K)
                                          : cloudaae_gemm_f32_ordered_workspace(M, N, K) == 0,
                     name, "this product is cut over K: workspace missing or smaller than cloudaae_gemm_f32_ordered_workspace");
    static float dummy_ws;      // (a product that stays whole never touches it; non-NULL selects the ordered plan)
    return gemm_f32_launch(name, trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, 0, fold_c,
                           (hipStream_t)stream, nullptr, workspace != nullptr ? workspace : &dummy_ws);
}

CLOUDAAE_API int cloudaae_gemm_f32_colstats_parts(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0)
        return 0;
    int BM, BN, splits;
    gemm_plan(M, N, K, BM, BN, splits);
    return splits == 1 ? ceil_div(M, BM) : 0;      // one row of sums per tile row
}

CLOUDAAE_API int cloudaae_gemm_f32_colstats(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                                            const float *B, int ldb, float *C, int ldc, const float *bias,
                                            double *colstats, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_gemm_f32_colstats";
    CLOUDAAE_REQUIRE(colstats != nullptr, name, "null argument");
    return gemm_f32_launch(name, trans_a, trans_b, M, N, K, A, lda, B, ldb, C, ldc, bias, 0, 0, 0,
                           (hipStream_t)stream, colstats);
}

// Several weight-gradient products C_j (+)= A_j^T B_j in one launch (struct cloudaae_gemm_tn_job).  Every C_j is
// added to with atomics: it must hold zeros (zeroed != 0: the caller cleared it; else this call clears it first).
CLOUDAAE_API int cloudaae_gemm_f32_tn_group(int count, const cloudaae_gemm_tn_job *jobs, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_gemm_f32_tn_group";
    CLOUDAAE_REQUIRE(count >= 1 && count <= GEMM_GROUP_MAX && jobs != nullptr, name, "1 to 8 products per launch");
    hipStream_t s = (hipStream_t)stream;
    GemmGroup g = {};
    g.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const cloudaae_gemm_tn_job &q = jobs[i];
        CLOUDAAE_REQUIRE(q.M > 0 && q.N > 0 && q.K > 0 && q.A && q.B && q.C, name, "bad product");
        CLOUDAAE_REQUIRE(q.lda >= q.M && q.ldb >= q.N, name, "leading dimension too small");
        CLOUDAAE_REQUIRE(q.fold_c ? (q.ldc == q.fold_c && (q.fold_c & (q.fold_c - 1)) == 0 && q.fold_c % 4 == 0)
                                  : q.ldc >= q.N, name, "bad output layout");
        GemmGroupJob &j = g.job[i];
        j.M = q.M; j.N = q.N; j.K = q.K; j.lda = q.lda; j.ldb = q.ldb; j.ldc = q.ldc;
        j.A = q.A; j.B = q.B; j.C = q.C;
        j.foldC.shift = q.fold_c ? __builtin_ctz((unsigned)q.fold_c) : -1;
        j.foldC.rows = q.M;
        j.tiles_x = ceil_div(q.N, 128);
        j.tiles = j.tiles_x * ceil_div(q.M, 64);
        // K slices: one wave of workgroups for the whole group (5 resident per CU), >= 256 k each, whole slices per XCD.
        // (every slice adds its whole tile to the same 32 KB with atomics: at K = 32768 the 320 slices of 112 k that fill
        //  the chip take 53 us, 128 of 256 k 44 us (step 1.590 -> 1.582 ms); at K = 131072 320 slices of 416 k stay best:
        //  profiles/notes_gemm_f32.md)
        const int kmin = 256;
        int splits = ((256 * 5) / count) / j.tiles;
        if (splits > q.K / kmin)
            splits = q.K / kmin;
        if (splits > 8)
            splits = splits / 8 * 8;
        if (splits < 1 || CLOUDAAE_KNOB("CLOUDAAE_DETERMINISTIC", 0) != 0)
            splits = 1;
        j.kchunk = ceil_div(ceil_div(q.K, splits), GEMM_BK) * GEMM_BK;
        j.splits = ceil_div(q.K, j.kchunk);
        j.vecA = (((uintptr_t)q.A & 15) == 0 && q.lda % 4 == 0) ? 1 : 0;
        j.vecB = (((uintptr_t)q.B & 15) == 0 && q.ldb % 4 == 0) ? 1 : 0;
        j.block0 = blocks;
        blocks += ceil_div(j.tiles * j.splits, 8) * 8;      // (jobs start on a multiple of 8: see the kernel)
        if (!q.zeroed) {
            if (q.fold_c)
                CLOUDAAE_CHECK_HIP(hipMemsetAsync(q.C, 0, sizeof(float) * (size_t)q.M * (size_t)q.N, s), name);
            else
                CLOUDAAE_CHECK_HIP(hipMemset2DAsync(q.C, sizeof(float) * (size_t)q.ldc, 0, sizeof(float) * (size_t)q.N,
                                                    (size_t)q.M, s), name);
        }
    }
    hipLaunchKernelGGL(gemm_f32_tn_group_kernel, dim3(blocks), dim3(GEMM_THREADS), 0, s, g);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}
