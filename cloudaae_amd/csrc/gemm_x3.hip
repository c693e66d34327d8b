// gemm_x3.hip -- fp32 products on the bf16 matrix cores by error-free splitting (opt-in: gemm_dtype "bf16x3").
//
// Every fp32 operand element is split, exactly, into three bfloat16 pieces v = h + m + l (h = bf16(v), m = bf16(v - h),
// l = bf16(v - h - m): 3 x 8 significand bits = the 24 of an fp32; both subtractions are exact), and a product a*b is
// taken as the six piece products of weight >= 2^-16
//     a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_m b_m + a_l b_h),
// each exact in fp32, accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  What is dropped (a_m b_l, a_l b_m, a_l b_l) is
// below 2^-23 of |a b| -- the size of ONE fp32 rounding -- so a K-long dot product carries the error of an fp32
// accumulation in a different order, not that of a bf16 product (measured against float64 in the tests next to
// cloudaae_gemm_f32: the same 1e-6-level agreement).  Six bf16 MFMAs of 32 cycles cover 16 k; the fp32 MFMA
// (v_mfma_f32_32x32x2_f32, 64 cycles for 2 k) needs 512 cycles for the same 16 k: 2.7 x less matrix-pipe time for the
// three dgcnn_agg products (reference utils/tf_util.py:161-166 and its two gradient products), which are MFMA-bound in
// fp32.  Not the default: BASELINE configs[1] is an fp32 configuration and the step's `dtype` stays what it computes in.
//
// Kernel shape as gemm_bf16.hip's lean loop (whole tiles, whole slabs of 32 k, 16-byte aligned rows; no folded operands):
// 4 waves own a BM x BN tile, the three planes of each operand sit in LDS as [plane][row][32 k + 8 pad] bf16, the next
// slab is prefetched through registers as fp32 and split on its way into LDS.
#include "common.h"
#include "gemm.h"
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int X3_BK = 32;
constexpr int X3_LDK = 40;
constexpr int X3_THREADS = 256;

enum { X3_STORE = 0, X3_ACCUM = 1, X3_ATOMIC = 2 };

// v = h + m + l exactly (each a bfloat16, round to nearest even)
// The leading piece saturates at the largest finite bfloat16 (|v| >= 0x7f7f8000 would round to infinity and the remainder
// to -infinity): every FINITE v is split exactly.  A non-finite v gives NaN pieces (inf - inf), so every result it
// contributes to is NaN -- where an fp32 product gives +-inf or NaN.
constexpr float X3_H_MAX = __builtin_bit_cast(float, 0x7f7f7fffu);      // the largest fp32 that rounds to a finite bfloat16
__device__ __forceinline__ void split3(float v, __bf16 &h, __bf16 &m, __bf16 &l)
{
    h = (__bf16)__builtin_amdgcn_fmed3f(v, -X3_H_MAX, X3_H_MAX);
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}
__device__ __forceinline__ unsigned pack_pair(__bf16 lo, __bf16 hi)
{
    const bf16x2 p = {lo, hi};
    unsigned w;
    __builtin_memcpy(&w, &p, 4);
    return w;
}

// One operand slab: ROWS outer indices x 32 k, fp32 in registers.  KC: memory is [outer][k]; else [k][outer].
template <int ROWS, bool KC>
struct Slab3 {
    static constexpr int ITEMS = KC ? ROWS * (X3_BK / 4) : (X3_BK / 2) * (ROWS / 4);
    static constexpr int PER = (ITEMS + X3_THREADS - 1) / X3_THREADS;
    static constexpr int PLANE = ROWS * X3_LDK;          // bf16 per plane
    float4v r0[2][PER], r1[2][KC ? 1 : PER];      // two register sets: two slabs in flight where the kernel asks for it
    unsigned boff[PER];
    static_assert(ROWS % 32 == 0, "tile sides are multiples of 32");
    // [k][outer] items: a half wave covers 4 k-pairs x 8 groups of four outer indices (see gemm_bf16.hip: SlabB)
    static __device__ __forceinline__ int item_kp(int it) { return ((it >> 5) & 3) * 4 + (it & 3); }
    static __device__ __forceinline__ int item_oq(int it) { return (it >> 7) * 8 + ((it >> 2) & 7); }
    static __device__ __forceinline__ bool live(int it) { return ITEMS % X3_THREADS == 0 || it < ITEMS; }

    __device__ __forceinline__ void init(int ld)
    {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = u * X3_THREADS + (int)threadIdx.x;
            if (KC)
                boff[u] = 4u * (unsigned)((it / (X3_BK / 4)) * ld + 4 * (it % (X3_BK / 4)));
            else
                boff[u] = 4u * (unsigned)(2 * item_kp(it) * ld + 4 * item_oq(it));
        }
    }
    template <int SET>
    __device__ __forceinline__ void load(const float *__restrict__ P0, int ld)
    {
        const char *base = reinterpret_cast<const char *>(P0);
        const char *base1 = reinterpret_cast<const char *>(P0 + ld);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = u * X3_THREADS + (int)threadIdx.x;
            if (live(it)) {
                r0[SET][u] = *reinterpret_cast<const float4v *>(base + boff[u]);
                if (!KC)
                    r1[SET][u] = *reinterpret_cast<const float4v *>(base1 + boff[u]);
            }
        }
    }
    template <int SET>
    __device__ __forceinline__ void stage(__bf16 *__restrict__ lds) const
    {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = u * X3_THREADS + (int)threadIdx.x;
            if (!live(it))
                continue;
            __bf16 a[3][4];
            split3(r0[SET][u].x, a[0][0], a[1][0], a[2][0]);
            split3(r0[SET][u].y, a[0][1], a[1][1], a[2][1]);
            split3(r0[SET][u].z, a[0][2], a[1][2], a[2][2]);
            split3(r0[SET][u].w, a[0][3], a[1][3], a[2][3]);
            if (KC) {
                const int o = it / (X3_BK / 4), kq = it % (X3_BK / 4);
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const unsigned w0 = pack_pair(a[p][0], a[p][1]), w1 = pack_pair(a[p][2], a[p][3]);
                    const uint2 w = {w0, w1};
                    *reinterpret_cast<uint2 *>(lds + p * PLANE + o * X3_LDK + 4 * kq) = w;
                }
            } else {
                __bf16 b[3][4];
                split3(r1[SET][u].x, b[0][0], b[1][0], b[2][0]);
                split3(r1[SET][u].y, b[0][1], b[1][1], b[2][1]);
                split3(r1[SET][u].z, b[0][2], b[1][2], b[2][2]);
                split3(r1[SET][u].w, b[0][3], b[1][3], b[2][3]);
                const int kp = item_kp(it), oq = item_oq(it);
                const int rot = (oq >> 1) & 3;          // rows written in a rotated order: all 32 banks
                const bool r1_ = (rot & 1) != 0, r2_ = (rot & 2) != 0;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const unsigned p0 = pack_pair(a[p][0], b[p][0]), p1 = pack_pair(a[p][1], b[p][1]),
                                   p2 = pack_pair(a[p][2], b[p][2]), p3 = pack_pair(a[p][3], b[p][3]);
                    const unsigned t0 = r1_ ? p1 : p0, t1 = r1_ ? p2 : p1, t2 = r1_ ? p3 : p2, t3 = r1_ ? p0 : p3;
                    const unsigned q0 = r2_ ? t2 : t0, q1 = r2_ ? t3 : t1, q2 = r2_ ? t0 : t2, q3 = r2_ ? t1 : t3;
                    __bf16 *dst = lds + p * PLANE + (4 * oq) * X3_LDK + 2 * kp;
                    *reinterpret_cast<unsigned *>(dst + ((0 + rot) & 3) * X3_LDK) = q0;
                    *reinterpret_cast<unsigned *>(dst + ((1 + rot) & 3) * X3_LDK) = q1;
                    *reinterpret_cast<unsigned *>(dst + ((2 + rot) & 3) * X3_LDK) = q2;
                    *reinterpret_cast<unsigned *>(dst + ((3 + rot) & 3) * X3_LDK) = q3;
                }
            }
        }
    }
};

template <int BM, int BN>
constexpr int x3_lds_bytes() { return 3 * (BM + BN) * X3_LDK * 2; }

// C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias[N]), fp32 in and out, split products
template <int BM, int BN, int WM, int WN, bool TA, bool TB, int DEPTH>
__global__ __launch_bounds__(X3_THREADS, 2) void gemm_x3_kernel(int M, int N, int K, const float *__restrict__ A, int lda,
                                                             const float *__restrict__ B, int ldb, float *__restrict__ C,
                                                             int ldc, const float *__restrict__ bias, int epilogue,
                                                             int kchunk, double *__restrict__ colstats)
{
    static_assert(WM * WN * 64 == X3_THREADS, "4 waves");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    typedef Slab3<BM, !TA> SA;
    typedef Slab3<BN, TB> SB;
    extern __shared__ __attribute__((aligned(16))) unsigned char x3_lds[];
    __bf16 *ldsA = reinterpret_cast<__bf16 *>(x3_lds);
    __bf16 *ldsB = ldsA + 3 * SA::PLANE;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles = gridDim.x * gridDim.y;
    int vid, slice;
    if (gridDim.z > 1 && (gridDim.z & 7) == 0) {      // all tiles of a K slice on one XCD (see gemm_bf16_kernel)
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
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.0f;

    const float *pa = A + (TA ? (size_t)kbeg * lda + m0 : (size_t)m0 * lda + kbeg);
    const float *pb = B + (TB ? (size_t)n0 * ldb + kbeg : (size_t)kbeg * ldb + n0);
    const size_t stepa = TA ? (size_t)X3_BK * lda : (size_t)X3_BK;
    const size_t stepb = TB ? (size_t)X3_BK : (size_t)X3_BK * ldb;
    SA sa;
    SB sb;
    sa.init(lda);
    sb.init(ldb);
    const int fr = lane & 31, fk = lane >> 5;
    auto multiply = [&]() {
#pragma unroll
        for (int s = 0; s < X3_BK / 16; ++s) {
            bf16x8 a[3][TM], b[3][TN];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[p][i] = *reinterpret_cast<const bf16x8 *>(ldsA + p * SA::PLANE + ((wm * TM + i) * 32 + fr) * X3_LDK +
                                                                16 * s + 8 * fk);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[p][j] = *reinterpret_cast<const bf16x8 *>(ldsB + p * SB::PLANE + ((wn * TN + j) * 32 + fr) * X3_LDK +
                                                                16 * s + 8 * fk);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // smallest pieces first (they meet an accumulator that already holds the earlier k anyway)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
                }
        }
    };
    if (DEPTH == 2) {
        // two slabs in flight: the slab staged in iteration i was requested in iteration i - 2; the two register sets
        // alternate, so the loop is unrolled by two.  Every iteration issues its loads UNCONDITIONALLY (past the end of
        // the K range it asks for the last slab again, which nobody consumes): with a branch around a load the
        // compiler's wait-count bookkeeping has to assume the path on which the load was not issued, and then waits
        // for the most recent loads where the older ones were meant -- which is one slab in flight again.
        const int nslab = (kend - kbeg) / X3_BK;
        auto slab_a = [&](int i) { return pa + (size_t)(i < nslab ? i : nslab - 1) * stepa; };
        auto slab_b = [&](int i) { return pb + (size_t)(i < nslab ? i : nslab - 1) * stepb; };
        sa.template load<0>(slab_a(0), lda);
        sb.template load<0>(slab_b(0), ldb);
        // (the first set's loads must all be OLDER than the second set's, here as in the loop: interleaved by the
        // scheduler, the loop's first wait has to cover the second set on the entry path, and the compiler then uses
        // that count on the back edge too)
        __builtin_amdgcn_sched_barrier(0);
        sa.template load<1>(slab_a(1), lda);
        sb.template load<1>(slab_b(1), ldb);
        __builtin_amdgcn_sched_barrier(0);
        for (int i = 0; i < nslab; i += 2) {
            __syncthreads();
            sa.template stage<0>(ldsA);
            sb.template stage<0>(ldsB);
            __syncthreads();
            sa.template load<0>(slab_a(i + 2), lda);
            sb.template load<0>(slab_b(i + 2), ldb);
            __builtin_amdgcn_sched_barrier(0);
            multiply();
            __builtin_amdgcn_sched_barrier(0);
            // (the launcher takes this variant only when every K slice holds an even number of slabs)
            __syncthreads();
            sa.template stage<1>(ldsA);
            sb.template stage<1>(ldsB);
            __syncthreads();
            sa.template load<1>(slab_a(i + 3), lda);
            sb.template load<1>(slab_b(i + 3), ldb);
            __builtin_amdgcn_sched_barrier(0);
            multiply();
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        sa.template load<0>(pa, lda);
        sb.template load<0>(pb, ldb);
        for (int k0 = kbeg; k0 < kend; k0 += X3_BK) {
            __syncthreads();
            sa.template stage<0>(ldsA);
            sb.template stage<0>(ldsB);
            __syncthreads();
            if (k0 + X3_BK < kend) {
                pa += stepa;
                pb += stepb;
                sa.template load<0>(pa, lda);
                sb.template load<0>(pb, ldb);
            }
            multiply();
        }
    }

    // epilogue: lane holds column (lane & 31), rows (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool add_bias = bias != nullptr && (epilogue != X3_ATOMIC || slice == 0);
    if (colstats != nullptr) {
        // column sums / sums of squares of this tile in fp64 (as gemm_f32_kernel); the staging array lies over the slabs
        __syncthreads();
        double (*cs)[WM][BN] = reinterpret_cast<double (*)[WM][BN]>(x3_lds);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cl = (wn * TN + j) * 32 + fr;
            const float bv = add_bias ? bias[n0 + cl] : 0.0f;
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double v = (double)(acc[i][j][r] + bv);
                    s1 += v;
                    s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (fk == 0) {
                cs[0][wm][cl] = s1;
                cs[1][wm][cl] = s2;
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * BN; t += X3_THREADS) {
            const int which = t / BN, cl = t % BN;
            double v = cs[which][0][cl];
#pragma unroll
            for (int w = 1; w < WM; ++w)
                v += cs[which][w][cl];
            colstats[((size_t)(m0 / BM) * 2 + which) * N + n0 + cl] = v;
        }
    }
    float *c0 = C + (size_t)(m0 + wm * TM * 32 + 4 * fk) * ldc + (n0 + wn * TN * 32 + fr);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float bv = add_bias ? bias[n0 + (wn * TN + j) * 32 + fr] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *dst = c0 + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + j * 32;
                const float v = acc[i][j][r] + bv;
                if (epilogue == X3_STORE)
                    *dst = v;
                else if (epilogue == X3_ACCUM)
                    *dst = *dst + v;
                else
                    atomicAdd(dst, v);
            }
        }
    }
}

// tile shape and K slices; false when the product is not one this file serves
static bool gemm_x3_plan(int ta, int tb, int M, int N, int K, int &BM, int &BN, int &splits)
{
    if (M <= 0 || N <= 0 || K <= 0 || K % X3_BK != 0 || (ta && tb))
        return false;
    BM = (ta && M % 160 == 0 && M % 128 != 0) ? 160 : 128;
    BN = (!ta && tb && N % 160 == 0 && N % 128 != 0) ? 160 : 128;
    if (M % BM != 0 || N % BN != 0)
        return false;
    const long long tiles = (long long)(M / BM) * (N / BN);
    splits = 1;
    if (tiles < 256 && K >= 256) {       // fill the chip (two workgroups per CU), whole slices per XCD
        splits = (int)((tiles <= 4 ? 256 : 512) / tiles);
        const int max_splits = K / 128 > 0 ? K / 128 : 1;
        if (splits > max_splits)
            splits = max_splits;
        if (splits < 1)
            splits = 1;
        if (splits > 8)
            splits = splits / 8 * 8;
    }
    return true;
}

template <int BM, int BN, int WM, int WN, bool TA, bool TB, int DEPTH>
static int launch_x3(const char *name, dim3 grid, hipStream_t s, int M, int N, int K, const float *A, int lda, const float *B,
                     int ldb, float *C, int ldc, const float *bias, int epi, int kchunk, double *cs)
{
    constexpr int bytes = x3_lds_bytes<BM, BN>();
    static bool raised[64] = {};      // (per device: more than 64 KB of dynamic LDS needs the attribute)
    int dev = 0;
    CLOUDAAE_CHECK_HIP(hipGetDevice(&dev), name);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        CLOUDAAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_x3_kernel<BM, BN, WM, WN, TA, TB, DEPTH>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes), name);
        raised[dev] = true;
    }
    hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, WM, WN, TA, TB, DEPTH>), grid, dim3(X3_THREADS), bytes, s, M, N, K, A, lda, B, ldb, C,
                       ldc, bias, epi, kchunk, cs);
    return 0;
}


// =====================================================================================================================
// Second generation (round 4): "streamed" products -- the big operand is split in REGISTERS, the small one ONCE.
//
// What the kernel above pays per 32 k: every element of both operands goes global -> registers -> seven VALU operations
// -> three LDS planes (ds_write) -> barrier -> fragment reads, in phases that no wave overlaps with its own MFMAs
// (forward product at B = 32: 146 us against a matrix-pipe floor of 52 us; 114 us with the MFMAs removed).  The two
// products with a k-contiguous big operand, y = x W (utils/tf_util.py:161-166) and dx = dy W^T, are reorganised:
//   * the SMALL operand (the weight, 320 x 1024) is split once per step by x3_split_kernel into bf16 planes laid out
//     the way the matrix cores read them ([K/16 step][plane][row][16 k], the two 16-byte units of a row swapped in
//     every other group of eight rows), so a workgroup's share of a step is linear 1 KB pieces that go global -> LDS by
//     DMA (global_load_lds_dwordx4): no VALU, no ds_write, no registers; a ring of three steps;
//   * the BIG operand stays fp32 and also goes global -> LDS by DMA, in whole 128-byte lines (8 rows x 128 B per wave
//     instruction, the 16-byte units of a row permuted on the SOURCE side so that the fragment reads are conflict
//     free); each wave reads the 8 consecutive k of ITS OWN rows (two ds_read_b128) and splits them in registers into
//     the three MFMA operands -- once per element (the four waves of a workgroup are stacked along M, so no two waves
//     share a row), between the MFMAs of the previous 16 k.  Its rows are private to the wave, so there is ONE buffer,
//     refilled in place as soon as the wave has read the slab's second half;
//   * one barrier per 16 k, in the middle of the step's MFMAs: behind it the next step's planes are visible and the DMA
//     of the step after goes out;
//   * two workgroups per CU (<= 256 registers, <= 80 KB of LDS): two waves per SIMD, so one wave's splitting, LDS reads,
//     DMA issue, barrier waits and epilogue stores sit under the other's MFMAs.
// Per 16 k and wave (64 x 128 tile): 48 MFMAs, 16 ds_read_b128, ~100 VALU.
// Arithmetic is unchanged: the same six piece products in the same order per accumulator, so results are bit-identical
// to the kernel above for a product that is not cut over K.
//
// (Finite / non-finite operands: see split3 above -- the same pieces here.)
__device__ __forceinline__ void split8(const float4v &x, const float4v &y, bf16x8 &h, bf16x8 &m, bf16x8 &l)
{
    const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hh = (__bf16)__builtin_amdgcn_fmed3f(v[e], -X3_H_MAX, X3_H_MAX);
        const float r1 = v[e] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        h[e] = hh;
        m[e] = mm;
        l[e] = (__bf16)(r1 - (float)mm);
    }
}

// planes of a [rows][K] operand: element (n, k), piece p at  ((k / 16 * 3 + p) * rows + n) * 16 + ((k % 16 / 8) ^ (n >> 3 & 1)) * 8 + k % 8
// src: [rows][K] (transposed == 0) or [K][rows] (transposed != 0), leading dimension ld
// (two jobs per launch: blockIdx.y picks (rows, K, transposed, planes) -- a weight is split for its forward and its
//  backward product at once)
__global__ __launch_bounds__(256) void x3_split_kernel(int rows0, int K0, int tr0, __bf16 *__restrict__ planes0, int rows1, int K1,
                                                       int tr1, __bf16 *__restrict__ planes1, const float *__restrict__ src, int ld)
{
    const bool second = blockIdx.y != 0;
    const int rows = second ? rows1 : rows0, K = second ? K1 : K0, transposed = second ? tr1 : tr0;
    __bf16 *__restrict__ planes = second ? planes1 : planes0;
    const int units = K >> 3;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)rows * units)
        return;
    int n, u;
    float4v x, y;
    if (transposed) {       // threads along n: every k row is read coalesced
        n = (int)(t % rows);
        u = (int)(t / rows);
        const float *s = src + (size_t)(8 * u) * ld + n;
        x = float4v{s[0], s[(size_t)ld], s[(size_t)2 * ld], s[(size_t)3 * ld]};
        y = float4v{s[(size_t)4 * ld], s[(size_t)5 * ld], s[(size_t)6 * ld], s[(size_t)7 * ld]};
    } else {
        u = (int)(t % units);
        n = (int)(t / units);
        const float4v *s = reinterpret_cast<const float4v *>(src + (size_t)n * ld + 8 * u);
        x = s[0];
        y = s[1];
    }
    bf16x8 h, m, l;
    split8(x, y, h, m, l);
    const int step = u >> 1, q = (u & 1) ^ ((n >> 3) & 1);
    __bf16 *d = planes + ((size_t)step * 3 * rows + n) * 16 + q * 8;
    *reinterpret_cast<bf16x8 *>(d) = h;
    *reinterpret_cast<bf16x8 *>(d + (size_t)rows * 16) = m;
    *reinterpret_cast<bf16x8 *>(d + (size_t)2 * rows * 16) = l;
}

template <int TM, int TN, int AS>
struct X3S {
    static constexpr int BM = 128 * TM, BN = 32 * TN;
    static constexpr int A_WAVE = 32 * TM * 128;        // bytes of one wave's rows, one slab of 32 k (fp32)
    static constexpr int A_STAGE = 4 * A_WAVE;
    static constexpr int A_BYTES = AS * A_STAGE;        // AS = 1: one buffer refilled in place; 2: two slabs in flight
    static constexpr int P_PLANE = BN * 32;             // bytes of one plane of the small operand, one step of 16 k
    static constexpr int P_STAGE = 3 * P_PLANE;
    static constexpr int P_PIECES = 3 * TN;             // 1 KB DMA pieces per step: [plane][32 rows]
    static constexpr int LDS = A_BYTES + 3 * P_STAGE;
    static constexpr int JB = (TN - 1) / 2;             // the step's barrier sits behind the MFMAs of tile JB
};

// C[M,N] (+)= A[M,K] * P^T (+ bias[N]); P = x3_split_kernel planes of the [N][K] operand.  M % (128 TM) == 0, N % (32 TN) == 0,
// K % 32 == 0.  grid.x = tiles.
template <int TM, int TN, int AS>
__global__ __launch_bounds__(256, 2) void gemm_x3s_kernel(int M, int N, int K, const float *__restrict__ A, int lda,
                                                          const __bf16 *__restrict__ P, float *__restrict__ C, int ldc,
                                                          const float *__restrict__ bias, int accumulate,
                                                          double *__restrict__ colstats)
{
    typedef X3S<TM, TN, AS> G;
    extern __shared__ __attribute__((aligned(16))) unsigned char x3_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int fr = lane & 31, fk = lane >> 5;
    const int tiles_n = N / G::BN;
    const int vid = xcd_contiguous(blockIdx.x, gridDim.x);       // the tiles that share rows of A: one XCD's L2
    const int m0 = (vid / tiles_n) * G::BM, n0 = (vid % tiles_n) * G::BN;
    const int nslab = K >> 5, nstep = K >> 4;

    // ---- DMA sources.  A: wave instruction t brings rows 8t .. 8t+7 of this wave's 32 TM rows; lane -> (row 8t + lane/8,
    // stored unit lane%8), which holds the row's unit (lane%8) ^ (row/2 % 8) [= (lane%8) ^ (lane/16) ^ 4 (t odd)].
    const float *asrc = A + (size_t)(m0 + wave * 32 * TM + (lane >> 3)) * lda + 4 * ((lane & 7) ^ (lane >> 4));
    const size_t a8 = (size_t)8 * lda;
    unsigned char *lds_a = x3_lds + wave * G::A_WAVE;
    auto dma_a = [&](int slab, int stage) {
        const float *ga = asrc + (size_t)slab * 32;
        unsigned char *la = lds_a + stage * G::A_STAGE;
#pragma unroll
        for (int t = 0; t < 4 * TM; ++t)
            __builtin_amdgcn_global_load_lds(ga + t * a8 + ((t & 1) ? ((lane & 4) ? -16 : 16) : 0), la + t * 1024, 16, 0, 0);
    };
    // P: piece pc (1 KB, linear) = plane pc / TN, rows 32 (pc % TN) ..; this wave takes pc = wave, wave + 4, ...
    const __bf16 *psrc = P + (size_t)n0 * 16 + 8 * lane;
    const size_t pplane = (size_t)N * 16, pstep = 3 * pplane;
    auto dma_p = [&](int step, int ring) {
        unsigned char *sp = x3_lds + G::A_BYTES + ring * G::P_STAGE;
        const __bf16 *gp = psrc + (size_t)step * pstep;
#pragma unroll
        for (int i = 0; i < (G::P_PIECES + 3) / 4; ++i) {
            const int pc = wave + 4 * i;
            if (G::P_PIECES % 4 == 0 || pc < G::P_PIECES) {
                const int plane = pc / TN, rg = pc % TN;
                __builtin_amdgcn_global_load_lds(gp + plane * pplane + (size_t)rg * 512, sp + pc * 1024, 16, 0, 0);
            }
        }
    };

    // ---- fragment addresses (bytes)
    const int swa = (fr >> 1) & 7;
    int offa[2][2];
#pragma unroll
    for (int ss = 0; ss < 2; ++ss)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            offa[ss][hh] = wave * G::A_WAVE + fr * 128 + (((4 * ss + 2 * fk + hh) ^ swa) << 4);
    const int offp = G::A_BYTES + fr * 32 + ((fk ^ ((fr >> 3) & 1)) << 4);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.0f;

    bf16x8 af[2][3][TM], bp[2][3];
    // the wave's operands of half ss of the slab in the A buffer -> register set buf
    auto prepare_a = [&](int buf, int ss, int stage) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float4v x = *reinterpret_cast<const float4v *>(x3_lds + offa[ss][0] + i * 4096 + stage * G::A_STAGE);
            const float4v y = *reinterpret_cast<const float4v *>(x3_lds + offa[ss][1] + i * 4096 + stage * G::A_STAGE);
            split8(x, y, af[buf][0][i], af[buf][1][i], af[buf][2][i]);
        }
    };
    auto read_p = [&](int pb, int ring, int j) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            bp[pb][p] = *reinterpret_cast<const bf16x8 *>(x3_lds + offp + ring * G::P_STAGE + p * G::P_PLANE + j * 1024);
    };
    // the six piece products of column tile j, smallest first
    auto tile = [&](int abuf, int pb, int j) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[abuf][PA[q]][i], bp[pb][PB[q]], acc[i][j], 0, 0, 0);
    };
    // One step of 16 k.  On entry: af[abuf] = the step's A operands, bp[pb0] = its first column tile, its planes in `ring`;
    // on exit the same for the next step (af[abuf ^ 1], bp[pb0 ^ (TN & 1)], ring + 1).  Every DMA is issued UNCONDITIONALLY
    // (past the end it asks for the last step / slab again, which nobody reads): a branch would end the scheduling region
    // and the preparation would not be spread between the MFMAs.
    auto step = [&](int g, int ring, int abuf, int pb0, bool second_half) {
        const int ring1 = ring == 2 ? 0 : ring + 1, ring2 = ring == 0 ? 2 : ring - 1;
        const int slab = g >> 1, ast = AS == 2 ? (slab & 1) : 0;
        if (!second_half)
            prepare_a(abuf ^ 1, 1, ast);                   // (first half of the slab: the second half is there already)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int pb = (pb0 + j) & 1;
            if (j + 1 < TN)
                read_p(pb ^ 1, ring, j + 1);
            else
                read_p(pb ^ 1, ring1, 0);
            tile(abuf, pb, j);
            if (j == G::JB) {
                // planes of step g + 1 (and, in a slab's second half, the next slab of A): this wave's pieces have landed
                // (vmcnt), everyone's (barrier); every wave is past step g - 1, whose ring slot the DMA below refills.
                // AS == 2: the slab of A requested one step ago (behind the planes awaited here) may stay in flight.
                if (AS == 2 && second_half)
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * TM) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                dma_p(min(g + 2, nstep - 1), ring2);
                if (!second_half)          // this wave has read both halves of its rows: refill that buffer
                    dma_a(min(slab + AS, nslab - 1), ast);
                else
                    prepare_a(abuf ^ 1, 0, AS == 2 ? (ast ^ 1) : 0);
            }
        }
    };

    dma_a(0, 0);
    if (AS == 2)
        dma_a(min(1, nslab - 1), 1);
    dma_p(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    dma_p(min(1, nstep - 1), 1);
    prepare_a(0, 0, 0);
    read_p(0, 0, 0);
    int ring = 0;
    for (int s = 0; s < nslab; ++s) {
        step(2 * s, ring, 0, 0, false);
        ring = ring == 2 ? 0 : ring + 1;
        step(2 * s + 1, ring, 1, TN & 1, true);
        ring = ring == 2 ? 0 : ring + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the repeated last pieces: nothing of them is read, but they
    __syncthreads();                                         //  must have landed before the statistics below reuse the space)

    // epilogue: lane holds column (lane & 31), rows (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int wrow = m0 + wave * 32 * TM;
    if (colstats != nullptr) {
        // column sums / sums of squares of this tile in fp64 (as gemm_f32_kernel); the staging array lies over the slabs
        double (*cs)[4][G::BN] = reinterpret_cast<double (*)[4][G::BN]>(x3_lds);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cl = j * 32 + fr;
            const float bv = bias != nullptr ? bias[n0 + cl] : 0.0f;
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const double v = (double)(acc[i][j][r] + bv);
                    s1 += v;
                    s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (fk == 0) {
                cs[0][wave][cl] = s1;
                cs[1][wave][cl] = s2;
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < 2 * G::BN; t += 256) {
            const int which = t / G::BN, cl = t % G::BN;
            const double v = ((cs[which][0][cl] + cs[which][1][cl]) + cs[which][2][cl]) + cs[which][3][cl];
            colstats[((size_t)(m0 / G::BM) * 2 + which) * N + n0 + cl] = v;
        }
    }
    float *c0 = C + (size_t)(wrow + 4 * fk) * ldc + (n0 + fr);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float bv = bias != nullptr ? bias[n0 + j * 32 + fr] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *dst = c0 + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc + j * 32;
                const float v = acc[i][j][r] + bv;
                *dst = accumulate ? *dst + v : v;
            }
        }
    }
}

template <int TM, int TN, int AS>
static int launch_x3s(const char *name, hipStream_t s, int M, int N, int K, const float *A, int lda, const void *planes, float *C,
                      int ldc, const float *bias, int accumulate, double *cs)
{
    typedef X3S<TM, TN, AS> G;
    static bool raised[64] = {};
    int dev = 0;
    CLOUDAAE_CHECK_HIP(hipGetDevice(&dev), name);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        CLOUDAAE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_x3s_kernel<TM, TN, AS>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS), name);
        raised[dev] = true;
    }
    const int tiles = (M / G::BM) * (N / G::BN);
    hipLaunchKernelGGL((gemm_x3s_kernel<TM, TN, AS>), dim3(tiles), dim3(256), G::LDS, s, M, N, K, A, lda,
                       reinterpret_cast<const __bf16 *>(planes), C, ldc, bias, accumulate, cs);
    return 0;
}

// tile shape of a streamed product; false when it is not served
static bool gemm_x3s_plan(int M, int N, int K, int &TM, int &TN)
{
    if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0 || M % 128 != 0)
        return false;
    TN = N % 128 == 0 ? 4 : (N % 160 == 0 ? 5 : 0);
    if (TN == 0)
        return false;
    // 256-row tiles (a wave's 64 x 128 tile reads each small-operand fragment for two row tiles) when they still give
    // every CU its two workgroups; 160-column tiles keep 128 rows (five accumulator tiles per row tile: 256 registers)
    TM = (TN == 4 && M % 256 == 0 && (long long)(M / 256) * (N / 128) >= 512) ? 2 : 1;
    return true;
}

} // namespace cloudaae

using namespace cloudaae;

CLOUDAAE_API long long cloudaae_x3_planes_bytes(int rows, int k)
{
    return rows > 0 && k > 0 && k % 32 == 0 ? (long long)rows * k * 6 : 0;
}

CLOUDAAE_API int cloudaae_x3_split(int rows, int k, const float *src, int ld, int transposed, void *planes, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_x3_split";
    CLOUDAAE_REQUIRE(rows > 0 && k > 0 && k % 32 == 0 && src && planes, name, "rows, k > 0, k a multiple of 32, non-null pointers");
    CLOUDAAE_REQUIRE(ld >= (transposed ? rows : k), name, "leading dimension too small");
    CLOUDAAE_REQUIRE(transposed || (ld % 4 == 0 && ((uintptr_t)src & 15) == 0), name, "rows must be 16-byte aligned");
    CLOUDAAE_REQUIRE(((uintptr_t)planes & 15) == 0, name, "planes must be 16-byte aligned");
    const long long items = (long long)rows * (k / 8);
    hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)((items + 255) / 256), 1), dim3(256), 0, (hipStream_t)stream, rows, k, transposed,
                       reinterpret_cast<__bf16 *>(planes), 0, 0, 0, (__bf16 *)nullptr, src, ld);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_x3_split_weight(int K, int N, const float *W, int ldw, void *planes_fwd, void *planes_bwd,
                                          cloudaae_stream_t stream)
{
    const char *name = "cloudaae_x3_split_weight";
    CLOUDAAE_REQUIRE(K > 0 && N > 0 && K % 32 == 0 && N % 32 == 0 && W && planes_fwd && planes_bwd, name,
                     "K, N multiples of 32, non-null pointers");
    CLOUDAAE_REQUIRE(ldw >= N && ldw % 4 == 0 && ((uintptr_t)W & 15) == 0, name, "rows of W must be 16-byte aligned");
    CLOUDAAE_REQUIRE(((uintptr_t)planes_fwd & 15) == 0 && ((uintptr_t)planes_bwd & 15) == 0, name, "planes must be 16-byte aligned");
    const long long items = (long long)K * N / 8;
    // job 0: the planes of W^T ([N][K], for y = x W); job 1: the planes of W ([K][N], for dx = dy W^T)
    hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)((items + 255) / 256), 2), dim3(256), 0, (hipStream_t)stream, N, K, 1,
                       reinterpret_cast<__bf16 *>(planes_fwd), K, N, 0, reinterpret_cast<__bf16 *>(planes_bwd), W, ldw);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_gemm_bf16x3p_supported(int M, int N, int K)
{
    int TM, TN;
    return gemm_x3s_plan(M, N, K, TM, TN) ? 1 : 0;
}

CLOUDAAE_API int cloudaae_gemm_bf16x3p_colstats_parts(int M, int N, int K)
{
    int TM, TN;
    return gemm_x3s_plan(M, N, K, TM, TN) ? M / (128 * TM) : 0;
}

CLOUDAAE_API int cloudaae_gemm_bf16x3p(int M, int N, int K, const float *A, int lda, const void *planes, float *C, int ldc,
                                       const float *bias, int accumulate, double *colstats, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_gemm_bf16x3p";
    int TM, TN;
    CLOUDAAE_REQUIRE(gemm_x3s_plan(M, N, K, TM, TN), name,
                     "product not served (M % 128 == 0, N a multiple of 128 or 160, K % 32 == 0; see cloudaae_gemm_bf16x3p_supported)");
    CLOUDAAE_REQUIRE(A && planes && C, name, "null argument");
    CLOUDAAE_REQUIRE(lda >= K && ldc >= N, name, "leading dimension too small");
    CLOUDAAE_REQUIRE(lda % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)planes & 15) == 0, name,
                     "operand rows must be 16-byte aligned");
    CLOUDAAE_REQUIRE(colstats == nullptr || accumulate == 0, name, "column statistics need an overwriting product");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // (128-row tiles keep two slabs of the big operand in flight: AS = 2)
    if (TM == 2)
        rc = launch_x3s<2, 4, 1>(name, s, M, N, K, A, lda, planes, C, ldc, bias, accumulate, colstats);
    else if (TN == 4)
        rc = launch_x3s<1, 4, 2>(name, s, M, N, K, A, lda, planes, C, ldc, bias, accumulate, colstats);
    else
        rc = launch_x3s<1, 5, 2>(name, s, M, N, K, A, lda, planes, C, ldc, bias, accumulate, colstats);
    if (rc != 0)
        return rc;
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_gemm_bf16x3_supported(int trans_a, int trans_b, int M, int N, int K)
{
    int BM, BN, splits;
    return gemm_x3_plan(trans_a, trans_b, M, N, K, BM, BN, splits) ? 1 : 0;
}

CLOUDAAE_API int cloudaae_gemm_bf16x3_colstats_parts(int M, int N, int K)
{
    int BM, BN, splits;
    if (!gemm_x3_plan(0, 0, M, N, K, BM, BN, splits) || splits != 1)
        return 0;
    int TMs, TNs;
    if (gemm_x3s_plan(M, N, K, TMs, TNs) &&
        (long long)(M / (128 * TMs)) * (N / (32 * TNs)) >= 192)
        return M / (128 * TMs);
    return M / BM;
}

CLOUDAAE_API int cloudaae_gemm_bf16x3(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda, const float *B,
                                      int ldb, float *C, int ldc, const float *bias, int accumulate, double *colstats,
                                      cloudaae_stream_t stream)
{
    const char *name = "cloudaae_gemm_bf16x3";
    int BM, BN, splits;
    CLOUDAAE_REQUIRE(gemm_x3_plan(trans_a, trans_b, M, N, K, BM, BN, splits), name,
                     "product not served (whole tiles of 128 / 160 and K a multiple of 32; see cloudaae_gemm_bf16x3_supported)");
    CLOUDAAE_REQUIRE(A && B && C, name, "null argument");
    CLOUDAAE_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, name, "leading dimension too small");
    CLOUDAAE_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, name,
                     "operand rows must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int TMs, TNs;
    // (a product with few output tiles is cut over K by the first-generation kernel below; the streamed one keeps K whole)
    if (!trans_a && gemm_x3s_plan(M, N, K, TMs, TNs) && (long long)(M / (128 * TMs)) * (N / (32 * TNs)) >= 192) {
        // the streamed kernel wants the second operand as planes: split it into scratch of this call (stream ordered).
        // A caller that multiplies by the same matrix more than once splits it itself (cloudaae_x3_split) and calls
        // cloudaae_gemm_bf16x3p.
        void *planes = nullptr;
        CLOUDAAE_CHECK_HIP(scratch_alloc(&planes, (size_t)cloudaae_x3_planes_bytes(N, K), s), name);
        int rc = cloudaae_x3_split(N, K, B, ldb, trans_b ? 0 : 1, planes, stream);
        if (rc == 0)
            rc = cloudaae_gemm_bf16x3p(M, N, K, A, lda, planes, C, ldc, bias, accumulate, colstats, stream);
        CLOUDAAE_CHECK_HIP(hipFreeAsync(planes, s), name);
        return rc;
    }
    int kchunk = ceil_div(ceil_div(K, splits), X3_BK) * X3_BK;
    splits = ceil_div(K, kchunk);
    CLOUDAAE_REQUIRE(colstats == nullptr || (splits == 1 && accumulate == 0), name,
                     "column statistics need an unsplit, overwriting product");
    int epi = accumulate == 1 ? X3_ACCUM : X3_STORE;
    if (splits > 1) {
        epi = X3_ATOMIC;
        if (!accumulate)
            CLOUDAAE_CHECK_HIP(hipMemset2DAsync(C, sizeof(float) * (size_t)ldc, 0, sizeof(float) * (size_t)N, (size_t)M, s),
                               name);
    }
    CLOUDAAE_REQUIRE(M / BM <= 65535, name, "M too large");
    dim3 grid(N / BN, M / BM, splits);
    int rc;
    // (one slab in flight: two measured equal -- forward 146 vs 150 us at B = 32 --, the kernel is not waiting for memory: per slab
    //  a wave spends ~640 cycles splitting, ~1540 in its 48 MFMAs, and the two waves of a SIMD contend for both pipes)
    if (!trans_a && !trans_b)
        rc = launch_x3<128, 128, 2, 2, false, false, 1>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else if (!trans_a && BN == 160)
        rc = launch_x3<128, 160, 4, 1, false, true, 1>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else if (!trans_a)
        rc = launch_x3<128, 128, 2, 2, false, true, 1>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else if (BM == 160)
        rc = launch_x3<160, 128, 1, 4, true, false, 1>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else
        rc = launch_x3<128, 128, 2, 2, true, false, 1>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    if (rc != 0)
        return rc;
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

