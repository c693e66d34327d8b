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
__device__ __forceinline__ __bf16 top_half(float v)       // the upper 16 bits of v, as a bfloat16 (truncation)
{
    const unsigned short u = (unsigned short)(__float_as_uint(v) >> 16);
    __bf16 b;
    __builtin_memcpy(&b, &u, 2);
    return b;
}
__device__ __forceinline__ void split3(float v, __bf16 &h, __bf16 &m, __bf16 &l)
{
#ifdef X3_TRUNCATE
    // (experiment) pieces by truncation: v & 0xffff0000, exact remainders; four operations instead of seven per element
    const float hf = __uint_as_float(__float_as_uint(v) & 0xffff0000u);
    const float r1 = v - hf;
    const float mf = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float lf = r1 - mf;
    h = top_half(hf);
    m = top_half(mf);
    l = top_half(lf);
#else
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
#endif
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

} // namespace cloudaae

using namespace cloudaae;

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
    // CLOUDAAE_X3_DEPTH2=1: two slabs in flight for the 128 x 128 tiles (every K slice an even number of slabs).  Measured
    // equal to one slab in flight (forward 146 vs 150 us at B = 32): the kernel is not waiting for memory -- per slab a
    // wave spends ~640 cycles splitting, ~1540 in its 48 MFMAs, and the two waves of a SIMD contend for both pipes.
    const bool deep = K % kchunk == 0 && (kchunk / X3_BK) % 2 == 0 && CLOUDAAE_KNOB("CLOUDAAE_X3_DEPTH2", 0) != 0;
    if (!trans_a && !trans_b && !deep)
        rc = launch_x3<128, 128, 2, 2, false, false, 1>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else if (!trans_a && !trans_b)
        rc = launch_x3<128, 128, 2, 2, false, false, 2>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else if (!trans_a && BN == 160)
        rc = launch_x3<128, 160, 4, 1, false, true, 1>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else if (!trans_a && !deep)
        rc = launch_x3<128, 128, 2, 2, false, true, 1>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else if (!trans_a)
        rc = launch_x3<128, 128, 2, 2, false, true, 2>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else if (BM == 160)
        rc = launch_x3<160, 128, 1, 4, true, false, 1>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else if (!deep)
        rc = launch_x3<128, 128, 2, 2, true, false, 1>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else
        rc = launch_x3<128, 128, 2, 2, true, false, 2>(name, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    if (rc != 0)
        return rc;
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}
