// gemm_b16.hip -- dense products whose operands already ARE bfloat16 in HBM (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// BASELINE configs[2] ("bf16 MLPs + fp32 Chamfer").  gemm_bf16.hip rounds fp32 tensors to bf16 on their way into LDS;
// measured on the dgcnn_agg block at 256 clouds its three products (reference utils/tf_util.py:161-166 and the two
// gradient products of that convolution) are bound by what the CUs pull out of L2 -- a kernel that ONLY issues their
// loads takes 0.39 / 0.31 / 0.56 ms against 0.64 / 0.42 / 0.56 ms for the whole products -- so the lever is bytes per
// operand element.  Here the widest activations of the network (the 320-channel concatenation, the 1024-channel
// output y of dgcnn_agg and its gradient) live in HBM as bf16: every operand element costs two bytes in L2 -> CU
// traffic, needs no conversion, and y / dy cost half the HBM traffic in the batch-norm passes around the products.
// The values are the ones gemm_bf16.hip would have fed to the matrix cores (round to nearest even), so the only new
// rounding point is the STORED y (the batch-norm statistics are still taken from the fp32 accumulators).
//
// Same decomposition as gemm_bf16.hip: 4 waves own a BM x BN tile as 32 x 32 accumulators, K in slabs -- of 64 here
// (one 128-byte line per k-contiguous row; the same bytes in flight per workgroup as 32 fp32), rows padded to 72
// bf16 (conflict-free ds_read_b128), next slab prefetched through registers.  A k-contiguous operand goes to LDS as
// it is (16-byte stores); a [k][outer] operand is transposed on the way in: a lane holds rows k, k + 1 of 8 outer
// indices and writes eight 32-bit words {k, k + 1} -- a half wave covers 4 k-pairs x 8 groups and each lane writes
// its eight rows in an order rotated by its group number, which spreads the words over all 32 banks (rows are
// 8 * 36 dwords = 0 mod 32 banks apart).  Only whole tiles / whole slabs / 16-byte aligned rows (cloudaae_gemm_b16_supported).
#include "common.h"
#include "gemm.h"
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int HB_BK = 64;            // k per slab
constexpr int HB_LDK = 72;           // bf16 per staged row (64 + 8 pad)
constexpr int HB_THREADS = 256;

enum { HB_STORE = 0, HB_ACCUM = 1, HB_ATOMIC = 2 };

__device__ __forceinline__ uint16_t to_bf16_bits(float v)
{
    const __bf16 b = (__bf16)v;
    uint16_t w;
    __builtin_memcpy(&w, &b, 2);
    return w;
}

// One operand slab: ROWS outer indices x 64 k.  KC: memory is [outer][k]; else [k][outer].
template <int ROWS, bool KC>
struct Slab16 {
    static constexpr int OQ = ROWS / 8;                                   // groups of 8 outer indices (!KC)
    static constexpr int ITEMS = KC ? ROWS * (HB_BK / 8) : 256 * ((OQ + 7) / 8);
    static constexpr int PER = (ITEMS + HB_THREADS - 1) / HB_THREADS;
    static_assert(ROWS % 32 == 0, "tile sides are multiples of 32");
    u32x4 r0[PER], r1[KC ? 1 : PER];
    unsigned boff[PER];

    // !KC item = (k-pair kp of 32, group oq): lane bits [1:0] kp low, [4:2] oq low, [7:5] kp high, [8..] oq high
    static __device__ __forceinline__ int item_kp(int it) { return ((it >> 5) & 7) * 4 + (it & 3); }
    static __device__ __forceinline__ int item_oq(int it) { return (it >> 8) * 8 + ((it >> 2) & 7); }
    static __device__ __forceinline__ bool live(int it)
    {
        if (KC)
            return ITEMS % HB_THREADS == 0 || it < ITEMS;
        return OQ % 8 == 0 || item_oq(it) < OQ;
    }

    __device__ __forceinline__ void init(int ld)
    {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = u * HB_THREADS + (int)threadIdx.x;
            if (KC)
                boff[u] = 2u * (unsigned)((it / (HB_BK / 8)) * ld + 8 * (it % (HB_BK / 8)));
            else
                boff[u] = 2u * (unsigned)(2 * item_kp(it) * ld + 8 * item_oq(it));
        }
    }
    __device__ __forceinline__ void load(const uint16_t *__restrict__ P0, int ld)
    {
        const char *base = reinterpret_cast<const char *>(P0);
        const char *base1 = reinterpret_cast<const char *>(P0 + ld);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = u * HB_THREADS + (int)threadIdx.x;
            if (live(it)) {
                r0[u] = *reinterpret_cast<const u32x4 *>(base + boff[u]);
                if (!KC)
                    r1[u] = *reinterpret_cast<const u32x4 *>(base1 + boff[u]);
            }
        }
    }
    __device__ __forceinline__ void stage(uint16_t *__restrict__ lds) const
    {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int it = u * HB_THREADS + (int)threadIdx.x;
            if (!live(it))
                continue;
            if (KC) {
                const int o = it / (HB_BK / 8), c = it % (HB_BK / 8);
                *reinterpret_cast<u32x4 *>(lds + o * HB_LDK + 8 * c) = r0[u];
            } else {
                const int kp = item_kp(it), oq = item_oq(it);
                // words {k, k + 1} of outer index 0 .. 7 (scalars, not an array: a select between two array elements
                // is folded into ONE dynamically indexed read, which costs a 7-compare chain per word)
                const unsigned w0 = __builtin_amdgcn_perm(r1[u][0], r0[u][0], 0x05040100u),
                               w1 = __builtin_amdgcn_perm(r1[u][0], r0[u][0], 0x07060302u),
                               w2 = __builtin_amdgcn_perm(r1[u][1], r0[u][1], 0x05040100u),
                               w3 = __builtin_amdgcn_perm(r1[u][1], r0[u][1], 0x07060302u),
                               w4 = __builtin_amdgcn_perm(r1[u][2], r0[u][2], 0x05040100u),
                               w5 = __builtin_amdgcn_perm(r1[u][2], r0[u][2], 0x07060302u),
                               w6 = __builtin_amdgcn_perm(r1[u][3], r0[u][3], 0x05040100u),
                               w7 = __builtin_amdgcn_perm(r1[u][3], r0[u][3], 0x07060302u);
                const int rot = oq & 7;     // lane writes row (i + rot) & 7 at step i: q_i = w_((i + rot) & 7)
                const bool r1_ = (rot & 1) != 0, r2_ = (rot & 2) != 0, r4_ = (rot & 4) != 0;
                // three conditional rotations by 1, 2, 4: 24 selects
                const unsigned t0 = r1_ ? w1 : w0, t1 = r1_ ? w2 : w1, t2 = r1_ ? w3 : w2, t3 = r1_ ? w4 : w3,
                               t4 = r1_ ? w5 : w4, t5 = r1_ ? w6 : w5, t6 = r1_ ? w7 : w6, t7 = r1_ ? w0 : w7;
                const unsigned v0 = r2_ ? t2 : t0, v1 = r2_ ? t3 : t1, v2 = r2_ ? t4 : t2, v3 = r2_ ? t5 : t3,
                               v4 = r2_ ? t6 : t4, v5 = r2_ ? t7 : t5, v6 = r2_ ? t0 : t6, v7 = r2_ ? t1 : t7;
                const unsigned q[8] = {r4_ ? v4 : v0, r4_ ? v5 : v1, r4_ ? v6 : v2, r4_ ? v7 : v3,
                                       r4_ ? v0 : v4, r4_ ? v1 : v5, r4_ ? v2 : v6, r4_ ? v3 : v7};
                uint16_t *dst = lds + (8 * oq) * HB_LDK + 2 * kp;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *reinterpret_cast<unsigned *>(dst + ((i + rot) & 7) * HB_LDK) = q[i];
            }
        }
    }
};

// C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias[N]); A, B bfloat16; C fp32 or (OUT16) bfloat16
template <int BM, int BN, int WM, int WN, bool TA, bool TB, bool OUT16>
__global__ __launch_bounds__(HB_THREADS) void gemm_b16_kernel(int M, int N, int K, const uint16_t *__restrict__ A, int lda,
                                                              const uint16_t *__restrict__ B, int ldb, void *__restrict__ Cv,
                                                              int ldc, const float *__restrict__ bias, int epilogue,
                                                              int kchunk, double *__restrict__ colstats)
{
    static_assert(WM * WN * 64 == HB_THREADS, "4 waves");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    typedef Slab16<BM, !TA> SA;
    typedef Slab16<BN, TB> SB;
    constexpr int LDC16 = BN + 8;        // bf16 per row of the output tile staged for 16-byte stores
    constexpr int LDS_HALVES = (BM + BN) * HB_LDK > (OUT16 ? BM * LDC16 : 0) ? (BM + BN) * HB_LDK : BM * LDC16;
    __shared__ __attribute__((aligned(16))) uint16_t lds[LDS_HALVES];
    uint16_t *ldsA = lds, *ldsB = lds + BM * HB_LDK;

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

    const uint16_t *pa = A + (TA ? (size_t)kbeg * lda + m0 : (size_t)m0 * lda + kbeg);
    const uint16_t *pb = B + (TB ? (size_t)n0 * ldb + kbeg : (size_t)kbeg * ldb + n0);
    const size_t stepa = TA ? (size_t)HB_BK * lda : (size_t)HB_BK;
    const size_t stepb = TB ? (size_t)HB_BK : (size_t)HB_BK * ldb;
    SA sa;
    SB sb;
    sa.init(lda);
    sb.init(ldb);
    sa.load(pa, lda);
    sb.load(pb, ldb);
    const int fr = lane & 31, fk = lane >> 5;
    for (int k0 = kbeg; k0 < kend; k0 += HB_BK) {
        __syncthreads();
        sa.stage(ldsA);
        sb.stage(ldsB);
        __syncthreads();
        if (k0 + HB_BK < kend) {
            pa += stepa;
            pb += stepb;
            sa.load(pa, lda);
            sb.load(pb, ldb);
        }
#pragma unroll
        for (int s = 0; s < HB_BK / 16; ++s) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const bf16x8 *>(ldsA + ((wm * TM + i) * 32 + fr) * HB_LDK + 16 * s + 8 * fk);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const bf16x8 *>(ldsB + ((wn * TN + j) * 32 + fr) * HB_LDK + 16 * s + 8 * fk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: lane holds column (lane & 31), rows (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool add_bias = bias != nullptr && (epilogue != HB_ATOMIC || slice == 0);
    if (colstats != nullptr) {
        // column sums / sums of squares of this tile in fp64, from the fp32 values (as gemm_bf16_kernel)
        __shared__ double cs[2][WM][BN];
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
        for (int t = threadIdx.x; t < 2 * BN; t += HB_THREADS) {
            const int which = t / BN, cl = t % BN;
            double v = cs[which][0][cl];
#pragma unroll
            for (int w = 1; w < WM; ++w)
                v += cs[which][w][cl];
            colstats[((size_t)(m0 / BM) * 2 + which) * N + n0 + cl] = v;
        }
    }
    if (OUT16) {
        // the tile goes through LDS (the operand slabs are dead) so that every global store is 16 bytes of one row
        uint16_t *C = reinterpret_cast<uint16_t *>(Cv);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cl = (wn * TN + j) * 32 + fr;
            const float bv = add_bias ? bias[n0 + cl] : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                    lds[rl * LDC16 + cl] = to_bf16_bits(acc[i][j][r] + bv);
                }
        }
        __syncthreads();
        constexpr int CH = BN / 8;                   // 16-byte chunks per tile row
        for (int t = threadIdx.x; t < BM * CH; t += HB_THREADS) {
            const int rl = t / CH, c = t % CH;
            *reinterpret_cast<u32x4 *>(C + (size_t)(m0 + rl) * ldc + n0 + 8 * c) =
                *reinterpret_cast<const u32x4 *>(lds + rl * LDC16 + 8 * c);
        }
    } else {
        float *C = reinterpret_cast<float *>(Cv);
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
                    if (epilogue == HB_STORE)
                        *dst = v;
                    else if (epilogue == HB_ACCUM)
                        *dst = *dst + v;
                    else
                        atomicAdd(dst, v);
                }
            }
        }
    }
}

// fp32 -> bf16 (round to nearest even), 8 elements per thread
__global__ __launch_bounds__(256) void to_bf16_kernel(long long n8, const float *__restrict__ src, uint16_t *__restrict__ dst)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const float4v a = *reinterpret_cast<const float4v *>(src + 8 * i);
        const float4v b = *reinterpret_cast<const float4v *>(src + 8 * i + 4);
        u32x4 o;
        o[0] = (unsigned)to_bf16_bits(a.x) | ((unsigned)to_bf16_bits(a.y) << 16);
        o[1] = (unsigned)to_bf16_bits(a.z) | ((unsigned)to_bf16_bits(a.w) << 16);
        o[2] = (unsigned)to_bf16_bits(b.x) | ((unsigned)to_bf16_bits(b.y) << 16);
        o[3] = (unsigned)to_bf16_bits(b.z) | ((unsigned)to_bf16_bits(b.w) << 16);
        *reinterpret_cast<u32x4 *>(dst + 8 * i) = o;
    }
}

// tile shape and K slices; false when the product is not one this file serves
static bool gemm_b16_plan(int ta, int tb, int M, int N, int K, int &BM, int &BN, int &splits)
{
    if (M <= 0 || N <= 0 || K <= 0 || K % HB_BK != 0 || (ta && tb))
        return false;
    if (!ta && !tb) {                    // y = x W
        BM = 128;
        BN = 128;
    } else if (!ta && tb) {              // dx = dy W^T
        BM = 128;
        BN = N % 160 == 0 && N % 128 != 0 ? 160 : 128;
    } else {                             // dW = x^T dy
        BM = M % 160 == 0 && M % 128 != 0 ? 160 : 128;
        BN = 128;
    }
    if (M % BM != 0 || N % BN != 0)
        return false;
    const long long tiles = (long long)(M / BM) * (N / BN);
    splits = 1;
    if (tiles < 256 && K >= 512) {       // as gemm_bf16_plan: fill the chip, whole slices per XCD
        splits = (int)((tiles <= 4 ? 256 : 1024) / tiles);
        const int max_splits = K / 256 > 0 ? K / 256 : 1;
        if (splits > max_splits)
            splits = max_splits;
        if (splits < 1)
            splits = 1;
        if (splits > 8)
            splits = splits / 8 * 8;
    }
    return true;
}

template <int BM, int BN, int WM, int WN, bool TA, bool TB>
static void launch_b16(bool out16, dim3 grid, hipStream_t s, int M, int N, int K, const uint16_t *A, int lda,
                       const uint16_t *B, int ldb, void *C, int ldc, const float *bias, int epi, int kchunk, double *cs)
{
    if (out16)
        hipLaunchKernelGGL((gemm_b16_kernel<BM, BN, WM, WN, TA, TB, true>), grid, dim3(HB_THREADS), 0, s, M, N, K, A, lda, B,
                           ldb, C, ldc, bias, epi, kchunk, cs);
    else
        hipLaunchKernelGGL((gemm_b16_kernel<BM, BN, WM, WN, TA, TB, false>), grid, dim3(HB_THREADS), 0, s, M, N, K, A, lda,
                           B, ldb, C, ldc, bias, epi, kchunk, cs);
}

} // namespace cloudaae

using namespace cloudaae;

CLOUDAAE_API int cloudaae_gemm_b16_supported(int trans_a, int trans_b, int M, int N, int K)
{
    int BM, BN, splits;
    return gemm_b16_plan(trans_a, trans_b, M, N, K, BM, BN, splits) ? 1 : 0;
}

CLOUDAAE_API int cloudaae_gemm_b16_colstats_parts(int M, int N, int K)
{
    int BM, BN, splits;
    if (!gemm_b16_plan(0, 0, M, N, K, BM, BN, splits) || splits != 1)
        return 0;
    return M / BM;
}

CLOUDAAE_API int cloudaae_gemm_b16(int trans_a, int trans_b, int M, int N, int K, const uint16_t *A, int lda,
                                   const uint16_t *B, int ldb, void *C, int ldc, int c_is_bf16, const float *bias,
                                   int accumulate, double *colstats, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_gemm_b16";
    int BM, BN, splits;
    CLOUDAAE_REQUIRE(gemm_b16_plan(trans_a, trans_b, M, N, K, BM, BN, splits), name,
                     "product not served (whole tiles of 128 / 160 and K a multiple of 64; see cloudaae_gemm_b16_supported)");
    CLOUDAAE_REQUIRE(A && B && C, name, "null argument");
    CLOUDAAE_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, name, "leading dimension too small");
    CLOUDAAE_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, name,
                     "operand rows must be 16-byte aligned");
    CLOUDAAE_REQUIRE(!c_is_bf16 || (ldc % 8 == 0 && ((uintptr_t)C & 15) == 0), name, "output rows must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int kchunk = ceil_div(ceil_div(K, splits), HB_BK) * HB_BK;
    splits = ceil_div(K, kchunk);
    CLOUDAAE_REQUIRE(splits == 1 || (!c_is_bf16 && colstats == nullptr), name,
                     "a product cut over K adds fp32 slices: no bf16 output, no column statistics");
    CLOUDAAE_REQUIRE(!c_is_bf16 || accumulate == 0, name, "a bf16 output is overwritten");
    CLOUDAAE_REQUIRE(colstats == nullptr || accumulate == 0, name, "column statistics need an overwriting product");
    int epi = accumulate == 1 ? HB_ACCUM : HB_STORE;
    if (splits > 1) {
        epi = HB_ATOMIC;
        if (!accumulate)
            CLOUDAAE_CHECK_HIP(hipMemset2DAsync(C, sizeof(float) * (size_t)ldc, 0, sizeof(float) * (size_t)N, (size_t)M, s),
                               name);
    }
    dim3 grid(N / BN, M / BM, splits);
    CLOUDAAE_REQUIRE(M / BM <= 65535, name, "M too large");
    const bool o16 = c_is_bf16 != 0;
    if (!trans_a && !trans_b)
        launch_b16<128, 128, 2, 2, false, false>(o16, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else if (!trans_a && BN == 160)
        launch_b16<128, 160, 4, 1, false, true>(o16, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else if (!trans_a)
        launch_b16<128, 128, 2, 2, false, true>(o16, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else if (BM == 160)
        launch_b16<160, 128, 1, 4, true, false>(o16, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    else
        launch_b16<128, 128, 2, 2, true, false>(o16, grid, s, M, N, K, A, lda, B, ldb, C, ldc, bias, epi, kchunk, colstats);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_to_bf16(long long n, const float *src, uint16_t *dst, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_to_bf16";
    CLOUDAAE_REQUIRE(n >= 0 && n % 8 == 0, name, "element count must be a multiple of 8");
    if (n == 0)
        return 0;
    CLOUDAAE_REQUIRE(src && dst && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, name,
                     "null or misaligned argument");
    const long long n8 = n / 8;
    const int blocks = (int)(n8 / 256 + 1 < 8192 ? n8 / 256 + 1 : 8192);
    hipLaunchKernelGGL(to_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n8, src, dst);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}
