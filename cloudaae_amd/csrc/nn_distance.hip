// nn_distance.hip -- Chamfer nearest-neighbour forward/backward for gfx950.
//
// Replaces NmDistanceKernelLauncher / NmDistanceGradKernelLauncher
// (reference tf_ops/nn_distance/tf_nndistance_g.cu:128-131,152-157) and computes
// exactly what the reference CPU op does (tf_nndistance.cpp:21-43,126-163):
//   dist = min_k ((dx*dx + dy*dy) + dz*dz), un-fused fp32, strict '<', first
//   minimum wins.
//
// Design (not the reference's 512-point smem tiles with the running minimum
// round-tripping through global memory):
//   * one launch covers both directions; a workgroup owns 256*Q queries of one
//     cloud and streams the whole other cloud through LDS in chunks, laid out
//     as quads [x0..x3 | y0..y3 | z0..z3] so one ds_read_b128 per coordinate is
//     a broadcast read of four candidates and the arithmetic is v_pk_*_f32 on
//     candidate pairs;
//   * the hot loop tracks only the minimum VALUE per 16-candidate tile (squared
//     distances are >= +0, so their bit patterns order as unsigned integers:
//     v_min3_u32, no NaN canonicalisation) plus the tile in which the running
//     minimum last improved; the arg-min is recovered afterwards by rescanning
//     that one tile -- first index whose distance equals the minimum, which is
//     the reference's first-wins rule;
//   * results stay in registers; each output is written once.
// Build flags matter: -ffp-contract=off keeps mul/add un-fused (bit-parity with
// the CPU reference).
#include "common.h"
#include <stdlib.h>
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

constexpr int NN_THREADS = 256;
constexpr int NN_CHUNK = 1024;  // candidates per LDS chunk (12 KiB)
constexpr int NN_TILE = 16;     // candidates per min-tracking tile

__device__ __forceinline__ unsigned umin3(unsigned a, unsigned b, unsigned c)
{
    unsigned t = a < b ? a : b;
    return t < c ? t : c;
}

// un-fused squared distance of one candidate, same association as the oracle
__device__ __forceinline__ float sqdist(float cx, float cy, float cz, float qx, float qy, float qz)
{
    const float dx = cx - qx, dy = cy - qy, dz = cz - qz;
    return dx * dx + dy * dy + dz * dz;
}

// rows of xyz2 that are distinct points (cloudaae_nn_distance_prefix); all of them without a count, or with a count
// outside (0, m]
__device__ __forceinline__ int prefix_rows(const long long *count2, int cloud, int m)
{
    if (count2 == nullptr)
        return m;
    const long long c = count2[cloud];
    return c > 0 && c < m ? (int)c : m;
}

// results of the copies: row j >= count2[cloud] of xyz2 is a copy of row row_src[j] < count2[cloud].  EVERY such row is
// written: a row whose row_src is not a distinct row (< 0 or >= count2[cloud] -- a caller that shuffled, truncated or
// augmented the target after the synthesis broke the contract) gets dist = NaN, idx = 0, i.e. a loss that says so and an
// index the gradient kernels may follow, instead of whatever the output buffer held.  verify (development knob
// CLOUDAAE_NN_PREFIX_VERIFY): the row must also BE its original, bit for bit, else the same NaN.
__global__ __launch_bounds__(256) void nn_distance_expand_kernel(int m, const long long *__restrict__ count2,
                                                               const int *__restrict__ row_src, const float *__restrict__ xyz2,
                                                               int verify, float *__restrict__ dist2, int *__restrict__ idx2)
{
    const int cloud = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    const int m_eff = prefix_rows(count2, cloud, m);
    if (j < m_eff || j >= m)
        return;
    const size_t base = (size_t)cloud * m;
    const int r = row_src != nullptr ? row_src[base + j] : -1;
    bool good = r >= 0 && r < m_eff;
    if (good && verify) {
        const unsigned *a = reinterpret_cast<const unsigned *>(xyz2 + (base + j) * 3);
        const unsigned *c = reinterpret_cast<const unsigned *>(xyz2 + (base + r) * 3);
        good = a[0] == c[0] && a[1] == c[1] && a[2] == c[2];
    }
    dist2[base + j] = good ? dist2[base + r] : __uint_as_float(0x7fc00000u);
    idx2[base + j] = good ? idx2[base + r] : 0;
}

template <int Q>
__global__ __launch_bounds__(NN_THREADS) void nn_distance_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    float *__restrict__ dist1, int *__restrict__ idx1, float *__restrict__ dist2,
    int *__restrict__ idx2, int tiles1, const long long *__restrict__ count2)
{
    __shared__ float4v lds[NN_CHUNK / 4 * 3];

    const int tid = threadIdx.x;
    const int cloud = blockIdx.y;
    const bool second = (int)blockIdx.x >= tiles1;
    const int tile = second ? (int)blockIdx.x - tiles1 : (int)blockIdx.x;
    const int m_eff = prefix_rows(count2, cloud, m);      // (see nn_distance_filter_kernel)
    const int nq = second ? m_eff : n;   // queries in this direction
    const int nc = second ? n : m_eff;   // candidates
    const int sq = second ? m : n, sc = second ? n : m;
    const float *from = (second ? xyz2 : xyz1) + (size_t)cloud * sq * 3;
    const float *to = (second ? xyz1 : xyz2) + (size_t)cloud * sc * 3;
    float *dist = (second ? dist2 : dist1) + (size_t)cloud * sq;
    int *idx = (second ? idx2 : idx1) + (size_t)cloud * sq;
    if (tile * (NN_THREADS * Q) >= nq)
        return;

    float qx[Q], qy[Q], qz[Q];
    unsigned best[Q];
    int btile[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int j = tile * (NN_THREADS * Q) + q * NN_THREADS + tid;
        const bool ok = j < nq;
        qx[q] = ok ? from[3 * j] : 0.0f;
        qy[q] = ok ? from[3 * j + 1] : 0.0f;
        qz[q] = ok ? from[3 * j + 2] : 0.0f;
        best[q] = 0x7f800000u;  // +inf
        btile[q] = 0;
    }

    float *lds_f = reinterpret_cast<float *>(lds);
    for (int c0 = 0; c0 < nc; c0 += NN_CHUNK) {
        const int cnt = min(NN_CHUNK, nc - c0);
        const int padded = (cnt + NN_TILE - 1) / NN_TILE * NN_TILE;
        __syncthreads();
        for (int f = tid; f < padded * 3; f += NN_THREADS) {
            const int k = f / 3, a = f - 3 * k;
            const float v = k < cnt ? to[(size_t)(c0 + k) * 3 + a] : __builtin_inff();
            lds_f[(k >> 2) * 12 + a * 4 + (k & 3)] = v;
        }
        __syncthreads();
        const int ntile = padded / NN_TILE;
        for (int t = 0; t < ntile; ++t) {
            unsigned tmin[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q)
                tmin[q] = 0x7f800000u;
#pragma unroll
            for (int g = 0; g < NN_TILE / 4; ++g) {
                const float4v X = lds[(t * (NN_TILE / 4) + g) * 3 + 0];
                const float4v Y = lds[(t * (NN_TILE / 4) + g) * 3 + 1];
                const float4v Z = lds[(t * (NN_TILE / 4) + g) * 3 + 2];
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const float2v x0 = {X.x, X.y}, x1 = {X.z, X.w};
                    const float2v y0 = {Y.x, Y.y}, y1 = {Y.z, Y.w};
                    const float2v z0 = {Z.x, Z.y}, z1 = {Z.z, Z.w};
                    const float2v dx0 = x0 - qx[q], dx1 = x1 - qx[q];
                    const float2v dy0 = y0 - qy[q], dy1 = y1 - qy[q];
                    const float2v dz0 = z0 - qz[q], dz1 = z1 - qz[q];
                    const float2v d0 = dx0 * dx0 + dy0 * dy0 + dz0 * dz0;
                    const float2v d1 = dx1 * dx1 + dy1 * dy1 + dz1 * dz1;
                    tmin[q] = umin3(tmin[q], __float_as_uint(d0.x), __float_as_uint(d0.y));
                    tmin[q] = umin3(tmin[q], __float_as_uint(d1.x), __float_as_uint(d1.y));
                }
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const bool better = tmin[q] < best[q];
                best[q] = better ? tmin[q] : best[q];
                btile[q] = better ? (c0 / NN_TILE + t) : btile[q];
            }
        }
    }

#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int j = tile * (NN_THREADS * Q) + q * NN_THREADS + tid;
        if (j >= nq)
            continue;
        if (nc == 0) {  // tf_nndistance.cpp:28-29: best = 0, besti = 0 survive an empty loop
            dist[j] = 0.0f;
            idx[j] = 0;
            continue;
        }
        const int base = btile[q] * NN_TILE;
        int arg = base;
        for (int s = NN_TILE - 1; s >= 0; --s) {
            const int k = base + s;
            if (k < nc) {
                const float d = sqdist(to[3 * (size_t)k], to[3 * (size_t)k + 1],
                                       to[3 * (size_t)k + 2], qx[q], qy[q], qz[q]);
                arg = (__float_as_uint(d) == best[q]) ? k : arg;
            }
        }
        dist[j] = __uint_as_float(best[q]);
        idx[j] = arg;
    }
}

// ---- second generation: matrix-core filter + exact verification -----------------------------------
// The hot loop above spends 8.6 lane-operations per pair on arithmetic whose only purpose is to find
// WHICH candidate is nearest.  Here the search runs on the matrix cores with a cheaper, differently
// rounded score, and the reference's arithmetic is applied only where the answer is decided:
//   * score s_ij = |b'_j|^2 - 2 a'_i . b'_j  (= d^2_ij - |a'_i|^2, the same order over j), primes =
//     coordinates relative to the candidate cloud's first point (keeps the magnitudes, hence the
//     rounding errors, small).  One 32 x 32 tile of scores is two v_mfma_f32_32x32x2_f32 (K = 4:
//     [bx by bz |b|^2] x [-2ax -2ay -2az 1]); rows = candidates, columns = queries, so a lane's 16
//     accumulator registers are 16 candidates of ITS query and the minimum is taken in the lane.
//   * per query: the three best UNIT scores (a unit = the 32 rows of two consecutive tiles that one lane
//     half holds) and the units of the first two.
//   * afterwards the 64 candidates of the two best units are evaluated with the reference's un-fused
//     arithmetic (first index of the exact minimum wins).  That is the answer if no third unit can
//     hold a candidate that is as near in the reference's arithmetic: s3 > s1 + M with
//     M = 32 * 2^-24 * (|a'| + max_j |b'_j|)^2.  Bound: a score is a 4-term fp32 sum of exact products
//     plus the rounded |b'|^2 (<= 7 ulp-units of R = (|a'|+|b'|max)^2), the translation by the centre
//     moves a squared distance by <= 4 * 2^-24 * sqrt(R d^2) <= 4 * 2^-24 * R, and the reference's
//     value is within 8 * 2^-24 * d^2 of the true one; two candidates whose scores differ by more than
//     2 * (7 + 4 + 4) * 2^-24 * R = 30 * 2^-24 * R are therefore ordered the same way by the reference.
//   * otherwise (three units within the margin) a SECOND pass over the scores settles it: every candidate whose
//     score is at most s1 + M -- nothing else can be as near in the reference's arithmetic -- is evaluated
//     exactly where it turns up (a tile whose minimum lies above the threshold costs its two MFMAs and the
//     minimum).  Random clouds need it for about one query in 10^4; clouds with DUPLICATED candidates for every
//     query: the reference's own training targets are the visible points padded with random re-draws
//     (utils/hidden_point_removal.py:38-40), so the nearest target point of a query sits in two to four units
//     with bitwise equal scores.  (Round 2 scanned all candidates of such a query exactly, the 64 lanes sharing
//     them: 4 x the time of the search at [32,16384]^2 with fourfold duplicates.)  Non-finite scores (overflow,
//     NaN: every comparison false) still take that full scan.
// Results are the reference's bit for bit; 2 MFMA + ~13 VALU instructions per 1024 pairs instead of
// ~140.
//
// Round 5: the scores on the BF16 matrix pipe (template SPLIT).  v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate and
// shares its datapath with the vector ALU: the digest of a tile (the minima, the top three) ADDS to the matrix time
// (profiles/r05_mfma_valu_overlap.txt).  v_mfma_f32_32x32x16_bf16 is its own pipe -- vector instructions of the same wave
// overlap it -- and does sixteen products per output element in half the cycles.  Every centred coordinate is split
// exactly into three bfloat16 pieces v = h + m + l (round to nearest even; 3 x 8 significand bits), the query pieces
// pre-multiplied by -2 (exact), and a score is the sum over 21 of the 32 K-slots of two such MFMAs:
//     MFMA 1:  Ch.Qh  Ch.Qm  Cm.Qh  Cm.Qm  (3 coordinates each)   |b'|^2 as its three pieces x 1
//     MFMA 2:  Ch.Ql  Cl.Qh
// i.e. every piece product of weight >= 2^-16; each product of two bfloat16 is exact in fp32.  What a score can be off the
// exact |b'|^2 - 2 a'.b' of the centred fp32 values: the dropped pieces (m.l, l.m, l.l: <= 2^-22 |a'||b'| <= 1 unit of
// 2^-24 R), the rounding of |b'|^2 itself (3 units), and the accumulation inside the two MFMAs -- 21 non-zero terms and
// the carried sum; charged here with ONE FULL ULP of R per K-slot (truncating alignment, 2 units each: 64 units), which is
// far above what the hardware shows (tests/test_00_ops_gpu.py::test_nn_distance_split_score_error measures <= 4 units).
// With the translation (4) and the reference's own rounding (4) two candidates whose scores differ by more than
// 2 * (1 + 3 + 64 + 4 + 4) = 152 units are ordered the same way by the reference: M = 160 * 2^-24 * R (five times the
// fp32 form's margin: still ~1e-6 of R against neighbour gaps of 1e-4 R, the second pass stays a rarity).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 nf_bf16x8 __attribute__((ext_vector_type(8)));

// v = h + m + l exactly (|v| below the overflow threshold of bfloat16; infinities and NaN give non-finite pieces and
// scores, which take the full scan)
__device__ __forceinline__ void nf_split3(float v, __bf16 &h, __bf16 &m, __bf16 &l)
{
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}
// K-slots of MFMA 1:  0-2 Ch.Qh | 3-5 Ch.Qm | 6-8 Cm.Qh | 9-11 Cm.Qm | 12-14 the pieces of |b'|^2 x 1 | 15 nothing
// K-slots of MFMA 2:  0-2 Ch.Ql | 3-5 Cl.Qh | 6-15 nothing
// A operands of a candidate row (centred x, y, z and |b'|^2): a1lo / a1hi = k 0-7 / 8-15 of MFMA 1, a2lo = k 0-7 of MFMA 2
__device__ __forceinline__ void nf_split_candidate(float x, float y, float z, float bb, bool padding, nf_bf16x8 &a1lo,
                                                   nf_bf16x8 &a1hi, nf_bf16x8 &a2lo)
{
    __bf16 xh, xm, xl, yh, ym, yl, zh, zm, zl, nh, nm, nl;
    nf_split3(x, xh, xm, xl);
    nf_split3(y, yh, ym, yl);
    nf_split3(z, zh, zm, zl);
    nf_split3(bb, nh, nm, nl);
    const __bf16 zero = (__bf16)0.0f;
    if (padding) {      // (|b'|^2 = +inf: inf - inf would make its lower pieces NaN)
        nm = zero;
        nl = zero;
    }
    a1lo = nf_bf16x8{xh, yh, zh, xh, yh, zh, xm, ym};
    a1hi = nf_bf16x8{zm, xm, ym, zm, nh, nm, nl, zero};
    a2lo = nf_bf16x8{xh, yh, zh, xl, yl, zl, zero, zero};
}
// B operands of a query (centred a'): this lane's k = 8 half .. 8 half + 7 of the two MFMAs
__device__ __forceinline__ void nf_split_query(float ax, float ay, float az, int half, nf_bf16x8 &b1, nf_bf16x8 &b2)
{
    __bf16 xh, xm, xl, yh, ym, yl, zh, zm, zl;
    nf_split3(-2.0f * ax, xh, xm, xl);
    nf_split3(-2.0f * ay, yh, ym, yl);
    nf_split3(-2.0f * az, zh, zm, zl);
    const __bf16 one = (__bf16)1.0f, zero = (__bf16)0.0f;
    b1 = half ? nf_bf16x8{zh, xm, ym, zm, one, one, one, zero} : nf_bf16x8{xh, yh, zh, xm, ym, zm, xh, yh};
    b2 = half ? nf_bf16x8{zero, zero, zero, zero, zero, zero, zero, zero} : nf_bf16x8{xl, yl, zl, xh, yh, zh, zero, zero};
}

constexpr int NF_QT = 2;            // query tiles (32 queries each) per wave (4: 208 VGPRs, two waves per SIMD, 116 us at
                                    // [32,4096]^2 and 186 us at [32,16384]x[32,1024]; 2: 109 VGPRs, four waves: 111 / 149 us; 1: 115 us)
constexpr int NF_WAVES = 4;
constexpr int NF_QBLOCK = NF_WAVES * NF_QT * 32;   // queries per workgroup
constexpr int NF_CHUNK = 2048;      // candidates per LDS chunk: 64 tiles x 64 lanes x 8 bytes = 32 KiB
constexpr int NF_CHUNK_SPLIT = 1024; // the split form stages 48 bytes per candidate: 32 tiles x (64 + 32) lanes x 16 bytes = 48 KiB

__device__ __forceinline__ bool key_less(unsigned d, int i, unsigned bd, int bi)
{
    return d < bd || (d == bd && i < bi);
}

// v_min3_f32 without the canonicalising v_max_f32 the compiler puts in front of fminf() on values it
// cannot prove quiet (a NaN score only has to make the final comparison fail, which it does)
__device__ __forceinline__ float nf_min3(float a, float b, float c)
{
    float r;
    __asm__("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// the 16 accumulator rows a lane holds (16 candidates of its query) folded into a running minimum
__device__ __forceinline__ float nf_min16(float tm, const f32x16 &v)
{
#pragma unroll
    for (int r = 0; r < 16; r += 2)
        tm = nf_min3(tm, v[r], v[r + 1]);
    return tm;
}

// the three best unit scores of a query seen so far (s1 <= s2 <= s3) and the units of the first two;
// a unit = the 32 rows of a PAIR of 32-candidate tiles that one lane half holds
struct NfTop {
    float s1, s2, s3;
    int u1, u2;
    __device__ __forceinline__ void init()
    {
        s1 = s2 = s3 = __builtin_inff();
        u1 = u2 = 0;
    }
    // units arrive in ascending order within a lane: strict '<' keeps the earlier one on ties
    __device__ __forceinline__ void push(float v, int u)
    {
        const bool lt1 = v < s1, lt2 = v < s2, lt3 = v < s3;
        s3 = lt2 ? s2 : (lt3 ? v : s3);
        s2 = lt1 ? s1 : (lt2 ? v : s2);
        u2 = lt1 ? u1 : (lt2 ? u : u2);
        s1 = lt1 ? v : s1;
        u1 = lt1 ? u : u1;
    }
    // the same with ties broken by the unit id (merging the two lane halves: both lanes of a query must
    // arrive at the same order)
    __device__ __forceinline__ void merge(float v, int u)
    {
        const bool lt1 = v < s1 || (v == s1 && u < u1), lt2 = v < s2 || (v == s2 && u < u2), lt3 = v < s3;
        s3 = lt2 ? s2 : (lt3 ? v : s3);
        s2 = lt1 ? s1 : (lt2 ? v : s2);
        u2 = lt1 ? u1 : (lt2 ? u : u2);
        s1 = lt1 ? v : s1;
        u1 = lt1 ? u : u1;
    }
};

template <bool SPLIT>
__global__ __launch_bounds__(NF_WAVES * 64, 2) void nn_distance_filter_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    float *__restrict__ dist1, int *__restrict__ idx1, float *__restrict__ dist2,
    int *__restrict__ idx2, int blocks1, int split1, int split2, unsigned long long *__restrict__ keys1,
    unsigned long long *__restrict__ keys2, const long long *__restrict__ count2)
{
    // split > 1: the candidates of that direction are cut into `split` ranges (multiples of NF_CHUNK), one
    // workgroup per (query block, range); every range delivers its exact first-index minimum and the ranges meet
    // in keys[] = min over (distance bits << 32 | index) -- squared distances are >= +0, so their bit patterns
    // order like the values, and equal distances resolve to the lower index, the reference's first-wins rule.
    // A cloud pair of very unequal sizes (the reference's own benchmark: 16384 x 1024 points) otherwise leaves
    // the few workgroups that own the short side's queries scanning the whole long side alone.
    constexpr int CHUNK = SPLIT ? NF_CHUNK_SPLIT : NF_CHUNK;
    // fp32 form: float2 per lane and tile; split form: the A operands of MFMA 1 (64 lanes) and of MFMA 2 (its k = 8 .. 15
    // are zeros: 32 lanes), 16 bytes each.  (+1 tile: the pipeline reads one ahead)
    __shared__ __attribute__((aligned(16))) char cand_raw[SPLIT ? (CHUNK / 32 + 1) * 96 * 16 : (CHUNK / 32 + 1) * 64 * 8];
    float2v (*cand)[64] = reinterpret_cast<float2v (*)[64]>(cand_raw);
    nf_bf16x8 (*cand2)[32] = reinterpret_cast<nf_bf16x8 (*)[32]>(cand_raw);
    nf_bf16x8 (*cand1)[64] = reinterpret_cast<nf_bf16x8 (*)[64]>(cand_raw + (CHUNK / 32 + 1) * 32 * 16);
    __shared__ float bmax_s[NF_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c32 = lane & 31, half = lane >> 5;
    const int cloud = blockIdx.y;
    const bool second = (int)blockIdx.x >= blocks1;
    const int split = second ? split2 : split1;
    const int lin = second ? (int)blockIdx.x - blocks1 : (int)blockIdx.x;
    const int blk = lin / split, part = lin % split;
    // count2 (cloudaae_nn_distance_prefix): only the first count2[cloud] rows of xyz2 are distinct points, the rest are
    // copies of them -- they are neither candidates (first index wins: the answer is among the originals) nor queries
    // (nn_distance_expand_kernel copies their results from the originals)
    const int m_eff = prefix_rows(count2, cloud, m);
    const int nq = second ? m_eff : n, nc = second ? n : m_eff;
    const int sq = second ? m : n, sc = second ? n : m;     // row counts of the arrays
    const float *from = (second ? xyz2 : xyz1) + (size_t)cloud * sq * 3;
    const float *to = (second ? xyz1 : xyz2) + (size_t)cloud * sc * 3;
    float *dist = (second ? dist2 : dist1) + (size_t)cloud * sq;
    int *idx = (second ? idx2 : idx1) + (size_t)cloud * sq;
    unsigned long long *keys = (second ? keys2 : keys1) + (size_t)cloud * sq;
    if (blk * NF_QBLOCK >= nq)
        return;                                         // (a query block of copies only)
    // this workgroup's candidate range
    const int range = ((nc + split - 1) / split + NF_CHUNK - 1) / NF_CHUNK * NF_CHUNK;     // (the host's cut: multiples of NF_CHUNK)
    const int cbeg = min(part * range, nc), cend = min(cbeg + range, nc);
    if (cbeg >= cend)
        return;                                         // (an empty trailing range)

    const float cx = to[0], cy = to[1], cz = to[2];
    // whole quads of candidates can be read as three 16-byte pieces (rows of 12 bytes: every fourth row starts one)
    const bool to_quads = ((uintptr_t)to & 15) == 0 && (sc & 3) == 0;
    float qx[NF_QT], qy[NF_QT], qz[NF_QT], b0[NF_QT], b1[NF_QT], a2[NF_QT];
    nf_bf16x8 B1[NF_QT], B2[NF_QT];      // split form: the B operands (this lane's k = 8 half .. 8 half + 7)
    NfTop top[NF_QT];
#pragma unroll
    for (int q = 0; q < NF_QT; ++q) {
        const int j = min(blk * NF_QBLOCK + (wv * NF_QT + q) * 32 + c32, nq - 1);
        qx[q] = from[3 * (size_t)j];
        qy[q] = from[3 * (size_t)j + 1];
        qz[q] = from[3 * (size_t)j + 2];
        const float ax = qx[q] - cx, ay = qy[q] - cy, az = qz[q] - cz;
        a2[q] = ax * ax + ay * ay + az * az;
        b0[q] = half ? -2.0f * ay : -2.0f * ax;
        b1[q] = half ? 1.0f : -2.0f * az;
        if (SPLIT)
            nf_split_query(ax, ay, az, half, B1[q], B2[q]);
        top[q].init();
    }

    // split form, in halves (the first pass interleaves the digest of the other accumulator pair with the second half)
    auto issue_first = [&](f32x16 (&acc)[NF_QT], int t) {
        const nf_bf16x8 A1 = cand1[t][lane];
#pragma unroll
        for (int q = 0; q < NF_QT; ++q) {
            f32x16 zero;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                zero[r] = 0.0f;
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1[q], zero, 0, 0, 0);
        }
    };
    auto issue_second = [&](f32x16 (&acc)[NF_QT], int t) {
        nf_bf16x8 A2 = cand2[t][c32];
        if (half)
            A2 = nf_bf16x8{(__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};
#pragma unroll
        for (int q = 0; q < NF_QT; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B2[q], acc[q], 0, 0, 0);
    };
    // scores of one candidate tile for the wave's NF_QT query tiles: 2 MFMAs each
    auto issue = [&](f32x16 (&acc)[NF_QT], int t) {
        if (SPLIT) {
            const nf_bf16x8 A1 = cand1[t][lane];
            nf_bf16x8 A2 = cand2[t][c32];
            if (half)
                A2 = nf_bf16x8{(__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};
#pragma unroll
            for (int q = 0; q < NF_QT; ++q) {
                f32x16 zero;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    zero[r] = 0.0f;
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1[q], zero, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < NF_QT; ++q)
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B2[q], acc[q], 0, 0, 0);
            return;
        }
        const float2v A = cand[t][lane];
#pragma unroll
        for (int q = 0; q < NF_QT; ++q) {
            f32x16 zero;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                zero[r] = 0.0f;
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, b0[q], zero, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < NF_QT; ++q)
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, b1[q], acc[q], 0, 0, 0);
    };
    // one chunk of candidates into LDS as MFMA A operands (lane r of a tile: row r; padding rows never win)
    float bmax2 = 0.0f;
    auto stage = [&](int c0, int cnt, int padded, bool track) {
        for (int k = tid; k < padded + 32; k += NF_WAVES * 64) {
            float x = 0.0f, y = 0.0f, z = 0.0f, bb = __builtin_inff();     // padding never wins
            if (k < cnt) {
                const float *p = to + (size_t)(c0 + k) * 3;
                x = p[0] - cx;
                y = p[1] - cy;
                z = p[2] - cz;
                bb = x * x + y * y + z * z;
                if (track)
                    bmax2 = fmaxf(bmax2, bb);
            }
            if (SPLIT) {
                nf_bf16x8 lo, hi, lo2;
                nf_split_candidate(x, y, z, bb, !(k < cnt), lo, hi, lo2);
                cand1[k >> 5][k & 31] = lo;
                cand1[k >> 5][32 + (k & 31)] = hi;
                cand2[k >> 5][k & 31] = lo2;
            } else {
                // MFMA A operand: lane r holds k = 0 of row r, lane 32 + r holds k = 1
                cand[k >> 5][k & 31] = float2v{x, z};
                cand[k >> 5][32 + (k & 31)] = float2v{y, bb};
            }
        }
    };
    // The digest of a tile (8 v_min3 per query tile; the insertion into the top three once per PAIR of
    // tiles) is NOT hidden behind the MFMAs of the next one: measured, the loop costs the matrix time
    // (78 us at B=32, 4096^2, 70 % of the pipe) PLUS the VALU time whether the two are interleaved by
    // hand, left to two waves per SIMD, or both -- so the VALU work is what gets minimised.
    float run[NF_QT];
    for (int c0 = cbeg; c0 < cend; c0 += CHUNK) {
        const int cnt = min(CHUNK, cend - c0);
        const int padded = (cnt + 31) & ~31;
        __syncthreads();
        stage(c0, cnt, padded, true);
        __syncthreads();
        const int ntile = padded >> 5, t0 = c0 >> 5;
        f32x16 accA[NF_QT], accB[NF_QT];
        issue(accA, 0);
        if (SPLIT)
            __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < ntile; t += 2) {
            if (SPLIT) {
                // nf_min3 is inline assembly, which the compiler's hazard recogniser does not see: it neither counts the
                // wait states an XDL result needs before a vector instruction may read it (v_mfma_f32_32x32x16_bf16: 11)
                // nor keeps its scheduler from putting the v_min3 of an accumulator right behind the MFMA that writes it
                // -- which it did: stale reads, a few wrong answers per thousand, different in every launch.  (The fp32
                // matrix instruction of the other form is not an XDL operation and is interlocked.)  The order is pinned
                // in four groups; inside a group the scheduler is free, and what a group reads was written at least one
                // whole group of NF_QT independent MFMAs (>= 64 cycles) earlier:
                //   1  first MFMAs of B          2  second MFMAs of B  |  minima of A (last written in group 4)
                //   3  first MFMAs of the next A  4  second MFMAs of A  |  minima of B (last written in group 2), push
                issue_first(accB, t + 1);       // (tile ntile is all padding: inf scores)
                __builtin_amdgcn_sched_barrier(0);
                issue_second(accB, t + 1);
#pragma unroll
                for (int q = 0; q < NF_QT; ++q)
                    run[q] = nf_min16(__builtin_inff(), accA[q]);
                __builtin_amdgcn_sched_barrier(0);
                issue_first(accA, min(t + 2, ntile));
                __builtin_amdgcn_sched_barrier(0);
                issue_second(accA, min(t + 2, ntile));
#pragma unroll
                for (int q = 0; q < NF_QT; ++q)
                    top[q].push(nf_min16(run[q], accB[q]), (t0 + t) >> 1);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            issue(accB, t + 1);             // (tile ntile is all padding: inf scores)
#pragma unroll
            for (int q = 0; q < NF_QT; ++q)
                run[q] = nf_min16(__builtin_inff(), accA[q]);
            issue(accA, min(t + 2, ntile));
#pragma unroll
            for (int q = 0; q < NF_QT; ++q)
                top[q].push(nf_min16(run[q], accB[q]), (t0 + t) >> 1);
        }
    }

    // radius of the candidate cloud around its first point (for the margin)
    {
        float v = bmax2;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            v = fmaxf(v, __shfl_xor(v, off, 64));
        if (lane == 0)
            bmax_s[wv] = v;
        __syncthreads();
        bmax2 = fmaxf(fmaxf(bmax_s[0], bmax_s[1]), fmaxf(bmax_s[2], bmax_s[3]));
    }
    const float rb = sqrtf(bmax2);

    unsigned best_b[NF_QT];
    int best_i[NF_QT];
    float theta[NF_QT];
    bool full[NF_QT];
#pragma unroll
    for (int q = 0; q < NF_QT; ++q) {
        const int j = blk * NF_QBLOCK + (wv * NF_QT + q) * 32 + c32;
        // the two lanes of a query merge what they saw: units 2 t + half are disjoint between them
        NfTop g = top[q];
        g.u1 = 2 * g.u1 + half;
        g.u2 = 2 * g.u2 + half;
        const float p1 = __shfl_xor(g.s1, 32, 64), p2 = __shfl_xor(g.s2, 32, 64), p3 = __shfl_xor(g.s3, 32, 64);
        const int pu1 = __shfl_xor(g.u1, 32, 64), pu2 = __shfl_xor(g.u2, 32, 64);
        g.merge(p1, pu1);
        g.merge(p2, pu2);
        g.merge(p3, 0x7fffffff);
        const float r = sqrtf(a2[q]) + rb;
        // (+ an absolute floor: below ~1e-37 the matrix cores may flush denormal terms, so relative bounds
        // mean nothing there and such clouds always take the full scan)
        const float margin = (SPLIT ? 160.0f : 32.0f) * 5.9604645e-8f * (r * r) + 1e-36f;
        // only the two best units can hold a candidate within the margin of the best score?
        const bool decided = g.s3 > g.s1 + margin;      // false for NaN / overflow as well

        // those two units in the reference's arithmetic: one per lane of the pair, 32 candidates each
        // (unit = 2 * tile pair + lane half: rows 4h..4h+3, 8+4h.. of tiles 2p and 2p+1)
        unsigned kb = 0xffffffffu;
        int ki = 0x7fffffff;
        const int unit = half ? g.u2 : g.u1;
        const int base = (unit >> 1) * 64 + 4 * (unit & 1);
        if (to_quads) {
            // four consecutive candidates are 48 contiguous, 16-byte aligned bytes: three dwordx4 loads per quad, four quads
            // in flight (the scalar form below is 96 dependent-latency loads per unit in batches of twelve: measured, the
            // exact evaluation was a quarter of the kernel)
#pragma unroll
            for (int g0 = 0; g0 < 8; g0 += 4) {
                float4v v[4][3];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int kq = base + 8 * (g0 + g);
                    const float4v *p = reinterpret_cast<const float4v *>(to + 3 * (size_t)(kq < cend ? kq : 0));
                    v[g][0] = p[0];
                    v[g][1] = p[1];
                    v[g][2] = p[2];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float f[12] = {v[g][0].x, v[g][0].y, v[g][0].z, v[g][0].w, v[g][1].x, v[g][1].y, v[g][1].z, v[g][1].w,
                                         v[g][2].x, v[g][2].y, v[g][2].z, v[g][2].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = base + 8 * (g0 + g) + e;
                        const unsigned u = __float_as_uint(sqdist(f[3 * e], f[3 * e + 1], f[3 * e + 2], qx[q], qy[q], qz[q]));
                        if (k < cend && u < kb) {       // ascending k: the first minimum stays
                            kb = u;
                            ki = k;
                        }
                    }
                }
            }
        } else {
#pragma unroll 4
            for (int s = 0; s < 32; ++s) {
                const int k = base + (s & 3) + 8 * (s >> 2);
                if (k < cend) {
                    const float d = sqdist(to[3 * (size_t)k], to[3 * (size_t)k + 1], to[3 * (size_t)k + 2], qx[q],
                                           qy[q], qz[q]);
                    const unsigned u = __float_as_uint(d);
                    if (u < kb) {       // ascending k: the first minimum stays
                        kb = u;
                        ki = k;
                    }
                }
            }
        }
        {
            const unsigned od = __shfl_xor(kb, 32, 64);
            const int oi = __shfl_xor(ki, 32, 64);
            if (key_less(od, oi, kb, ki)) {
                kb = od;
                ki = oi;
            }
        }
        best_b[q] = kb;
        best_i[q] = ki;
        // undecided with finite numbers: the second pass below; anything else (NaN, overflow): the full scan
        const float limit = g.s1 + margin;
        const bool finite = limit < __builtin_inff() && limit > -__builtin_inff();   // false for NaN too
        theta[q] = (!decided && finite && j < nq) ? limit : -__builtin_inff();
        full[q] = !decided && !finite && half == 0 && j < nq;
    }

    // ---- second pass: every candidate with a score <= s1 + M, exactly ----
    // It costs the whole workgroup another walk over the candidates, so it is taken when MANY of the workgroup's
    // queries are undecided (duplicated candidates: all of them); a stray undecided query (random clouds: one in
    // 10^4) is cheaper to settle by the full scan below, which only occupies its own wave.
    {
        int mine = 0;
#pragma unroll
        for (int q = 0; q < NF_QT; ++q)
            mine += (theta[q] > -__builtin_inff() && half == 0) ? 1 : 0;
        const int undecided = __syncthreads_count(mine > 0) + (NF_QT > 1 ? __syncthreads_count(mine > 1) : 0);
        if (undecided < NF_QBLOCK / 16) {
#pragma unroll
            for (int q = 0; q < NF_QT; ++q) {
                full[q] = full[q] || (theta[q] > -__builtin_inff() && half == 0);
                theta[q] = -__builtin_inff();
            }
        } else {
            for (int c0 = cbeg; c0 < cend; c0 += CHUNK) {
                const int cnt = min(CHUNK, cend - c0);
                const int padded = (cnt + 31) & ~31;
                __syncthreads();
                stage(c0, cnt, padded, false);
                __syncthreads();
                const int ntile = padded >> 5;
                for (int t = 0; t < ntile; ++t) {
                    f32x16 acc[NF_QT];
                    issue(acc, t);              // the same instructions on the same operands: the same scores
                    if (SPLIT) {                // (the wait states of the XDL results, by hand: see the first pass)
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_nop 15");
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int q = 0; q < NF_QT; ++q) {
                        const float tm = nf_min16(__builtin_inff(), acc[q]);
                        if (__any(tm <= theta[q])) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int k = c0 + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                                if (acc[q][e] <= theta[q] && k < cend) {
                                    const unsigned u = __float_as_uint(sqdist(to[3 * (size_t)k], to[3 * (size_t)k + 1],
                                                                              to[3 * (size_t)k + 2], qx[q], qy[q], qz[q]));
                                    if (key_less(u, k, best_b[q], best_i[q])) {
                                        best_b[q] = u;
                                        best_i[q] = k;
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
    }

#pragma unroll
    for (int q = 0; q < NF_QT; ++q) {
        const int j = blk * NF_QBLOCK + (wv * NF_QT + q) * 32 + c32;
        unsigned kb = best_b[q];
        int ki = best_i[q];
        {   // the two lanes of a query saw disjoint rows in the second pass
            const unsigned od = __shfl_xor(kb, 32, 64);
            const int oi = __shfl_xor(ki, 32, 64);
            if (key_less(od, oi, kb, ki)) {
                kb = od;
                ki = oi;
            }
        }
        // non-finite scores: the wave scans every candidate of that query together
        unsigned long long todo = __ballot(full[q]);
        while (todo) {
            const int L = __builtin_ctzll(todo);
            todo &= todo - 1;
            const float fx = __shfl(qx[q], L, 64), fy = __shfl(qy[q], L, 64), fz = __shfl(qz[q], L, 64);
            unsigned sb = 0xffffffffu;
            int si = 0x7fffffff;
            constexpr int RU = 4;       // loads of four candidates in flight per lane
            for (int k0 = cbeg + lane; k0 < cend; k0 += 64 * RU) {
                float px[RU], py[RU], pz[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int k = min(k0 + 64 * u, cend - 1);
                    px[u] = to[3 * (size_t)k];
                    py[u] = to[3 * (size_t)k + 1];
                    pz[u] = to[3 * (size_t)k + 2];
                }
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int k = k0 + 64 * u;
                    const unsigned v = __float_as_uint(sqdist(px[u], py[u], pz[u], fx, fy, fz));
                    if (k < cend && v < sb) {   // ascending k within the lane: the first minimum stays
                        sb = v;
                        si = k;
                    }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned od = __shfl_xor(sb, off, 64);
                const int oi = __shfl_xor(si, off, 64);
                if (key_less(od, oi, sb, si)) {
                    sb = od;
                    si = oi;
                }
            }
            if (c32 == L) {
                kb = sb;
                ki = si;
            }
        }
        if (half == 0 && j < nq) {
            if (split > 1) {
                atomicMin(&keys[j], ((unsigned long long)kb << 32) | (unsigned)ki);
            } else {
                dist[j] = __uint_as_float(kb);
                idx[j] = ki;
            }
        }
    }
}

// Development / test entry (cloudaae_dev_nn_split_scores): the split scores of up to 32 queries against up to 32 candidates,
// built by the SAME operand functions and MFMAs as nn_distance_filter_kernel<true>, next to R = (|a'| + max |b'|)^2 per
// query -- what tests/test_00_ops_gpu.py compares with float64 to show how far inside the margin's accumulation charge
// the hardware stays.
__global__ __launch_bounds__(64) void nn_split_scores_kernel(int nq, int nc, const float *__restrict__ from,
                                                             const float *__restrict__ to, float *__restrict__ scores,
                                                             float *__restrict__ R)
{
    const int lane = threadIdx.x, c32 = lane & 31, half = lane >> 5;
    const float cx = to[0], cy = to[1], cz = to[2];
    // candidate row c32 (A operand), query column c32 (B operand)
    const int kc = min(c32, nc - 1), jq = min(c32, nq - 1);
    const float x = to[3 * kc] - cx, y = to[3 * kc + 1] - cy, z = to[3 * kc + 2] - cz;
    const float bb = c32 < nc ? x * x + y * y + z * z : __builtin_inff();
    nf_bf16x8 a1lo, a1hi, a2lo, b1, b2;
    nf_split_candidate(c32 < nc ? x : 0.0f, c32 < nc ? y : 0.0f, c32 < nc ? z : 0.0f, bb, !(c32 < nc), a1lo, a1hi, a2lo);
    const float ax = from[3 * jq] - cx, ay = from[3 * jq + 1] - cy, az = from[3 * jq + 2] - cz;
    nf_split_query(ax, ay, az, half, b1, b2);
    const nf_bf16x8 zero8 = {(__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(half ? a1hi : a1lo, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(half ? zero8 : a2lo, b2, acc, 0, 0, 0);
    float bmax = c32 < nc ? bb : 0.0f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        bmax = fmaxf(bmax, __shfl_xor(bmax, off, 64));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;      // candidate
        if (row < nc && c32 < nq)
            scores[c32 * 32 + row] = acc[r];
    }
    if (half == 0 && c32 < nq) {
        const float rr = sqrtf(ax * ax + ay * ay + az * az) + sqrtf(bmax);
        R[c32] = rr * rr;
    }
}

// keys[] of a direction whose candidates were split -> the two output arrays
__global__ __launch_bounds__(256) void nn_distance_unpack_kernel(long long total, const unsigned long long *__restrict__ keys,
                                                                float *__restrict__ dist, int *__restrict__ idx)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i < total) {
        const unsigned long long k = keys[i];
        dist[i] = __uint_as_float((unsigned)(k >> 32));
        idx[i] = (int)(unsigned)k;
    }
}

// Backward: one thread per point and direction.  grad_a[j] += 2 g (a_j - b_t),
// grad_b[t] -= same (tf_nndistance.cpp:131-161).  Both sweeps run concurrently
// and accumulate with hardware fp32 atomics into zero-filled outputs, i.e. the
// summation ORDER differs from the reference's sequential CPU sweep (as does the
// reference's own GPU kernel, tf_nndistance_g.cu:143-148).
__global__ __launch_bounds__(256) void nn_distance_grad_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ grad_dist1, const int *__restrict__ idx1,
    const float *__restrict__ grad_dist2, const int *__restrict__ idx2,
    float *__restrict__ grad_xyz1, float *__restrict__ grad_xyz2, int blocks1,
    const float *__restrict__ uniform, float uniform_scale)
{
    const int cloud = blockIdx.y;
    const bool second = (int)blockIdx.x >= blocks1;
    const int blk = second ? (int)blockIdx.x - blocks1 : (int)blockIdx.x;
    const int j = blk * 256 + (int)threadIdx.x;
    const int na = second ? m : n, nb = second ? n : m;
    if (j >= na)
        return;
    const float *A = (second ? xyz2 : xyz1) + (size_t)cloud * na * 3;
    const float *B = (second ? xyz1 : xyz2) + (size_t)cloud * nb * 3;
    float *gA = second ? grad_xyz2 : grad_xyz1;
    float *gB = second ? grad_xyz1 : grad_xyz2;
    const int t = (second ? idx2 : idx1)[(size_t)cloud * na + j];
    // uniform: every distance has the same upstream gradient uniform[0] * uniform_scale (a mean over them)
    const float g = (uniform != nullptr ? uniform[0] * uniform_scale
                                        : (second ? grad_dist2 : grad_dist1)[(size_t)cloud * na + j]) * 2;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = g * (A[3 * j + a] - B[3 * (size_t)t + a]);
        if (gA)
            atomicAdd(&gA[((size_t)cloud * na + j) * 3 + a], v);
        if (gB)
            atomicAdd(&gB[((size_t)cloud * nb + t) * 3 + a], -v);
    }
}

// The same sums with the additions done in LDS: one workgroup per (cloud, chunk of 2048 points of one output array) holds
// that chunk's gradient rows in LDS, writes every point's own term (plain stores: one thread per point), walks the indices
// of the OTHER sweep and adds the terms that name its points with LDS atomics, and stores the chunk once -- no memset, no
// global atomics.  The global form above funnels every collision through one L2 address: a decoder early in training maps
// most of the target onto a few predicted points ([32, 4096 | 4096]: 27 us, [32, 4096 | 16384]: 243 us for 0.8 / 2 M additions).
// All loads of a thread's items are issued before the first is used (clamped addresses instead of branches): a workgroup
// is four memory round trips long.  Order of the additions: as unordered as before.
constexpr int NG_LDS_THREADS = 1024;
constexpr int NG_CH = 2048;                             // points per chunk: 24 KB of LDS
constexpr int NG_U = 4;                                 // items of the other sweep per thread and batch
__global__ __launch_bounds__(NG_LDS_THREADS) void nn_distance_grad_lds_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ grad_dist1, const int *__restrict__ idx1,
    const float *__restrict__ grad_dist2, const int *__restrict__ idx2,
    float *__restrict__ grad_xyz1, float *__restrict__ grad_xyz2, int chunks1,
    const float *__restrict__ uniform, float uniform_scale)
{
    __shared__ float ng_acc[3 * NG_CH];
    const int cloud = blockIdx.y;
    const bool second = (int)blockIdx.x >= chunks1;         // this workgroup's output: grad_xyz2
    float *out = second ? grad_xyz2 : grad_xyz1;
    if (out == nullptr)
        return;
    const int na = second ? m : n, nb = second ? n : m;     // na: this side's points, nb: the other side's
    const int p0 = ((int)blockIdx.x - (second ? chunks1 : 0)) * NG_CH, np = min(NG_CH, na - p0);
    const float *A = (second ? xyz2 : xyz1) + (size_t)cloud * na * 3;
    const float *B = (second ? xyz1 : xyz2) + (size_t)cloud * nb * 3;
    const int *own_idx = (second ? idx2 : idx1) + (size_t)cloud * na;
    const int *oth_idx = (second ? idx1 : idx2) + (size_t)cloud * nb;
    const float *own_g = (second ? grad_dist2 : grad_dist1), *oth_g = (second ? grad_dist1 : grad_dist2);
    const float gu = uniform != nullptr ? uniform[0] * uniform_scale * 2 : 0.0f;
    const int tid = threadIdx.x, lane = tid & 63;
    {   // own terms of the chunk's points
        constexpr int U = NG_CH / NG_LDS_THREADS;
        int jj[U], tt[U];
        float gg[U], av[U][3], bv[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            jj[u] = p0 + min(u * NG_LDS_THREADS + tid, np - 1);
            tt[u] = own_idx[jj[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            gg[u] = uniform != nullptr ? gu : own_g[(size_t)cloud * na + jj[u]] * 2;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                av[u][a] = A[3 * (size_t)jj[u] + a];
                bv[u][a] = B[3 * (size_t)tt[u] + a];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u * NG_LDS_THREADS + tid < np) {
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    ng_acc[3 * (u * NG_LDS_THREADS + tid) + a] = gg[u] * (av[u][a] - bv[u][a]);
            }
    }
    __syncthreads();
    for (int q0 = 0; q0 < nb; q0 += NG_U * NG_LDS_THREADS) {
        int qq[NG_U], tt[NG_U];
        bool in[NG_U];
        float v[NG_U][3];
#pragma unroll
        for (int u = 0; u < NG_U; ++u) {
            const int q = q0 + u * NG_LDS_THREADS + tid;
            qq[u] = min(q, nb - 1);
            tt[u] = oth_idx[qq[u]];
            in[u] = q < nb && tt[u] >= p0 && tt[u] < p0 + np;
            if (!in[u])
                tt[u] = p0;                                 // (a point that exists; its term is dropped below)
        }
#pragma unroll
        for (int u = 0; u < NG_U; ++u) {
            const float g = uniform != nullptr ? gu : oth_g[(size_t)cloud * nb + qq[u]] * 2;
#pragma unroll
            for (int a = 0; a < 3; ++a)
                v[u][a] = -(g * (B[3 * (size_t)qq[u] + a] - A[3 * (size_t)tt[u] + a]));
        }
#pragma unroll
        for (int u = 0; u < NG_U; ++u) {
            bool live = in[u];
            const int t = tt[u] - p0;
            // Lanes of a wave that name the SAME point add up in registers first (additions to one LDS address take
            // turns): while the first live lane's point is shared by at least four lanes, that group is summed and
            // leaves with one addition per coordinate.
            for (int round = 0; round < 16; ++round) {
                const unsigned long long act = __ballot(live);
                if (act == 0)
                    break;
                const int leader = __ffsll((long long)act) - 1;
                const int lt = __shfl(t, leader);
                const bool same = live && t == lt;
                if (__popcll(__ballot(same)) < 4)
                    break;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float sum = wave_sum(same ? v[u][a] : 0.0f);
                    if (lane == leader)
                        atomicAdd(&ng_acc[3 * lt + a], sum);
                }
                live = live && !same;
            }
            if (live) {
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    atomicAdd(&ng_acc[3 * t + a], v[u][a]);
            }
        }
    }
    __syncthreads();
    float *o = out + ((size_t)cloud * na + p0) * 3;
    for (int i = tid; i < 3 * np; i += NG_LDS_THREADS)
        o[i] = ng_acc[i];
}

// launches the LDS form; false (CLOUDAAE_NND_GRAD_LDS=0): the caller takes the global-atomic form
static bool nn_distance_grad_lds(int b, int n, int m, const float *xyz1, const float *xyz2, const float *g1, const int *idx1,
                                 const float *g2, const int *idx2, float *grad_xyz1, float *grad_xyz2, const float *uniform,
                                 float uniform_scale, hipStream_t s)
{
    if (CLOUDAAE_KNOB("CLOUDAAE_NND_GRAD_LDS", 1) == 0)
        return false;
    const int c1 = ceil_div(n, NG_CH), c2 = ceil_div(m, NG_CH);
    hipLaunchKernelGGL(nn_distance_grad_lds_kernel, dim3(c1 + c2, b), dim3(NG_LDS_THREADS), 0, s, n, m, xyz1, xyz2, g1, idx1, g2,
                       idx2, grad_xyz1, grad_xyz2, c1, uniform, uniform_scale);
    return true;
}

// The same gradients in the ORDER of the reference's CPU loops (tf_nndistance.cpp:126-163): one thread per output point,
// which walks the other cloud's nearest-neighbour indices in ascending order and adds the terms that name it --
//   grad_xyz1[p] = (sweep 1: its own term) then (sweep 2, j ascending with idx2[j] == p) -= g2_j (b_j - a_p)
//   grad_xyz2[t] = (sweep 1, j ascending with idx1[j] == t) -= g1_j (a_j - b_t) then (sweep 2: its own term)
// -- no atomics, every output stored once: bit-identical to the sequential sweep, and to itself from run to run.
// O(n m) index comparisons per cloud (the indices of the other side stream through LDS as broadcast reads):
// ~10 x the time of the atomic kernel, for the deterministic mode.
constexpr int NG_CHUNK = 2048;
__global__ __launch_bounds__(256) void nn_distance_grad_ordered_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ grad_dist1, const int *__restrict__ idx1,
    const float *__restrict__ grad_dist2, const int *__restrict__ idx2,
    float *__restrict__ grad_xyz1, float *__restrict__ grad_xyz2, int blocks1,
    const float *__restrict__ uniform, float uniform_scale)
{
    __shared__ int sidx[NG_CHUNK];
    const int cloud = blockIdx.y;
    const bool second = (int)blockIdx.x >= blocks1;        // outputs of xyz2
    const int blk = second ? (int)blockIdx.x - blocks1 : (int)blockIdx.x;
    const int p = blk * 256 + (int)threadIdx.x;
    const int na = second ? m : n, nb = second ? n : m;     // na: this side's points, nb: the other side's
    const float *A = (second ? xyz2 : xyz1) + (size_t)cloud * na * 3;
    const float *B = (second ? xyz1 : xyz2) + (size_t)cloud * nb * 3;
    const int *own_idx = (second ? idx2 : idx1) + (size_t)cloud * na;
    const int *oth_idx = (second ? idx1 : idx2) + (size_t)cloud * nb;
    const float *own_g = second ? grad_dist2 : grad_dist1, *oth_g = second ? grad_dist1 : grad_dist2;
    float *out = second ? grad_xyz2 : grad_xyz1;
    const bool live = p < na && out != nullptr;
    const float ax = live ? A[3 * p] : 0.0f, ay = live ? A[3 * p + 1] : 0.0f, az = live ? A[3 * p + 2] : 0.0f;
    const float gu = uniform != nullptr ? uniform[0] * uniform_scale : 0.0f;
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    auto own = [&]() {
        const int t = own_idx[p];
        const float g = (uniform != nullptr ? gu : own_g[(size_t)cloud * na + p]) * 2;
        sx += g * (ax - B[3 * (size_t)t]);
        sy += g * (ay - B[3 * (size_t)t + 1]);
        sz += g * (az - B[3 * (size_t)t + 2]);
    };
    if (live && !second)
        own();                                              // sweep 1 writes grad_xyz1[j] first
    for (int c0 = 0; c0 < nb; c0 += NG_CHUNK) {
        const int cnt = min(NG_CHUNK, nb - c0);
        __syncthreads();
        for (int q = threadIdx.x; q < cnt; q += 256)
            sidx[q] = oth_idx[c0 + q];
        __syncthreads();
        if (live)
            for (int q = 0; q < cnt; ++q)
                if (sidx[q] == p) {
                    const int j = c0 + q;
                    const float g = (uniform != nullptr ? gu : oth_g[(size_t)cloud * nb + j]) * 2;
                    sx -= g * (B[3 * (size_t)j] - ax);
                    sy -= g * (B[3 * (size_t)j + 1] - ay);
                    sz -= g * (B[3 * (size_t)j + 2] - az);
                }
    }
    if (live && second)
        own();                                              // sweep 2 adds grad_xyz2[j] last
    if (live) {
        out[((size_t)cloud * na + p) * 3 + 0] = sx;
        out[((size_t)cloud * na + p) * 3 + 1] = sy;
        out[((size_t)cloud * na + p) * 3 + 2] = sz;
    }
}

} // namespace cloudaae

using namespace cloudaae;

static int nn_distance_impl(const char *name, int b, int n, const float *xyz1, int m, const float *xyz2,
                            const long long *count2, const int *row_src2, float *dist1, int *idx1, float *dist2, int *idx2,
                            cloudaae_stream_t stream)
{
    CLOUDAAE_REQUIRE(b >= 0 && n >= 0 && m >= 0, name, "negative size");
    if (b == 0 || (n == 0 && m == 0))
        return 0;
    CLOUDAAE_REQUIRE(b <= 65535, name, "batch > 65535");
    hipStream_t s = (hipStream_t)stream;
    // large clouds: the search on the matrix cores, the reference's arithmetic where it decides
    // (CLOUDAAE_NN_FILTER=0/1 forces the choice, for tests and measurements)
    bool filter = n >= 512 && m >= 512 && (long long)b * ((long long)n + m) >= 64 * NF_QBLOCK;
    if (CLOUDAAE_KNOB_SET("CLOUDAAE_NN_FILTER"))
        filter = CLOUDAAE_KNOB("CLOUDAAE_NN_FILTER", 0) != 0 && n > 0 && m > 0;
    if (filter) {
        const int t1 = ceil_div(n, NF_QBLOCK), t2 = ceil_div(m, NF_QBLOCK);
        // A direction with fewer query blocks than the chip has CUs and many candidates (clouds of unequal size) is
        // cut over its candidates until it has ~4 workgroups per CU, ranges no shorter than one LDS chunk.
        // (Measured at [32,4096]^2, 256 query blocks per direction: two ranges 152 us, one 117 us -- the fixed
        // cost per workgroup, 64 exact evaluations per query, doubles; [32,16384]x[32,1024]: 365 -> 199 us.)
        auto splits_of = [&](int tq, int nc) {
            int sp = 1;
            if ((long long)tq * b < 256 && nc >= 2 * NF_CHUNK) {
                sp = (int)(1024 / ((long long)tq * b));
                if (sp > nc / NF_CHUNK)
                    sp = nc / NF_CHUNK;
                if (sp < 1)
                    sp = 1;
            }
            if (CLOUDAAE_KNOB_SET("CLOUDAAE_NN_SPLIT"))
                sp = CLOUDAAE_KNOB("CLOUDAAE_NN_SPLIT", 1) > 0 ? CLOUDAAE_KNOB("CLOUDAAE_NN_SPLIT", 1) : 1;
            return sp;
        };
        const int s1 = splits_of(t1, m), s2 = splits_of(t2, n);
        unsigned long long *keys = nullptr, *k1 = nullptr, *k2 = nullptr;
        const size_t c1 = s1 > 1 ? (size_t)b * n : 0, c2 = s2 > 1 ? (size_t)b * m : 0;
        if (c1 + c2 > 0) {      // scratch of the call, stream ordered: no state outlives it
            CLOUDAAE_CHECK_HIP(scratch_alloc((void **)&keys, (c1 + c2) * sizeof(unsigned long long), s), name);
            CLOUDAAE_CHECK_HIP(hipMemsetAsync(keys, 0xff, (c1 + c2) * sizeof(unsigned long long), s), name);
            k1 = keys;
            k2 = keys + c1;
        }
        // scores on the bf16 matrix pipe (three-piece split, round 5; <false>, the fp32 matrix instruction of round 4, is
        // no longer instantiated: profiles/notes_chamfer_r5.md has the A/B)
        hipLaunchKernelGGL(nn_distance_filter_kernel<true>, dim3(t1 * s1 + t2 * s2, b), dim3(NF_WAVES * 64), 0, s, n, m, xyz1,
                           xyz2, dist1, idx1, dist2, idx2, t1 * s1, s1, s2, k1, k2, count2);
        if (c1)
            hipLaunchKernelGGL(nn_distance_unpack_kernel, dim3(ceil_div((long long)c1, 256)), dim3(256), 0, s,
                               (long long)c1, k1, dist1, idx1);
        if (c2)
            hipLaunchKernelGGL(nn_distance_unpack_kernel, dim3(ceil_div((long long)c2, 256)), dim3(256), 0, s,
                               (long long)c2, k2, dist2, idx2);
        if (count2 != nullptr && m > 0)
            hipLaunchKernelGGL(nn_distance_expand_kernel, dim3(ceil_div(m, 256), b), dim3(256), 0, s, m, count2, row_src2, xyz2,
                               CLOUDAAE_KNOB("CLOUDAAE_NN_PREFIX_VERIFY", 0), dist2, idx2);
        CLOUDAAE_CHECK_LAUNCH(name);
        if (keys != nullptr)
            CLOUDAAE_CHECK_HIP(hipFreeAsync(keys, s), name);
        return 0;
    }
    // queries per lane: enough workgroups to fill 256 CUs first, then amortise
    // LDS reads over more queries
    const long long total = (long long)b * ((long long)n + m);
    int Q = total >= 4LL * 256 * 1024 ? 4 : (total >= 256LL * 1024 ? 2 : 1);
    const int t1 = ceil_div(n, NN_THREADS * Q), t2 = ceil_div(m, NN_THREADS * Q);
    dim3 grid(t1 + t2, b), block(NN_THREADS);
    if (Q == 4)
        hipLaunchKernelGGL(nn_distance_kernel<4>, grid, block, 0, s, n, m, xyz1, xyz2, dist1, idx1,
                           dist2, idx2, t1, count2);
    else if (Q == 2)
        hipLaunchKernelGGL(nn_distance_kernel<2>, grid, block, 0, s, n, m, xyz1, xyz2, dist1, idx1,
                           dist2, idx2, t1, count2);
    else
        hipLaunchKernelGGL(nn_distance_kernel<1>, grid, block, 0, s, n, m, xyz1, xyz2, dist1, idx1,
                           dist2, idx2, t1, count2);
    if (count2 != nullptr && m > 0)
        hipLaunchKernelGGL(nn_distance_expand_kernel, dim3(ceil_div(m, 256), b), dim3(256), 0, s, m, count2, row_src2, xyz2,
                               CLOUDAAE_KNOB("CLOUDAAE_NN_PREFIX_VERIFY", 0), dist2, idx2);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_dev_nn_split_scores(int nq, int nc, const float *queries, const float *candidates, float *scores,
                                              float *R, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_dev_nn_split_scores";
    CLOUDAAE_REQUIRE(nq >= 1 && nq <= 32 && nc >= 1 && nc <= 32 && queries && candidates && scores && R, name, "1 .. 32 points each");
    hipLaunchKernelGGL(nn_split_scores_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, nq, nc, queries, candidates, scores, R);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_nn_distance(int b, int n, const float *xyz1, int m, const float *xyz2,
                                      float *dist1, int *idx1, float *dist2, int *idx2,
                                      cloudaae_stream_t stream)
{
    return nn_distance_impl("cloudaae_nn_distance", b, n, xyz1, m, xyz2, nullptr, nullptr, dist1, idx1, dist2, idx2, stream);
}

CLOUDAAE_API int cloudaae_nn_distance_prefix(int b, int n, const float *xyz1, int m, const float *xyz2,
                                             const long long *count2, const int *row_src2, float *dist1, int *idx1,
                                             float *dist2, int *idx2, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_nn_distance_prefix";
    CLOUDAAE_REQUIRE((count2 == nullptr) == (row_src2 == nullptr), name, "count2 and row_src2 come together");
    return nn_distance_impl(name, b, n, xyz1, m, xyz2, count2, row_src2, dist1, idx1, dist2, idx2, stream);
}

CLOUDAAE_API int cloudaae_nn_distance_grad(int b, int n, const float *xyz1, int m,
                                           const float *xyz2, const float *grad_dist1,
                                           const int *idx1, const float *grad_dist2,
                                           const int *idx2, float *grad_xyz1, float *grad_xyz2,
                                           cloudaae_stream_t stream)
{
    const char *name = "cloudaae_nn_distance_grad";
    CLOUDAAE_REQUIRE(b >= 0 && n >= 0 && m >= 0, name, "negative size");
    CLOUDAAE_REQUIRE(b <= 65535, name, "batch > 65535");
    hipStream_t s = (hipStream_t)stream;
    if (b > 0 && n > 0 && m > 0 &&
        nn_distance_grad_lds(b, n, m, xyz1, xyz2, grad_dist1, idx1, grad_dist2, idx2, grad_xyz1, grad_xyz2, nullptr, 0.0f, s)) {
        CLOUDAAE_CHECK_LAUNCH(name);        // (every output element stored once: no zero fill)
        return 0;
    }
    // the callee zero-fills, as tf_nndistance_g.cu:153-154 does
    if (grad_xyz1 && (size_t)b * n)
        CLOUDAAE_CHECK_HIP(hipMemsetAsync(grad_xyz1, 0, sizeof(float) * (size_t)b * n * 3, s), name);
    if (grad_xyz2 && (size_t)b * m)
        CLOUDAAE_CHECK_HIP(hipMemsetAsync(grad_xyz2, 0, sizeof(float) * (size_t)b * m * 3, s), name);
    if (b == 0 || n == 0 || m == 0)
        return 0;
    const int b1 = ceil_div(n, 256), b2 = ceil_div(m, 256);
    hipLaunchKernelGGL(nn_distance_grad_kernel, dim3(b1 + b2, b), dim3(256), 0, s, n, m, xyz1, xyz2,
                       grad_dist1, idx1, grad_dist2, idx2, grad_xyz1, grad_xyz2, b1, nullptr, 0.0f);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_nn_distance_grad_uniform(int b, int n, const float *xyz1, int m, const float *xyz2,
                                                   const float *grad, float scale, const int *idx1,
                                                   const int *idx2, float *grad_xyz1, float *grad_xyz2,
                                                   int outputs_zeroed, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_nn_distance_grad_uniform";
    CLOUDAAE_REQUIRE(b >= 0 && n >= 0 && m >= 0 && grad != nullptr, name, "bad argument");
    CLOUDAAE_REQUIRE(b <= 65535, name, "batch > 65535");
    hipStream_t s = (hipStream_t)stream;
    if (b > 0 && n > 0 && m > 0 &&
        nn_distance_grad_lds(b, n, m, xyz1, xyz2, nullptr, idx1, nullptr, idx2, grad_xyz1, grad_xyz2, grad, scale, s)) {
        CLOUDAAE_CHECK_LAUNCH(name);
        return 0;
    }
    if (!outputs_zeroed) {
        if (grad_xyz1 && (size_t)b * n)
            CLOUDAAE_CHECK_HIP(hipMemsetAsync(grad_xyz1, 0, sizeof(float) * (size_t)b * n * 3, s), name);
        if (grad_xyz2 && (size_t)b * m)
            CLOUDAAE_CHECK_HIP(hipMemsetAsync(grad_xyz2, 0, sizeof(float) * (size_t)b * m * 3, s), name);
    }
    if (b == 0 || n == 0 || m == 0)
        return 0;
    const int b1 = ceil_div(n, 256), b2 = ceil_div(m, 256);
    hipLaunchKernelGGL(nn_distance_grad_kernel, dim3(b1 + b2, b), dim3(256), 0, s, n, m, xyz1, xyz2, nullptr, idx1,
                       nullptr, idx2, grad_xyz1, grad_xyz2, b1, grad, scale);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}

CLOUDAAE_API int cloudaae_nn_distance_grad_ordered(int b, int n, const float *xyz1, int m, const float *xyz2,
                                                   const float *grad_dist1, const int *idx1, const float *grad_dist2,
                                                   const int *idx2, const float *uniform, float uniform_scale,
                                                   float *grad_xyz1, float *grad_xyz2, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_nn_distance_grad_ordered";
    CLOUDAAE_REQUIRE(b >= 0 && n >= 0 && m >= 0 && idx1 && idx2, name, "bad argument");
    CLOUDAAE_REQUIRE(uniform != nullptr || (grad_dist1 && grad_dist2), name, "no upstream gradient");
    CLOUDAAE_REQUIRE(b <= 65535, name, "batch > 65535");
    hipStream_t s = (hipStream_t)stream;
    if (b == 0)
        return 0;
    if (n == 0 || m == 0) {     // (an empty cloud: nothing names anything)
        if (grad_xyz1 && n)
            CLOUDAAE_CHECK_HIP(hipMemsetAsync(grad_xyz1, 0, sizeof(float) * (size_t)b * n * 3, s), name);
        if (grad_xyz2 && m)
            CLOUDAAE_CHECK_HIP(hipMemsetAsync(grad_xyz2, 0, sizeof(float) * (size_t)b * m * 3, s), name);
        return 0;
    }
    const int b1 = ceil_div(n, 256), b2 = ceil_div(m, 256);
    hipLaunchKernelGGL(nn_distance_grad_ordered_kernel, dim3(b1 + b2, b), dim3(256), 0, s, n, m, xyz1, xyz2, grad_dist1,
                       idx1, grad_dist2, idx2, grad_xyz1, grad_xyz2, b1, uniform, uniform_scale);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}
