// knn.hip -- fused pairwise distance + k-smallest selection (gfx950).
//
// Replaces tf_util.pairwise_xyz_distance + tf_util.knn (reference
// utils/tf_util.py:597-632), which materialise a [B,N,N] fp32 matrix (134 MB per
// layer at B=32, N=1024) and then run top_k over it.  Here the matrix never
// exists: a lane owns one query point (its features and its sorted k-list live in
// registers), the cloud's points stream through LDS as broadcast reads, and only
// the [B,N,k] int32 indices are written.
//
// Numerics = oracle_knn (oracle/cloudaae_oracle.c):
//   D[i][j] = (|x_i|^2 + (-2 * <x_i,x_j>)) + |x_j|^2             (tf_util.py:618)
//   <,>   : channel-ordered fp32 fma chain from +0 (what v_mfma_f32 computes)
//   |.|^2 : sequential un-fused sum of rounded squares
//   order : ascending D, ties -> lower j (TopKV2)
// A workgroup is 4 waves x 64 queries; wave w scans candidate quarter w in
// ascending j, then wave 0 merges the four sorted lists in wave order with the
// same stable insertion, which preserves the tie rule.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/cloudaae_hip.h"

namespace cloudaae {

constexpr int KNN_WAVES = 4;
constexpr int KNN_THREADS = 64 * KNN_WAVES;

// Dynamic LDS above the default limit must be requested once per kernel AND device.
template <typename F>
static hipError_t raise_lds_limit(F kernel, bool (&raised)[64])
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        dev = 0;
    if (!raised[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess)
            return e;
        raised[dev] = true;
    }
    return hipSuccess;
}


template <int K>
struct TopK {
    float d[K];
    int i[K];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int p = 0; p < K; ++p) {
            d[p] = __builtin_inff();
            i[p] = 0;
        }
    }
    // insert_lex: branch free (a divergent `if` around 2K registers of state drags ~4K register copies along in the
    // compiled code), for newcomers that arrive in ANY order: (d, index) compared lexicographically; (+inf, _) is a no-op
    __device__ __forceinline__ void insert_lex(float nd, int ni)
    {
        const bool in = nd < d[K - 1] || (nd == d[K - 1] && ni < i[K - 1]);
        d[K - 1] = in ? nd : d[K - 1];
        i[K - 1] = in ? ni : i[K - 1];
#pragma unroll
        for (int p = K - 1; p > 0; --p) {
            const bool sw = d[p] < d[p - 1] || (d[p] == d[p - 1] && i[p] < i[p - 1]);
            const float a = d[p - 1], b = d[p];
            const int ia = i[p - 1], ib = i[p];
            d[p - 1] = sw ? b : a;
            d[p] = sw ? a : b;
            i[p - 1] = sw ? ib : ia;
            i[p] = sw ? ia : ib;
        }
    }
    // stable: the newcomer only passes entries that are strictly larger
    __device__ __forceinline__ void insert(float nd, int ni)
    {
        if (nd < d[K - 1]) {
            d[K - 1] = nd;
            i[K - 1] = ni;
#pragma unroll
            for (int p = K - 1; p > 0; --p) {
                const bool sw = d[p] < d[p - 1];
                const float a = d[p - 1], b = d[p];
                const int ia = i[p - 1], ib = i[p];
                d[p - 1] = sw ? b : a;
                d[p] = sw ? a : b;
                i[p - 1] = sw ? ib : ia;
                i[p] = sw ? ia : ib;
            }
        }
    }
};

// A (distance, index) pair as ONE positive double that orders the way the pair does lexicographically: the float's
// bits made monotone (sign folded) times 2^16, plus the index (< 65536); 48 bits, exact.  A sorted insert is then K
// min/max pairs (20 instructions for K = 10, against ~90 for the compare-and-select network on (float, int) pairs).
__device__ __forceinline__ double knn_key(float d, int j)
{
    d = d + 0.0f;                                           // -0 -> +0: they compare equal as floats
    const unsigned b = __float_as_uint(d);
    const unsigned u = b ^ ((unsigned)((int)b >> 31) | 0x80000000u);
    return __builtin_fma((double)u, 65536.0, (double)j);
}
__device__ __forceinline__ int knn_key_index(double key)   // an empty slot (+inf) reads as index 0
{
    const double hi = __builtin_trunc(__builtin_ldexp(key, -16));
    const int j = (int)__builtin_fma(-hi, 65536.0, key);
    return key < __builtin_inf() ? j : 0;
}

template <int K>
struct TopKey {                      // the K smallest keys, ascending
    double key[K];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int p = 0; p < K; ++p)
            key[p] = __builtin_inf();
    }
    __device__ __forceinline__ void insert(double x)
    {
#pragma unroll
        for (int p = 0; p < K; ++p) {
            // (v_min_f64 / v_max_f64 spelled out: __builtin_fmin would first canonicalize the loop-carried key,
            // a third instruction per slot; keys are never NaN)
            double lo, hi;
            asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(key[p]), "v"(x));
            asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(key[p]), "v"(x));
            key[p] = lo;
            x = hi;
        }
    }
    // the same while only the first T slots are occupied (the T-th insert into an empty list): T pairs instead of K
    template <int T>
    __device__ __forceinline__ void insert_first(double x)
    {
        static_assert(T < K, "a full list takes insert()");
#pragma unroll
        for (int p = 0; p < T; ++p) {
            double lo, hi;
            asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(key[p]), "v"(x));
            asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(key[p]), "v"(x));
            key[p] = lo;
            x = hi;
        }
        key[T] = x;
    }
};

template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F &&>(f));
    }
}

// index of an orderable key that is not +inf: its low 16 bits (the key is an integer below 2^48: exact after + 2^52)
__device__ __forceinline__ int knn_key_low16(double key)
{
    return (int)(__double_as_longlong(key + 4503599627370496.0) & 0xffff);
}

template <int K>
__device__ __forceinline__ void merge_and_store(TopK<K> &top, float *mbuf_d, int *mbuf_i, int wave,
                                                int lane, bool valid, int k, int *dst)
{
    // lists of waves 1..3 go through LDS, [wave-1][p][lane]; the buffer aliases
    // the scan buffers, so every wave must have finished scanning first
    __syncthreads();
    if (wave > 0) {
#pragma unroll
        for (int p = 0; p < K; ++p) {
            mbuf_d[((wave - 1) * K + p) * 64 + lane] = top.d[p];
            mbuf_i[((wave - 1) * K + p) * 64 + lane] = top.i[p];
        }
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 0; w < KNN_WAVES - 1; ++w) {
#pragma unroll
            for (int p = 0; p < K; ++p)
                top.insert(mbuf_d[(w * K + p) * 64 + lane], mbuf_i[(w * K + p) * 64 + lane]);
        }
        if (valid) {
#pragma unroll
            for (int p = 0; p < K; ++p)
                if (p < k)
                    dst[p] = top.i[p];
        }
    }
}

// ---- C = 3 (xyz slice of a [*, ld] row) -----------------------------------
constexpr int KNN3_CHUNK = 256;  // candidates per wave per LDS refill

template <int K>
__global__ __launch_bounds__(KNN_THREADS) void knn3_kernel(int n, int ld, int k,
                                                           const float *__restrict__ x,
                                                           int *__restrict__ nn_idx)
{
    // scan buffer (x, y, z, |.|^2 per candidate) and merge buffer share LDS
    constexpr int SCAN_BYTES = KNN_WAVES * KNN3_CHUNK * 16;
    constexpr int MERGE_BYTES = (KNN_WAVES - 1) * K * 64 * 8;
    __shared__ __attribute__((aligned(16))) char smem[SCAN_BYTES > MERGE_BYTES ? SCAN_BYTES : MERGE_BYTES];
    float4v(*cand)[KNN3_CHUNK] = reinterpret_cast<float4v(*)[KNN3_CHUNK]>(smem);
    float *mbuf_d = reinterpret_cast<float *>(smem);
    int *mbuf_i = reinterpret_cast<int *>(smem) + (KNN_WAVES - 1) * K * 64;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tile, cloud;
    xcd_cloud_tile(tile, cloud);   // a cloud's candidate rows stay in ONE XCD's L2
    const float *X = x + (size_t)cloud * n * ld;
    const int i = tile * 64 + lane;
    const bool valid = i < n;
    const int ii = valid ? i : 0;
    const float qx = X[(size_t)ii * ld], qy = X[(size_t)ii * ld + 1], qz = X[(size_t)ii * ld + 2];
    float sqi;
    {
        const float a = qx * qx, b = qy * qy, c = qz * qz;
        sqi = 0.0f + a;
        sqi = sqi + b;
        sqi = sqi + c;
    }
    TopK<K> top;
    top.init();

    const int per = (n + KNN_WAVES - 1) / KNN_WAVES;
    const int j_begin = min(wave * per, n), j_end = min(j_begin + per, n);
    const int rounds = (per + KNN3_CHUNK - 1) / KNN3_CHUNK;  // same for all waves
    for (int r = 0; r < rounds; ++r) {
        const int c0 = j_begin + r * KNN3_CHUNK;
        const int cnt = max(0, min(KNN3_CHUNK, j_end - c0));
        __syncthreads();
        for (int s = lane; s < cnt; s += 64) {
            const float *row = X + (size_t)(c0 + s) * ld;
            const float cx = row[0], cy = row[1], cz = row[2];
            const float a = cx * cx, b = cy * cy, c = cz * cz;
            float sq = 0.0f + a;
            sq = sq + b;
            sq = sq + c;
            cand[wave][s] = float4v{cx, cy, cz, sq};
        }
        __syncthreads();
        for (int s = 0; s < cnt; ++s) {
            const float4v c = cand[wave][s];
            float inner = fmaf(qx, c.x, 0.0f);
            inner = fmaf(qy, c.y, inner);
            inner = fmaf(qz, c.z, inner);
            const float m2 = -2.0f * inner;
            const float t = sqi + m2;
            const float d = t + c.w;
            top.insert(d, c0 + s);
        }
    }
    merge_and_store<K>(top, mbuf_d, mbuf_i, wave, lane, valid, k,
                       nn_idx + ((size_t)cloud * n + ii) * k);
}

// ---- any other channel count (model variants outside the named configs) ----
// one lane per query, features re-read from global per candidate; slow, exact.
template <int K>
__global__ __launch_bounds__(KNN_THREADS) void knn_generic_kernel(int n, int c, int ld, int k,
                                                                  const float *__restrict__ x,
                                                                  int *__restrict__ nn_idx)
{
    const int cloud = blockIdx.y;
    const float *X = x + (size_t)cloud * n * ld;
    const int i = blockIdx.x * KNN_THREADS + threadIdx.x;
    if (i >= n)
        return;
    float sqi = 0.0f;
    for (int ch = 0; ch < c; ++ch) {
        const float v = X[(size_t)i * ld + ch];
        const float v2 = v * v;
        sqi = sqi + v2;
    }
    TopK<K> top;
    top.init();
    for (int j = 0; j < n; ++j) {
        float inner = 0.0f, sqj = 0.0f;
        for (int ch = 0; ch < c; ++ch) {
            const float v = X[(size_t)j * ld + ch];
            const float v2 = v * v;
            sqj = sqj + v2;
            inner = fmaf(X[(size_t)i * ld + ch], v, inner);
        }
        const float m2 = -2.0f * inner;
        const float t = sqi + m2;
        top.insert(t + sqj, j);
    }
    for (int p = 0; p < k && p < K; ++p)
        nn_idx[((size_t)cloud * n + i) * k + p] = top.i[p];
}

// ---- C = 64 on the matrix cores -----------------------------------------------------------
// The inner products of a 32-candidate x 32-query tile are ONE chain of 32
// v_mfma_f32_32x32x2_f32 (channels (2s, 2s+1) in step s): bitwise the channel-ordered fma
// chain of the oracle, at the matrix-pipe rate and with almost no LDS traffic (the VALU
// kernel above is LDS-bound: 16 broadcast ds_read_b128 per candidate per wave).  A
// workgroup = 4 waves = the SAME 32 queries; wave w scans candidate quarter w.  In the
// accumulator layout a lane holds query column (lane & 31) and 16 candidate rows, so lanes l
// and l+32 keep separate sorted k-lists for the same query over disjoint candidates; the
// 8 lists of a query (4 waves x 2 half-waves) are merged lexicographically by
// (distance, index) at the end, which is exactly "ascending distance, ties -> lower index".
// The selection (VALU) of one wave overlaps the MFMA chain of the others.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KM_TILE = 32;
constexpr int KM_LD = 65;   // LDS row stride of a staged candidate tile (odd: conflict-free column reads)

template <int K>
__global__ __launch_bounds__(KNN_THREADS) void knn64_mfma_kernel(int n, int ld, int k,
                                                                 const float *__restrict__ x,
                                                                 int *__restrict__ nn_idx)
{
    constexpr int TILE_FLOATS = KM_TILE * KM_LD + KM_TILE;            // rows + their |.|^2
    constexpr int SCAN_BYTES = KNN_WAVES * TILE_FLOATS * 4;
    constexpr int MERGE_BYTES = 2 * KNN_WAVES * K * 32 * 8;           // 8 lists x K x 32 queries x (d, idx)
    __shared__ __attribute__((aligned(16))) char smem[SCAN_BYTES > MERGE_BYTES ? SCAN_BYTES : MERGE_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *tile = reinterpret_cast<float *>(smem) + wave * TILE_FLOATS;
    float *csq = tile + KM_TILE * KM_LD;

    int qtile, cloud;
    xcd_cloud_tile(qtile, cloud);
    const float *X = x + (size_t)cloud * n * ld;
    const int col = lane & 31, half = lane >> 5;
    const int qi = qtile * KM_TILE + col;               // this lane's query
    const bool qvalid = qi < n;
    const int qs = qvalid ? qi : 0;

    // B operand: query channels of parity `half`, one register per MFMA step; and |x_q|^2
    float bq[32];
    float sqi = 0.0f;
    {
        const float *row = X + (size_t)qs * ld;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float4v v = *reinterpret_cast<const float4v *>(row + 4 * g);
            const float a = v.x * v.x, b = v.y * v.y, c = v.z * v.z, d = v.w * v.w;
            sqi = sqi + a;
            sqi = sqi + b;
            sqi = sqi + c;
            sqi = sqi + d;
            bq[2 * g] = half ? v.y : v.x;               // channels 4g + half, 4g + 2 + half
            bq[2 * g + 1] = half ? v.w : v.z;
        }
    }
    TopK<K> top;
    top.init();

    const int per = ((n + KNN_WAVES - 1) / KNN_WAVES + KM_TILE - 1) / KM_TILE * KM_TILE;
    const int j_begin = min(wave * per, n), j_end = min(j_begin + per, n);
    const int rounds = per / KM_TILE;                    // identical for all waves
    const int srow = lane >> 1, shalf = lane & 1;        // two lanes stage one candidate row
    for (int r = 0; r < rounds; ++r) {
        const int c0 = j_begin + r * KM_TILE;
        const int cnt = max(0, min(KM_TILE, j_end - c0));
        __syncthreads();
        {
            const bool ok = srow < cnt;
            const float *row = X + (size_t)(ok ? c0 + srow : 0) * ld + 32 * shalf;
            float4v v[8];
#pragma unroll
            for (int g = 0; g < 8; ++g)
                v[g] = ok ? *reinterpret_cast<const float4v *>(row + 4 * g) : float4v{0, 0, 0, 0};
            float lo = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float a = v[g].x * v[g].x, b = v[g].y * v[g].y, c = v[g].z * v[g].z, d = v[g].w * v[g].w;
                lo = lo + a;
                lo = lo + b;
                lo = lo + c;
                lo = lo + d;
            }
            float part = __shfl(lo, lane & ~1, 64);      // the odd lane continues the even lane's sum
            if (shalf) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const float a = v[g].x * v[g].x, b = v[g].y * v[g].y, c = v[g].z * v[g].z,
                                d = v[g].w * v[g].w;
                    part = part + a;
                    part = part + b;
                    part = part + c;
                    part = part + d;
                }
                csq[srow] = part;
            }
            float *dst = tile + srow * KM_LD + 32 * shalf;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                dst[4 * g + 0] = v[g].x;
                dst[4 * g + 1] = v[g].y;
                dst[4 * g + 2] = v[g].z;
                dst[4 * g + 3] = v[g].w;
            }
        }
        __syncthreads();
        if (cnt > 0) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                acc[e] = 0.0f;
            const float *arow = tile + col * KM_LD + half;   // A operand: candidate row `col`, parity `half`
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                // step s covers channels 2s (lanes 0-31) and 2s+1 (lanes 32-63); bq[] is stored in
                // the same order: bq[s] = channel 2s + half
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * s], bq[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rr = (e & 3) + 8 * (e >> 2) + 4 * half;    // candidate row of acc[e]
                if (rr < cnt) {
                    const float m2 = -2.0f * acc[e];
                    const float t = sqi + m2;
                    top.insert(t + csq[rr], c0 + rr);
                }
            }
        }
    }

    // merge the 8 sorted lists of every query: [list][p][query]
    __syncthreads();
    float *md = reinterpret_cast<float *>(smem);
    int *mi = reinterpret_cast<int *>(smem) + 2 * KNN_WAVES * K * 32;
    const int list = wave * 2 + half;
#pragma unroll
    for (int p = 0; p < K; ++p) {
        md[(list * K + p) * 32 + col] = top.d[p];
        mi[(list * K + p) * 32 + col] = top.i[p];
    }
    __syncthreads();
    if (wave == 0 && half == 0 && qvalid) {
        int head[2 * KNN_WAVES];
#pragma unroll
        for (int l = 0; l < 2 * KNN_WAVES; ++l)
            head[l] = 0;
        int *dst = nn_idx + ((size_t)cloud * n + qi) * k;
        for (int p = 0; p < k; ++p) {
            float bd = __builtin_inff();
            int bi = 0x7fffffff, bl = 0;
#pragma unroll
            for (int l = 0; l < 2 * KNN_WAVES; ++l) {
                const int h = head[l];
                const float d = h < K ? md[(l * K + h) * 32 + col] : __builtin_inff();
                const int i = h < K ? mi[(l * K + h) * 32 + col] : 0x7fffffff;
                const bool better = d < bd || (d == bd && i < bi);
                bd = better ? d : bd;
                bi = better ? i : bi;
                bl = better ? l : bl;
            }
#pragma unroll
            for (int l = 0; l < 2 * KNN_WAVES; ++l)
                head[l] += (l == bl) ? 1 : 0;
            dst[p] = bi == 0x7fffffff ? 0 : bi;
        }
    }
}

// ---- C = 64, second generation: one wave per 32-query tile scans the WHOLE cloud -------------
// The kernel above gives every query eight short lists (4 candidate quarters x 2 lane halves); each
// list sees only n/8 candidates, so nearly every candidate still passes its threshold and the
// 40-instruction sorted insert runs for all 64 lanes almost every time -- the selection network,
// not the MFMAs, is what that kernel spends its time on (MFMA utilisation 17 %).
// Here a workgroup owns QW consecutive 32-query tiles of one cloud (one wave each) and walks ALL
// candidates in 32-row tiles that the waves stage cooperatively (each row is fetched once per
// workgroup instead of once per wave) into a double-buffered LDS tile, prefetched one tile ahead
// through registers.  A lane keeps ONE list per (query, lane half), so thresholds tighten fast, and
// the selection is split in two:
//   filter : d < (current k-th best of this lane)?  -> push (d, j) on the lane's LDS queue
//            (three instructions; runs for every candidate)
//   drain  : when some lane's queue could overflow next round, every lane pops its queue through
//            the sorted insert; the wave pays max-over-lanes pops instead of one insert per
//            candidate for which ANY lane passes.
// |x_j|^2 of the whole cloud is computed once per workgroup (sequential un-fused sum, as the oracle
// defines it) and kept in LDS.  Arithmetic and tie rule are those of the kernels above, so the
// indices stay bit-identical to oracle_knn.
constexpr int KS_QCAP = 24;          // queue slots per lane; a round pushes at most 16
#ifndef KS_POP_V
#define KS_POP_V 6
#endif
constexpr int KS_POP = KS_POP_V;      // entries popped per round when the queue is not about to overflow

// QW query tiles per workgroup, CS waves per query tile (wave cs scans the candidate tiles t = r*CS + cs)
template <int K, int QW, int CS>
__global__ __launch_bounds__(64 * QW * CS) void knn64_scan_kernel(int n, int ld, int k,
                                                                  const float *__restrict__ x,
                                                                  int *__restrict__ nn_idx)
{
    constexpr int WAVES = QW * CS, THREADS = 64 * WAVES;
    // staged tile row = [32 even channels | 32 odd channels | 4 pad]: a lane's 32 MFMA operands (channel
    // parity = lane half) are contiguous, eight ds_read_b128 instead of 32 ds_read_b32; 68-float rows keep
    // the 16-byte reads of 16 consecutive rows on distinct banks
    constexpr int KS_LD = 68;
    constexpr int TILE_FLOATS = KM_TILE * KS_LD;
    extern __shared__ __attribute__((aligned(16))) char ks_smem[];
    // layout: tile[2][CS][TILE_FLOATS] | queue d[WAVES][QCAP][64] | queue i[WAVES][QCAP][64] | sq[n]
    float *tiles = reinterpret_cast<float *>(ks_smem);
    float *qd_all = tiles + 2 * CS * TILE_FLOATS;
    int *qi_all = reinterpret_cast<int *>(qd_all + WAVES * KS_QCAP * 64);
    float *sq = reinterpret_cast<float *>(qi_all + WAVES * KS_QCAP * 64);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qt = wave / CS, cs = wave % CS;
    int qgroup, cloud;
    xcd_cloud_tile(qgroup, cloud);
    const float *X = x + (size_t)cloud * n * ld;
    float *qd = qd_all + wave * KS_QCAP * 64;
    int *qi = qi_all + wave * KS_QCAP * 64;

    // |x_j|^2 for every point of the cloud
    for (int j = tid; j < n; j += THREADS) {
        const float *row = X + (size_t)j * ld;
        float acc = 0.0f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float4v v = *reinterpret_cast<const float4v *>(row + 4 * g);
            const float a = v.x * v.x, b = v.y * v.y, c = v.z * v.z, d = v.w * v.w;
            acc = acc + a;
            acc = acc + b;
            acc = acc + c;
            acc = acc + d;
        }
        sq[j] = acc;
    }

    const int col = lane & 31, half = lane >> 5;
    const int qi0 = (qgroup * QW + qt) * KM_TILE + col;       // this lane's query
    const bool qvalid = qi0 < n;
    const int qs = qvalid ? qi0 : 0;
    // B operand: query channels of parity `half`, one register per MFMA step
    float bq[32];
    {
        const float *row = X + (size_t)qs * ld;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float4v v = *reinterpret_cast<const float4v *>(row + 4 * g);
            bq[2 * g] = half ? v.y : v.x;               // channels 4g + half, 4g + 2 + half
            bq[2 * g + 1] = half ? v.w : v.z;
        }
    }

    // staging map: a round = CS tiles of 32 rows x 16 float4, spread over the workgroup
    constexpr int VECS = CS * KM_TILE * 16, PER = (VECS + THREADS - 1) / THREADS;
    float4v stage[PER];
    auto fetch = [&](int r) {
        const int c0 = r * CS * KM_TILE;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int v = u * THREADS + tid;
            const int row = v >> 4, q4 = v & 15;
            stage[u] = (v < VECS && c0 + row < n)
                           ? *reinterpret_cast<const float4v *>(X + (size_t)(c0 + row) * ld + 4 * q4)
                           : float4v{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto commit = [&](float *buf) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int v = u * THREADS + tid;
            if (VECS % THREADS == 0 || v < VECS) {
                const int row = v >> 4, q4 = v & 15;      // row in [0, CS*32): tile row / 32, line row % 32
                // channels 4*q4 .. 4*q4+3: (x, z) are even channels 2*q4, 2*q4+1 of the even half, (y, w) odd
                float *dst = buf + (row >> 5) * TILE_FLOATS + (row & 31) * KS_LD + 2 * q4;
                *reinterpret_cast<float2v *>(dst) = float2v{stage[u].x, stage[u].z};
                *reinterpret_cast<float2v *>(dst + 32) = float2v{stage[u].y, stage[u].w};
            }
        }
    };

    TopK<K> top;
    top.init();
    float thr = __builtin_inff();
    int cnt = 0;
    // The queue is popped a few entries per round (pop_some) rather than all at once when it fills up:
    // the waves of a workgroup meet at a barrier every round, and a wave that stops to pop twenty
    // entries makes the other seven wait -- with eight waves nearly every round had such a wave.
    int head = 0;
    auto pop_some = [&](int limit) {
        for (int t = 0; t < limit && __any(head < cnt); ++t)
            if (head < cnt) {
                top.insert(qd[head * 64 + lane], qi[head * 64 + lane]);
                ++head;
            }
        if (head >= cnt) {
            head = 0;
            cnt = 0;
        }
        thr = top.d[K - 1];
    };
    auto drain = [&]() { pop_some(KS_QCAP); };

    const int ntiles = (n + KM_TILE - 1) / KM_TILE;
    const int rounds = (ntiles + CS - 1) / CS;
    fetch(0);
    commit(tiles);
    __syncthreads();                                      // round 0 and sq[] visible
    const float sqi = sq[qs];
    // filter of one finished tile: acc[e] = <x_q, x_{c0 + row(e)}>, csq[e] = |x_{c0 + row(e)}|^2 (read
    // from LDS ahead of time: the queue lives in the same LDS array, so the compiler cannot move
    // those reads across the queue writes by itself)
    auto filter = [&](const f32x16 &acc, const float (&csq)[16], int c0) {
        const int cnt_rows = min(KM_TILE, n - c0);        // <= 0 for a tile past the end
        if (__any(cnt > KS_QCAP - 17))
            drain();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rr = (e & 3) + 8 * (e >> 2) + 4 * half;    // candidate row of acc[e]
            const float m2 = -2.0f * acc[e];
            const float tt = sqi + m2;
            const float d = tt + csq[e];
            // branch-free push: the slot is always written, the count moves only for a pass
            qd[cnt * 64 + lane] = d;
            qi[cnt * 64 + lane] = c0 + rr;
            cnt += (rr < cnt_rows && d < thr) ? 1 : 0;
        }
    };
    auto load_csq = [&](float (&csq)[16], int c0) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
            csq[e] = sq[min(c0 + (e & 3) + 8 * (e >> 2) + 4 * half, n - 1)];
    };
    f32x16 prev;
    float pcsq[16];
    for (int r = 0; r < rounds; ++r) {
        const int c0 = (r * CS + cs) * KM_TILE;           // this wave's tile of the round
        const float *cur = tiles + ((r & 1) * CS + cs) * TILE_FLOATS;
        const float4v *arow = reinterpret_cast<const float4v *>(cur + col * KS_LD + 32 * half);   // candidate row `col`
        float aop[32], ccsq[16];
#pragma unroll
        for (int s = 0; s < 8; ++s) {                     // all LDS reads of the tile in flight at once
            const float4v v = arow[s];
            aop[4 * s] = v.x;
            aop[4 * s + 1] = v.y;
            aop[4 * s + 2] = v.z;
            aop[4 * s + 3] = v.w;
        }
        load_csq(ccsq, c0);
        if (r + 1 < rounds)
            fetch(r + 1);                                 // global -> registers behind the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            acc[e] = 0.0f;
        if (r > 0) {
            if (__any(cnt > KS_QCAP - 17))
                drain();
            else
                pop_some(KS_POP);
        }
        // the filter of the PREVIOUS tile (VALU + LDS pushes, independent of acc) is issued between the
        // MFMAs of this tile, two MFMAs per candidate row, so it runs in their shadow
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[s], bq[s], acc, 0, 0, 0);
            if (r > 0 && (s & 1)) {
                const int e = s >> 1;
                const int pc0 = c0 - CS * KM_TILE;
                const int rr = (e & 3) + 8 * (e >> 2) + 4 * half;
                const float m2 = -2.0f * prev[e];
                const float tt = sqi + m2;
                const float d = tt + pcsq[e];
                qd[cnt * 64 + lane] = d;
                qi[cnt * 64 + lane] = pc0 + rr;
                cnt += (rr < min(KM_TILE, n - pc0) && d < thr) ? 1 : 0;
            }
        }
        prev = acc;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            pcsq[e] = ccsq[e];
        if (r + 1 < rounds)
            commit(tiles + ((r + 1) & 1) * CS * TILE_FLOATS);   // last read of that buffer: one barrier ago
        __syncthreads();
    }
    filter(prev, pcsq, ((rounds - 1) * CS + cs) * KM_TILE);
    drain();

    // merge the 2*CS lists of every query lexicographically by (d, j): each is already sorted that
    // way (its candidates arrive in ascending j, the insert is stable).  The lists go through the
    // queue area of the query tile's first wave: [list][p][query].
    __syncthreads();
    float *md = qd_all + (qt * CS) * KS_QCAP * 64;
    int *mi = qi_all + (qt * CS) * KS_QCAP * 64;
    static_assert(2 * CS * K * 32 <= CS * KS_QCAP * 64, "merge lists must fit the queue area of one query tile");
    const int list = cs * 2 + half;
#pragma unroll
    for (int p = 0; p < K; ++p) {
        md[(list * K + p) * 32 + col] = top.d[p];
        mi[(list * K + p) * 32 + col] = top.i[p];
    }
    __syncthreads();
    // (the merging lanes of the query tiles sit in different SIMDs: wave qt * CS + cs, SIMD = wave % 4)
    if (cs == ((qt * CS) >> 2) % CS && half == 0 && qvalid) {
        int head[2 * CS];
#pragma unroll
        for (int l = 0; l < 2 * CS; ++l)
            head[l] = 0;
        int *dst = nn_idx + ((size_t)cloud * n + qi0) * k;
        for (int p = 0; p < k; ++p) {
            float bd = __builtin_inff();
            int bi = 0x7fffffff, bl = 0;
#pragma unroll
            for (int l = 0; l < 2 * CS; ++l) {
                const int h = head[l];
                const float d = h < K ? md[(l * K + h) * 32 + col] : __builtin_inff();
                const int i = h < K ? mi[(l * K + h) * 32 + col] : 0x7fffffff;
                const bool better = d < bd || (d == bd && i < bi);
                bd = better ? d : bd;
                bi = better ? i : bi;
                bl = better ? l : bl;
            }
#pragma unroll
            for (int l = 0; l < 2 * CS; ++l)
                head[l] += (l == bl) ? 1 : 0;
            dst[p] = bi == 0x7fffffff ? 0 : bi;
        }
    }
}

template <int K, int QW, int CS>
static hipError_t launch_knn_scan(int b, int n, int ld, int k, const float *x, int *nn_idx, hipStream_t s)
{
    const size_t lds = sizeof(float) * (2 * CS * KM_TILE * 68 + 2 * QW * CS * KS_QCAP * 64 + (size_t)n);
    static bool raised[64] = {};
    if (hipError_t e = raise_lds_limit(&knn64_scan_kernel<K, QW, CS>, raised); e != hipSuccess)
        return e;
    hipLaunchKernelGGL((knn64_scan_kernel<K, QW, CS>), dim3(ceil_div(n, KM_TILE * QW), b), dim3(64 * QW * CS), lds, s,
                       n, ld, k, x, nn_idx);
    return hipSuccess;
}

// ---- C = 64, third generation: a BOUND on the k-th distance first, then one filtered scan ------------
// What the scan kernel above still spends its time on is the start of every lane's stream: until a lane's
// list has tightened, nearly every candidate passes its running threshold (k (1 + ln(n/k)) ~ 50 sorted inserts
// per lane at n = 512 per lane, the wave paying the maximum over its lanes, six pops per round whether needed
// or not).  Here the threshold is known BEFORE the scan:
//   pass A : a quarter of the candidate tiles (every stride-th one).  Per lane and tile the minimum of every
//            group of four candidate rows (a "unit") goes into a value-only sorted list; afterwards the K-th
//            smallest unit minimum over the query's 2*CS lane lists is tau.  The K smallest unit minima belong
//            to K DISTINCT candidates, so tau >= the true K-th smallest distance; with 64 units of four
//            (n = 1024) about 40 of the 1024 candidates lie at or below it.
//   pass B : every tile; a candidate with d <= tau is APPENDED to its query's LDS queue (shared by the query's
//            2*CS lanes, slots handed out by an LDS atomic; ~46 of 1024 candidates).  The sorted lists are built
//            once, after the scan: each of the query's lanes takes every (2*CS)-th queue entry through a
//            branch-free lexicographic insert, then the 2*CS lists are merged by (d, j) as before.  A query whose
//            queue is full (duplicated points, adversarial clouds) raises the workgroup's flag and the workgroup
//            repeats the scan the plain way (sorted insert per candidate): correctness never depends on the bound.
// Both passes evaluate d with the SAME instructions on the same MFMA results, so "d <= tau" in pass B is exact
// and the indices stay bit-identical to oracle_knn.  Matrix work: 1.25 x the N x N x 64 products.

// the LDS address of a pointer into shared memory (a flat LDS address is aperture : offset)
__device__ __forceinline__ unsigned lds_offset(const void *p) { return (unsigned)(uintptr_t)p; }

template <int K>
struct MinK {                        // the K smallest values seen, ascending
    float d[K];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int p = 0; p < K; ++p)
            d[p] = __builtin_inff();
    }
    __device__ __forceinline__ void insert(float nd)
    {
        d[K - 1] = fminf(d[K - 1], nd);
#pragma unroll
        for (int p = K - 1; p > 0; --p) {
            const float a = d[p - 1], b = d[p];
            d[p - 1] = fminf(a, b);
            d[p] = fmaxf(a, b);
        }
    }
};

// queue slots per query of knn64_wide_kernel: 144 while the cloud's norms leave room for them, 128 above
static int knn_wide_qpq(int n) { return n <= 3328 ? 144 : 128; }
static size_t knn_wide_lds_bytes(int n)
{
    return sizeof(float) * (4 * KM_TILE * 68 + (size_t)ceil_div(n, KM_TILE) * KM_TILE + 4 + 128 + (size_t)ceil_div(n, KM_TILE)) +
           (sizeof(float) + sizeof(unsigned short)) * 128 * (size_t)knn_wide_qpq(n) + sizeof(int) * 128;
}

// ---- C = 64, bound pass + filtered scan with FOUR waves per SIMD (16-wave workgroups) ---------------------------
// knn64_bound_kernel above leaves the matrix pipe idle two thirds of the time: with one 8-wave workgroup per CU
// (256 workgroups for 256 CUs at B = 32) two lock-stepped waves share a SIMD, and each wave's own serial chain --
// its dependent MFMAs, its vector instructions, its LDS and barrier waits -- sets the kernel time.  Here a query
// tile is scanned by CS = 4 waves (a quarter of the candidate tiles each), a workgroup is QW x CS = 16 waves, so
// every SIMD has four waves to overlap and each wave's chain is half as long.  What that costs: 128 registers per
// lane (no operand double-buffering, the filter runs right after its tile's MFMAs) and one LDS tile buffer per wave
// slot (two barriers per round: tiles landed / operands read, the second one inside the MFMA chain).  What else differs
// from the kernel above:
//   * tiles go global -> LDS without passing through registers (global_load_lds_dword, one staged row per instruction);
//   * |x_j|^2 is computed from the staged tiles (two waves, one row per lane, after the round's first barrier), so the
//     workgroup never reads the cloud a second time;
//   * sorted inserts and merges run on ONE orderable double per (distance, index) pair (knn_key / TopKey): 10 pairs of
//     v_min_f64 / v_max_f64 per insert;
//   * the merging lanes of the four query tiles sit in waves 0, 5, 10, 15: one per SIMD.
// Same arithmetic, same bound, same queues (144 six-byte entries per query), same flagged fallback.
// K = 20 (BASELINE configs[4]: k = 20 neighbours, 4096 points): the bound comes from HALF of the candidate tiles
// instead of a quarter (the expected number of candidates at or below tau is k x tiles / sampled tiles: 80 of 4096
// from a quarter, with a tail that overflows the 128-slot queues; 40 from half), and the final merge runs in two
// stages (the eight key lists of a query, 1280 bytes, do not fit its 768-byte share of the queue area: the lane
// halves merge through registers first, four lists go through LDS).
// REUSE (the launcher sets it when pass A is at most two rounds: K <= 10, n <= 1024): the sampled tiles' distances stay
// in registers until tau is known, go through the filter then, and pass B covers only the tiles that were NOT sampled:
// 1.0 x the N x N x 64 products instead of 1.25 x, two rounds fewer.
// TWO (every other launch): the bound in two stages.  The sample's even slots give a first bound (pass A1); the odd slots
// are scanned WITH it (pass A2: their candidates at or below it go to the queues, their unit minima join the lists), the
// K-th smallest over both halves is the final bound, and pass B leaves A2's tiles out: 1.25 x the products instead of
// 1.5 x for K = 20 (half of the tiles sampled), 1.125 x instead of 1.25 x for K = 10.  The second merge finds the queue
// area occupied, so its lists (the 8 smallest of every lane: the K-th over fewer values is still a bound) go through the
// tile buffers, which are idle at that point.
template <int K, int QPQ, bool REUSE, bool TWO>
__global__ __launch_bounds__(1024) void knn64_wide_kernel(int n, int ld, int k, const float *__restrict__ x,
                                                          int *__restrict__ nn_idx, const unsigned char *__restrict__ only)
{
    // only != nullptr: the launch repairs the query groups knn64_split_kernel flagged (a byte per workgroup of this grid)
    if (only != nullptr) {
        int qg, cl;
        xcd_cloud_tile(qg, cl);
        if (only[cl * gridDim.x + qg] == 0)
            return;
    }
    constexpr int QW = 4, CS = 4, THREADS = 1024;
    constexpr int KS_LD = 68;                              // staged row: [32 even channels | 32 odd | 4 pad]
    constexpr int TILE_FLOATS = KM_TILE * KS_LD;
    // QPQ: queue slots per query (its 8 lanes share them)
    extern __shared__ __attribute__((aligned(16))) char kw_smem[];
    // layout: tile[CS][TILE_FLOATS] | queue d[128 queries][QPQ] (fp32) | queue j, same shape (u16) | queue lengths [128] |
    //         sq[ntiles * 32] | 1.0 | overflow flag | (pad) | tau [128 queries] | REUSE / TWO: pass B's tile list [ntiles]
    float *tiles = reinterpret_cast<float *>(kw_smem);
    float *qd_all = tiles + CS * TILE_FLOATS;
    unsigned short *qj_all = reinterpret_cast<unsigned short *>(qd_all + QW * 32 * QPQ);
    int *qn_all = reinterpret_cast<int *>(qj_all + QW * 32 * QPQ);
    float *sq = reinterpret_cast<float *>(qn_all + QW * 32);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qt = wave / CS, cs = wave % CS;
    int qgroup, cloud;
    xcd_cloud_tile(qgroup, cloud);
    const float *X = x + (size_t)cloud * n * ld;
    const int ntiles = (n + KM_TILE - 1) / KM_TILE;

    for (int j = tid; j < QW * 32; j += THREADS)
        qn_all[j] = 0;
    // sq[j] = |x_j|^2, filled in as tiles are staged (norms() below); +inf until then and past the end; behind it 1.0
    // (the 33rd step's other operand) and 0 = the workgroup's "a queue overflowed" flag
    for (int j = tid; j <= ntiles * KM_TILE + 1; j += THREADS)
        sq[j] = j == ntiles * KM_TILE ? 1.0f : (j > ntiles * KM_TILE ? 0.0f : __builtin_inff());

    const int col = lane & 31, half = lane >> 5;
    const int qi0 = (qgroup * QW + qt) * KM_TILE + col;   // this lane's query
    const bool qvalid = qi0 < n;
    const int qs = qvalid ? qi0 : qgroup * QW * KM_TILE;   // (a row whose norm the prologue computes)
    // pass A's sample: S tiles, every stride-th one (a quarter of the tiles; half of them for K > 10)
    const int S = min(ntiles, max((ntiles + (K > 10 ? 1 : 3)) / (K > 10 ? 2 : 4), 4));
    const int stride = ntiles / S;
    static_assert(!(REUSE && TWO), "one or the other");
    const int SA1 = TWO ? (S + 1) / 2 : S, SA2 = TWO ? S / 2 : 0;         // sample slots of pass A1 (TWO: the even ones) / A2
    // pass B's tile list = the tiles whose distances are not kept, ascending: all but the sample (REUSE) / all but A2's
    // tiles (TWO).  (A table in LDS: computed where it is needed, the integer divisions cost every wave ~45 vector
    // instructions per round, and a vector instruction costs matrix time here -- see the round below)
    const int nB = REUSE ? ntiles - S : ntiles - SA2;
    float *tauv = sq + ntiles * KM_TILE + 4;
    int *tile_list = reinterpret_cast<int *>(tauv + 128);
    if (REUSE || TWO) {
        for (int t = tid; t < ntiles; t += THREADS) {
            const int below = min((t + stride - 1) / stride, S);          // sample slots in front of tile t
            const bool sampled = t % stride == 0 && t / stride < S;
            const bool kept = sampled && (REUSE || ((t / stride) & 1));  // its distances are kept: not pass B's
            if (!kept)
                tile_list[t - (REUSE ? below : below / 2)] = t;
        }
    }
    auto tile_b = [&](int u, bool uniform) {               // u < nB
        return !(REUSE || TWO) ? u : uniform ? __builtin_amdgcn_readfirstlane(tile_list[u]) : tile_list[u];
    };

    // staging: a round = CS tiles of 32 rows; a wave brings 8 rows, each with ONE global_load_lds_dword: lane l fetches
    // the channel that belongs at position l of the staged row ([32 even | 32 odd]), the 256 bytes land in LDS without
    // passing through registers (a register-staged copy cost 8 VGPRs for the whole round and spilled).  Rows past the
    // end repeat row n - 1: their |x|^2 reads +inf, so their distances are +inf.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int chan = 2 * col + half;
    // first row of the tile in slot `slot` of round r.  what: 0 = pass B's tiles, 1 = pass A's (A1's) sample, 2 = the
    // query tiles, 3 = every tile in order (the fallback scan), 4 = pass A2's sample
    auto tile_row0 = [&](int what, int slot, bool uniform) {
        return what == 1   ? (slot < SA1 ? (TWO ? 2 * slot : slot) * stride * KM_TILE : n)
               : what == 4 ? (slot < SA2 ? (2 * slot + 1) * stride * KM_TILE : n)
               : what == 2 ? (qgroup * QW + slot) * KM_TILE
               : what == 0 ? (slot < nB ? tile_b(slot, uniform) * KM_TILE : n)
                           : slot * KM_TILE;
    };
    float *qstage = qd_all;                                // the query tiles are staged in the (still unused) queue area
    auto stage_rows = [&](int what, int r) {
        const int slot = r * CS + (wave_u >> 2);           // a wave's 8 rows belong to one tile
        const int c0 = tile_row0(what, slot, true);
        float *dst = (what == 2 ? qstage : tiles) + (wave_u >> 2) * TILE_FLOATS;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = (wave_u & 3) * 8 + u;          // row within the tile
            const int g = min(c0 + row, n - 1);
            __builtin_amdgcn_global_load_lds(X + (size_t)g * ld + chan, dst + row * KS_LD, 4, 0, 0);
        }
    };
    auto staged = [&]() {                                  // this wave's rows have landed (and its queue writes); then the barrier
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // |x|^2 of the 128 staged rows, the un-fused sequential sum the oracle defines: two waves, one row per lane, between
    // the round's two barriers (the 33rd step's operand is read after the second one).  A vector instruction takes matrix
    // time on its SIMD (tools/dev/mfma_valu_overlap.hip: the two do not overlap), so the pair of waves alternates between
    // SIMDs 0-1 and 2-3 from round to round, and the squares are packed multiplies (two per instruction, same rounding).
    typedef float float2v __attribute__((ext_vector_type(2)));
    auto norms = [&](int what, int r) {
        if ((wave_u >> 1) == (r & 1)) {
            const int rowu = (wave_u & 1) * 64 + lane;
            const int g = tile_row0(what, r * CS + (rowu >> 5), false) + (rowu & 31);
            const float4v *ev = reinterpret_cast<const float4v *>((what == 2 ? qstage : tiles) + (rowu >> 5) * TILE_FLOATS +
                                                                  (rowu & 31) * KS_LD);
            float acc = 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {                  // channels 8 q .. 8 q + 7 = even[4q..4q+3] interleaved with odd[..]
                if (q == 4)
                    __builtin_amdgcn_sched_barrier(0);     // (two batches of eight 16-byte reads: 32 registers, not 64)
                const float4v e = ev[q], o = ev[8 + q];
                const float2v e0 = {e.x, e.y}, e1 = {e.z, e.w}, o0 = {o.x, o.y}, o1 = {o.z, o.w};
                const float2v pe0 = e0 * e0, pe1 = e1 * e1, po0 = o0 * o0, po1 = o1 * o1;
                const float a0 = pe0.x, a1 = po0.x, a2 = pe0.y, a3 = po0.y, a4_ = pe1.x, a5 = po1.x, a6 = pe1.y, a7 = po1.y;
                acc = acc + a0;
                acc = acc + a1;
                acc = acc + a2;
                acc = acc + a3;
                acc = acc + a4_;
                acc = acc + a5;
                acc = acc + a6;
                acc = acc + a7;
            }
            if (g < n)
                sq[g] = acc;
        }
    };
    const float4v *arow = reinterpret_cast<const float4v *>(tiles + cs * TILE_FLOATS + col * KS_LD + 32 * half);
    const int xoff = half ? col : ntiles * KM_TILE;        // 33rd step, candidate side: sq[c0 + col] (k = 1) or 1.0 (k = 0)
    const int xmul = half;

    f32x16 acc;
    float bq[32];                                         // B operand: -2 x the query's channels of parity `half`
    float bx = 1.0f;
    // one round: operands of this wave's tile -> registers, the 33 MFMA steps, next round's tiles -> LDS.  The round's
    // second barrier ("every wave holds its operands: the buffer is free") sits INSIDE the MFMA chain, after PRE steps:
    // placed before the chain every wave idled while the operands travelled (measured, B = 32: 69.7 us with PRE = 0,
    // 64.7 / 63.2 / 63.0 with PRE = 4 / 16 / 24)
    constexpr int PRE = 24;
    auto round = [&](int what, int r, int rounds, int c0) {
        staged();                                          // this round's tiles are in LDS
        norms(what, r);                                    // (two waves; before their operand reads: registers)
        float4v a4[8];
#pragma unroll
        for (int s = 0; s < 8; ++s)
            a4[s] = arow[s];
#pragma unroll
        for (int e = 0; e < 16; ++e)
            acc[e] = 0.0f;
#pragma unroll
        for (int s = 0; s < PRE; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s >> 2][s & 3], bq[s], acc, 0, 0, 0);
        if (PRE > 0)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                   // every wave holds its operands: the buffer is free
        const float ax = sq[xoff + xmul * c0];
        if (r + 1 < rounds)
            stage_rows(what, r + 1);                       // travels behind the MFMAs
#pragma unroll
        for (int s = PRE; s < 32; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s >> 2][s & 3], bq[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bx, acc, 0, 0, 0);
    };

    // ---------------- pass A: tau ----------------
    const int roundsA = (SA1 + CS - 1) / CS;
    stage_rows(2, 0);                                      // the workgroup's 4 query tiles, staged like candidate tiles
    stage_rows(1, 0);                                      // (the first sample tiles travel with them)
    staged();
    norms(2, 0);
    {
        const float4v *qrow = reinterpret_cast<const float4v *>(qstage + qt * TILE_FLOATS + col * KS_LD + 32 * half);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float4v v = qrow[s];
            bq[4 * s] = -2.0f * v.x;
            bq[4 * s + 1] = -2.0f * v.y;
            bq[4 * s + 2] = -2.0f * v.z;
            bq[4 * s + 3] = -2.0f * v.w;
        }
    }
    __syncthreads();                                       // the query norms are in sq
    bx = half ? 1.0f : sq[qs];
    // the queues (pass A2 of the two-stage bound and pass B fill them)
    const int qq = qt * 32 + col;
    float *qd = qd_all + qq * QPQ;
    unsigned short *qj = qj_all + qq * QPQ;
    // the lane's entries of one tile: ONE slot request for all of them (a request per entry is a dependent LDS round
    // trip per accumulator register that the wave finishing its MFMAs last cannot hide); past the end of a full queue the
    // last slot is overwritten: the count still says "overflowed".
    // The LDS instructions are written out: behind `atomicAdd` and plain stores the compiler waits for the tile loads in
    // flight first (vmcnt(0): they write LDS too, and it cannot tell the regions apart), which serialises the push behind
    // the next round's tiles.  The waves wait for these writes (lgkmcnt) at the next barrier.
    const unsigned qn_at = lds_offset(&qn_all[qq]), qd_at = lds_offset(qd), qj_at = lds_offset(qj);
    float tau = 0.0f;                                      // (read after the next barrier)
    auto push_tile = [&](const f32x16 &d, int jb) {
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            cnt += d[e] <= tau ? 1 : 0;
        if (cnt > 0) {
            int sl;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(sl) : "v"(qn_at), "v"(cnt) : "memory");
            if (sl + cnt <= QPQ) {                         // (nearly always: no clamping, two running addresses)
                unsigned ad = qd_at + 4 * sl, aj = qj_at + 2 * sl;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (d[e] <= tau) {
                        const int j = jb + (e & 3) + 8 * (e >> 2);
                        asm volatile("ds_write_b32 %0, %1\n\tds_write_b16 %2, %3" ::"v"(ad), "v"(d[e]), "v"(aj), "v"(j) : "memory");
                        ad += 4;
                        aj += 2;
                    }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (d[e] <= tau) {
                        const int at = min(sl, QPQ - 1);
                        const int j = jb + (e & 3) + 8 * (e >> 2);
                        asm volatile("ds_write_b32 %0, %1\n\tds_write_b16 %2, %3" ::"v"(qd_at + 4 * at), "v"(d[e]), "v"(qj_at + 2 * at),
                                     "v"(j)
                                     : "memory");
                        ++sl;
                    }
            }
        }
    };
    // the lane's unit minima: its UL smallest (UL = 12 for K = 20: the K-th smallest over the union of shorter lists is still a
    // bound, and looser only if one of a query's eight lanes held more than 12 of the 20 smallest; eight registers and sixteen
    // vector instructions per insert less where this kernel spills)
    constexpr int UL = K > 12 ? 12 : K;
    MinK<UL> um;
    um.init();
    // REUSE: the sampled tiles' distances are kept for the filter: the last round's in `acc`, the round before in the
    // part of the query tile's queue area that the scratch lists below leave free ([16 values][4 waves x 64 lanes])
    float *svl = reinterpret_cast<float *>(reinterpret_cast<char *>(qd_all) + qt * (32 * QPQ * 6) + 2 * CS * (UL + 1) * 32 * 4) +
                 cs * 64 + lane;
    static_assert(!REUSE || 2 * CS * (UL + 1) * 32 * 4 + 16 * CS * 64 * 4 <= 32 * QPQ * 6, "saved distances must fit beside the lists");
    for (int r = 0; r < roundsA; ++r) {
        const int slot = r * CS + cs;
        round(1, r, roundsA, slot < SA1 ? (TWO ? 2 * slot : slot) * stride * KM_TILE : ntiles * KM_TILE - KM_TILE);
        const bool live = slot < SA1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                      // units: the lane's rows 8 g + 4 half + (0..3)
            const float m = fminf(fminf(acc[4 * g], acc[4 * g + 1]), fminf(acc[4 * g + 2], acc[4 * g + 3]));
            um.insert(live ? m : __builtin_inff());
        }
        if constexpr (REUSE) {
            if (r + 1 < roundsA) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    svl[e * CS * 64] = acc[e];
            }
        }
    }
    // K-th smallest unit minimum over the query's 2*CS lists, through the query tile's share of the queue area
    float *md = reinterpret_cast<float *>(reinterpret_cast<char *>(qd_all) + qt * (32 * QPQ * 6));
    // (UL values and a +inf behind them per list: a head that has taken a whole list reads the sentinel)
    static_assert(2 * CS * (UL + 1) * 32 * 4 <= 32 * QPQ * 6 && (32 * QPQ * 6) % 8 == 0,
                  "scratch lists must fit the queue area of one query tile");
    const int list = cs * 2 + half;
    __syncthreads();
    int slot0 = list * (UL + 1) * 32 + col;                // (opaque: keeps the compiler from deriving these addresses
    asm volatile("" : "+v"(slot0));                        //  before the scan loop and spilling them across it)
#pragma unroll
    for (int p = 0; p < UL; ++p)
        md[slot0 + p * 32] = um.d[p];
    md[slot0 + UL * 32] = __builtin_inff();
    const int roundsB = (nB + CS - 1) / CS;
    const int roundsA2 = (SA2 + CS - 1) / CS;
    if (TWO ? roundsA2 > 0 : roundsB > 0)
        stage_rows(TWO ? 4 : 0, 0);                        // the next pass's first tiles travel during the merge below
    __syncthreads();
    // K-th smallest over the query's lists: ONE wave per query tile merges (waves 0, 5, 10, 15: one per SIMD), the bound
    // reaches the other lanes through LDS after the next barrier
    if (cs == qt) {
        float t = __builtin_inff();
        // K steps of "smallest head, advance it" over the 2*CS sorted lists.  Equal heads advance together, which can
        // only make the bound larger (it stays valid).
        int head[2 * CS];
#pragma unroll
        for (int l = 0; l < 2 * CS; ++l)
            head[l] = (l * (UL + 1)) * 32 + col;
#pragma unroll
        for (int p = 0; p < K; ++p) {
            float hv[2 * CS];
#pragma unroll
            for (int l = 0; l < 2 * CS; ++l)
                hv[l] = md[head[l]];                       // (a head stops at its list's sentinel: +inf is never the minimum
                                                           //  unless every list is exhausted, and then it stays the answer)
            float m = hv[0];
#pragma unroll
            for (int l = 1; l < 2 * CS; ++l)
                m = fminf(m, hv[l]);
#pragma unroll
            for (int l = 0; l < 2 * CS; ++l)
                head[l] += hv[l] == m ? 32 : 0;
            t = m;
        }
        if (half == 0)
            tauv[qt * 32 + col] = fminf(t, 3.4028234664e38f);                 // rows past the end (+inf) never pass
    }

    if constexpr (TWO) {
        // ---------------- pass A2: the sample's odd slots, scanned with the first bound ----------------
        // (the scratch lists are consumed and the bounds written before the first round's second barrier; the queues
        //  are only appended to after it)
        for (int r = 0; r < roundsA2; ++r) {
            const int slot = r * CS + cs;
            const int c0 = slot < SA2 ? (2 * slot + 1) * stride * KM_TILE : ntiles * KM_TILE - KM_TILE;
            round(4, r, roundsA2, c0);
            if (r == 0)
                tau = tauv[qq];
            const bool live = slot < SA2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float m = fminf(fminf(acc[4 * g], acc[4 * g + 1]), fminf(acc[4 * g + 2], acc[4 * g + 3]));
                um.insert(live ? m : __builtin_inff());
            }
            if (live)
                push_tile(acc, c0 + 4 * half);
        }
        // the final bound: K-th smallest over the lists of both halves.  The queue area is in use, the tile buffers are
        // not (every wave is past the last round's second barrier, nothing is in flight): the 8 smallest of every lane
        // ([list][8][32 queries] per query tile), merged as above with the heads checked against the end of their lists
        constexpr int LP = 8;
        static_assert(QW * 2 * CS * LP * 32 <= CS * TILE_FLOATS && LP <= UL, "second-stage lists must fit the tile buffers");
        float *m2 = tiles + qt * (2 * CS * LP * 32);
#pragma unroll
        for (int p = 0; p < LP; ++p)
            m2[(list * LP + p) * 32 + col] = um.d[p];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (and pass A2's queue writes, for the clean-up below)
        __syncthreads();
        if (cs == qt) {
            float t = __builtin_inff();
            int hd[2 * CS];
#pragma unroll
            for (int l = 0; l < 2 * CS; ++l)
                hd[l] = 0;
#pragma unroll
            for (int p = 0; p < K; ++p) {
                float hv[2 * CS];
#pragma unroll
                for (int l = 0; l < 2 * CS; ++l) {
                    const float v = m2[(l * LP + min(hd[l], LP - 1)) * 32 + col];
                    hv[l] = hd[l] < LP ? v : __builtin_inff();
                }
                float m = hv[0];
#pragma unroll
                for (int l = 1; l < 2 * CS; ++l)
                    m = fminf(m, hv[l]);
#pragma unroll
                for (int l = 0; l < 2 * CS; ++l)
                    hd[l] += hv[l] == m ? 1 : 0;
                t = m;
            }
            if (half == 0)
                tauv[qt * 32 + col] = fminf(tauv[qt * 32 + col], t);          // (both are bounds; the first one is finite)
        }
        __syncthreads();                                   // the lists are consumed: the tile buffers may fill again
        tau = tauv[qq];
        if (roundsB > 0)
            stage_rows(0, 0);
        // Pass A2 queued by the FIRST bound; what lies above the final one only takes slots (and overflows the queues on
        // clustered features: measured in the config-5 step, layer 4: 1635 us against 1461 without the second stage).
        // Clean-up in place: a query = eight neighbouring lanes of one wave (queries 8 w .. 8 w + 7), every lane reads its
        // share of the entries (i = sub, sub + 8, ...) into registers, keeps those at or below the final bound and writes
        // them back behind the kept entries of the lanes before it (nobody else touches these queues before the next
        // round's first barrier).
        {
            constexpr int PER = (QPQ + 7) / 8;
            const int ql = wave * 8 + (lane >> 3), sub = lane & 7;
            const int have = qn_all[ql];
            const float t2 = tauv[ql];
            float *qdl = qd_all + ql * QPQ;
            unsigned short *qjl = qj_all + ql * QPQ;
            float dv[PER];
            unsigned short jv[PER];
            int mine = 0;
            if (have <= QPQ) {                             // (an overflowed queue stays as it is: the flag will be raised)
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int i = sub + 8 * u;
                    dv[u] = qdl[min(i, QPQ - 1)];
                    jv[u] = qjl[min(i, QPQ - 1)];
                    mine += (i < have && dv[u] <= t2) ? 1 : 0;
                }
            }
            int before = mine;                             // inclusive prefix over the eight lanes
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                const int v = __shfl_up(before, o, 8);
                before += sub >= o ? v : 0;
            }
            const int total = __shfl(before, 7, 8);
            if (have <= QPQ) {
                int at = before - mine;
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int i = sub + 8 * u;
                    if (i < have && dv[u] <= t2) {
                        qdl[at] = dv[u];
                        qjl[at] = jv[u];
                        ++at;
                    }
                }
                if (sub == 0)
                    qn_all[ql] = total;
            }
        }
    }

    // ---------------- pass B: everything at or below tau goes to the query's queue ----------------
    if constexpr (REUSE) {
        f32x16 sv;
        if (roundsA > 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                sv[e] = svl[e * CS * 64];
        }
        __syncthreads();                                   // lists and saved distances are consumed: the queues may fill
        tau = tauv[qq];
        if (roundsA > 1)
            push_tile(sv, cs * stride * KM_TILE + 4 * half);                  // (round 0: slot cs < S always)
        if ((roundsA - 1) * CS + cs < S)
            push_tile(acc, ((roundsA - 1) * CS + cs) * stride * KM_TILE + 4 * half);
    }
    // (neither REUSE nor TWO: the scratch lists are consumed and the bounds written before the first round's second
    //  barrier; the queues are only appended to after it)
    for (int r = 0; r < roundsB; ++r) {
        const int slot = r * CS + cs;
        const int c0 = tile_b(min(slot, nB - 1), true) * KM_TILE;
        round(0, r, roundsB, c0);
        if (!REUSE && !TWO && r == 0)
            tau = tauv[qq];                                // (written before this round's barriers)
        if (slot < nB)
            push_tile(acc, c0 + 4 * half);
    }
    int *flag = reinterpret_cast<int *>(sq + ntiles * KM_TILE + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (the queue writes of push_tile)
    __syncthreads();
    if (qn_all[qq] > QPQ)
        *flag = 1;
    __syncthreads();
    TopKey<K> top;
    top.init();
    if (*flag == 0) {
        // the query's 8 lanes take every 8th entry of its queue (they arrive in any order: keyed insert); the t-th
        // insert into an empty list is t min/max pairs, not K
        const int nq_ = min(qn_all[qq], QPQ);
        int ro = list;
        float nd = qd[min(ro, QPQ - 1)];
        int ni = (int)qj[min(ro, QPQ - 1)];
        auto next_key = [&]() {
            const double key = ro < nq_ ? knn_key(nd, ni) : __builtin_inf();
            ro += 2 * CS;
            const int rn = min(ro, QPQ - 1);
            nd = qd[rn];
            ni = (int)qj[rn];
            return key;
        };
        bool more = __any(ro < nq_);
        static_for<K>([&](auto t) {
            if (more) {
                top.template insert_first<decltype(t)::value>(next_key());
                more = __any(ro < nq_);
            }
        });
        while (more) {
            top.insert(next_key());
            more = __any(ro < nq_);
        }
    } else {
        // a queue overflowed somewhere in this workgroup: the plain scan (every candidate through the sorted insert)
        const int roundsF = (ntiles + CS - 1) / CS;
        stage_rows(3, 0);
        for (int r = 0; r < roundsF; ++r) {
            const int slot = r * CS + cs;
            const int c0 = min(slot, ntiles - 1) * KM_TILE;
            round(3, r, roundsF, c0);
            if (slot < ntiles) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    top.insert(knn_key(acc[e], c0 + 4 * half + (e & 3) + 8 * (e >> 2)));   // (rows past n are +inf)
            }
        }
    }

    // merge the sorted key lists of every query, through the query tile's share of the queue area; the merging
    // lanes of the four query tiles sit in waves 0, 5, 10, 15: one per SIMD
    __syncthreads();
    // lists of K keys and a +inf behind them (a head that has taken a whole list reads the sentinel)
    double *mk = reinterpret_cast<double *>(reinterpret_cast<char *>(qd_all) + qt * (32 * QPQ * 6));
    constexpr int KL = K + 1;
    constexpr bool ONE_STAGE = 2 * CS * KL * 32 * 8 <= 32 * QPQ * 6;
    constexpr int LISTS = ONE_STAGE ? 2 * CS : CS;
    static_assert(LISTS * KL * 32 * 8 <= 32 * QPQ * 6, "key lists must fit a query tile's queues");
    if constexpr (ONE_STAGE) {
        int slot1 = list * KL * 32 + col;
        asm volatile("" : "+v"(slot1));
#pragma unroll
        for (int p = 0; p < K; ++p)
            mk[slot1 + p * 32] = top.key[p];
        mk[slot1 + K * 32] = __builtin_inf();
        __syncthreads();
    } else {
        // first the two lane halves of a wave: the upper half's list goes through LDS into the lower half's
        int slot1 = cs * KL * 32 + col;
        asm volatile("" : "+v"(slot1));
        if (half == 1) {
#pragma unroll
            for (int p = 0; p < K; ++p)
                mk[slot1 + p * 32] = top.key[p];
        }
        __syncthreads();
        if (half == 0) {
            for (int p = 0; p < K; ++p)
                top.insert(mk[slot1 + p * 32]);
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int p = 0; p < K; ++p)
                mk[slot1 + p * 32] = top.key[p];
            mk[slot1 + K * 32] = __builtin_inf();
        }
        __syncthreads();
    }
    if (cs == qt && half == 0 && qvalid) {
        // k steps of "smallest head, advance it": one lane per query, the heads as running addresses (keys are unique -- a
        // candidate is in one list -- except +inf, which only shows when the lists hold fewer than k keys: never, k <= n)
        const double *hp[LISTS];
#pragma unroll
        for (int l = 0; l < LISTS; ++l)
            hp[l] = mk + l * KL * 32 + col;
        int *dst = nn_idx + ((size_t)cloud * n + qi0) * k;
#pragma unroll
        for (int p = 0; p < K; ++p) {
            if (p < k) {
                double hk[LISTS];
#pragma unroll
                for (int l = 0; l < LISTS; ++l)
                    hk[l] = *hp[l];
                double best = hk[0];
#pragma unroll
                for (int l = 1; l < LISTS; ++l)
                    asm("v_min_f64 %0, %1, %2" : "=v"(best) : "v"(best), "v"(hk[l]));
#pragma unroll
                for (int l = 0; l < LISTS; ++l)
                    hp[l] += hk[l] == best ? 32 : 0;
                dst[p] = best < __builtin_inf() ? knn_key_low16(best) : 0;
            }
        }
    }
}

template <int K, int QPQ, bool REUSE, bool TWO>
static hipError_t launch_knn_wide_q(int b, int n, int ld, int k, const float *x, int *nn_idx, hipStream_t s,
                                    const unsigned char *only = nullptr)
{
    const size_t lds = knn_wide_lds_bytes(n);
    static bool raised[64] = {};
    if (hipError_t e = raise_lds_limit(&knn64_wide_kernel<K, QPQ, REUSE, TWO>, raised); e != hipSuccess)
        return e;
    hipLaunchKernelGGL((knn64_wide_kernel<K, QPQ, REUSE, TWO>), dim3(ceil_div(n, KM_TILE * 4), b), dim3(1024), lds, s, n, ld,
                       k, x, nn_idx, only);
    return hipSuccess;
}
template <int K>
static hipError_t launch_knn_wide(int b, int n, int ld, int k, const float *x, int *nn_idx, hipStream_t s,
                                  const unsigned char *only = nullptr)
{
    if constexpr (K <= 10) {
        // pass A of at most two rounds (n <= 1024): its distances are kept and pass B skips the sampled tiles
        if (ceil_div(n, KM_TILE) <= 32 && CLOUDAAE_KNOB("CLOUDAAE_KNN_REUSE", 1) != 0)
            return launch_knn_wide_q<K, 144, true, false>(b, n, ld, k, x, nn_idx, s, only);
    }
    // otherwise the bound in two stages (knob CLOUDAAE_KNN_TWO = 0: one stage, the sample scanned twice)
    if (CLOUDAAE_KNOB("CLOUDAAE_KNN_TWO", 1) != 0)
        return knn_wide_qpq(n) == 144 ? launch_knn_wide_q<K, 144, false, true>(b, n, ld, k, x, nn_idx, s, only)
                                      : launch_knn_wide_q<K, 128, false, true>(b, n, ld, k, x, nn_idx, s, only);
    return knn_wide_qpq(n) == 144 ? launch_knn_wide_q<K, 144, false, false>(b, n, ld, k, x, nn_idx, s, only)
                                  : launch_knn_wide_q<K, 128, false, false>(b, n, ld, k, x, nn_idx, s, only);
}

// ---- C = 3 on the matrix cores: the bound pass + filtered scan of knn64_wide_kernel without its rounds ----------------
// knn3_scan_kernel below spends its time in the vector pipe: ~11 instructions per candidate and lane for the distance and
// the branch-free push, then the sorted inserts of every lane's own stream (a wave pays the maximum over its lanes).  Here
// the distances of a 32 x 32 tile are THREE v_mfma_f32_32x32x2_f32 (channels x, y | z, 0 | the two norms) -- the same fma
// chain from +0 in channel order, so the same bits as knn3_kernel / oracle_knn -- and the selection is the one of
// knn64_wide_kernel: a bound tau on the k-th distance from a sample of the tiles (unit minima), every candidate at or
// below it appended to its query's LDS queue (one slot request per lane and tile), keyed inserts and a list merge at the
// end, a flagged plain rescan when a queue overflows.  The whole cloud (x, y, z, |.|^2) sits in LDS from the start, so
// the passes need no staging, no barriers and no per-round norms: a wave simply walks its quarter of the tiles.
// A workgroup = 16 waves = 4 query tiles x 4 candidate slices, one per CU.
static int knn3_wide_qpq(int n, int k) { return n <= 2048 ? 144 : (k > 10 ? 124 : 112); }
static size_t knn3_wide_lds_bytes(int n, int k)
{
    return 16 * (size_t)ceil_div(n, KM_TILE) * KM_TILE + 6 * 128 * (size_t)knn3_wide_qpq(n, k) + 4 * 128 + 16 + 4 * 128;
}

template <int K, int QPQ>
__global__ __launch_bounds__(1024) void knn3_wide_kernel(int n, int ld, int k, const float *__restrict__ x,
                                                         int *__restrict__ nn_idx)
{
    constexpr int QW = 4, CS = 4, THREADS = 1024;
    extern __shared__ __attribute__((aligned(16))) char k3w_smem[];
    // layout: cloud float4[ntiles * 32] (x, y, z, |.|^2; rows past the end 0, 0, 0, +inf) | queue d[128 queries][QPQ] (fp32) |
    //         queue j, same shape (u16) | queue lengths [128] | overflow flag (+ pad) | tau [128]
    const int ntiles = (n + KM_TILE - 1) / KM_TILE;
    float4v *cand = reinterpret_cast<float4v *>(k3w_smem);
    float *qd_all = reinterpret_cast<float *>(cand + ntiles * KM_TILE);
    unsigned short *qj_all = reinterpret_cast<unsigned short *>(qd_all + QW * 32 * QPQ);
    int *qn_all = reinterpret_cast<int *>(qj_all + QW * 32 * QPQ);
    int *flag = qn_all + QW * 32;
    float *tauv = reinterpret_cast<float *>(flag + 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qt = wave / CS, cs = wave % CS;
    int qgroup, cloud;
    xcd_cloud_tile(qgroup, cloud);
    const float *X = x + (size_t)cloud * n * ld;
    for (int j = tid; j < ntiles * KM_TILE; j += THREADS) {
        const float *row = X + (size_t)min(j, n - 1) * ld;
        const float cx = row[0], cy = row[1], cz = row[2];
        const float a = cx * cx, b = cy * cy, c = cz * cz;
        // the un-fused sequential sum the oracle defines.  (This file is compiled WITHOUT packed-fp32 instructions, csrc/Makefile:
        //  the v_pk_add_f32 with op_sel the compiler made of these three lines is the instruction behind the wrong neighbour
        //  lists of two processes sharing a GPU -- profiles/notes_two_processes_one_gpu.md, round 6; tests/test_isa_rules.py)
        float sq = 0.0f + a;
        sq = sq + b;
        sq = sq + c;
        cand[j] = j < n ? float4v{cx, cy, cz, sq} : float4v{0.0f, 0.0f, 0.0f, __builtin_inff()};
    }
    if (tid < QW * 32)
        qn_all[tid] = 0;
    if (tid == 0)
        *flag = 0;
    __syncthreads();

    const int col = lane & 31, half = lane >> 5;
    const int qi0 = (qgroup * QW + qt) * KM_TILE + col;   // this lane's query
    const bool qvalid = qi0 < n;
    const float4v me = cand[qvalid ? qi0 : 0];
    // B operands: -2 x the query's channel of parity `half` (steps 0 and 1; the fourth channel is a zero), then the norms
    const float bq0 = -2.0f * (half ? me.y : me.x), bq1 = half ? 0.0f : -2.0f * me.z, bx = half ? 1.0f : me.w;
    const int S = min(ntiles, max((ntiles + (K > 10 ? 1 : 3)) / (K > 10 ? 2 : 4), 4));   // the sample: every stride-th tile
    const int stride = ntiles / S;

    f32x16 acc;
    auto tile = [&](int t) {                               // distances of candidate tile t to this lane's query
        const float4v c = cand[t * KM_TILE + col];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? c.y : c.x, bq0, f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, bq1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? c.w : 1.0f, bx, acc, 0, 0, 0);
    };

    // ---------------- pass A: tau ----------------
    MinK<K> um;
    um.init();
    for (int s = cs; s < S; s += CS) {
        tile(s * stride);
#pragma unroll
        for (int g = 0; g < 4; ++g)                        // units: the lane's rows 8 g + 4 half + (0..3)
            um.insert(fminf(fminf(acc[4 * g], acc[4 * g + 1]), fminf(acc[4 * g + 2], acc[4 * g + 3])));
    }
    float *md = reinterpret_cast<float *>(reinterpret_cast<char *>(qd_all) + qt * (32 * QPQ * 6));
    static_assert(2 * CS * K * 32 * 4 <= 32 * QPQ * 6 && (32 * QPQ * 6) % 8 == 0,
                  "scratch lists must fit the queue area of one query tile");
    const int list = cs * 2 + half;
#pragma unroll
    for (int p = 0; p < K; ++p)
        md[(list * K + p) * 32 + col] = um.d[p];
    __syncthreads();
    if (cs == qt) {
        float t = __builtin_inff();
        // K steps of "smallest head, advance it" over the 2*CS sorted lists.  Equal heads advance together, which can
        // only make the bound larger (it stays valid).
        int head[2 * CS];
#pragma unroll
        for (int l = 0; l < 2 * CS; ++l)
            head[l] = (l * K) * 32 + col;
#pragma unroll
        for (int p = 0; p < K; ++p) {
            float hv[2 * CS];
#pragma unroll
            for (int l = 0; l < 2 * CS; ++l)
                hv[l] = md[head[l]];                       // (a head moves at most once per step: never past its list)
            float m = hv[0];
#pragma unroll
            for (int l = 1; l < 2 * CS; ++l)
                m = fminf(m, hv[l]);
#pragma unroll
            for (int l = 0; l < 2 * CS; ++l)
                head[l] += hv[l] == m ? 32 : 0;
            t = m;
        }
        if (half == 0)
            tauv[qt * 32 + col] = fminf(t, 3.4028234664e38f);                 // rows past the end (+inf) never pass
    }
    __syncthreads();                                       // the lists are consumed, the bounds written: the queues may fill

    // ---------------- pass B: everything at or below tau goes to the query's queue ----------------
    const int qq = qt * 32 + col;
    float *qd = qd_all + qq * QPQ;
    unsigned short *qj = qj_all + qq * QPQ;
    const float tau = tauv[qq];
    for (int t = cs; t < ntiles; t += CS) {
        tile(t);
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            cnt += acc[e] <= tau ? 1 : 0;
        if (cnt > 0) {                                     // ONE slot request for the lane's entries of the tile
            int sl = atomicAdd(&qn_all[qq], cnt);
            const int jb = t * KM_TILE + 4 * half;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (acc[e] <= tau) {
                    // past the end of a full queue the last slot is overwritten: the count still says "overflowed"
                    const int at = min(sl, QPQ - 1);
                    qd[at] = acc[e];
                    qj[at] = (unsigned short)(jb + (e & 3) + 8 * (e >> 2));
                    ++sl;
                }
        }
    }
    __syncthreads();
    if (qn_all[qq] > QPQ)
        *flag = 1;
    __syncthreads();
    TopKey<K> top;
    top.init();
    if (*flag == 0) {
        // the query's 8 lanes take every 8th entry of its queue (they arrive in any order: keyed insert); the t-th
        // insert into an empty list is t min/max pairs, not K
        const int nq_ = min(qn_all[qq], QPQ);
        int ro = list;
        float nd = qd[min(ro, QPQ - 1)];
        int ni = (int)qj[min(ro, QPQ - 1)];
        auto next_key = [&]() {
            const double key = ro < nq_ ? knn_key(nd, ni) : __builtin_inf();
            ro += 2 * CS;
            const int rn = min(ro, QPQ - 1);
            nd = qd[rn];
            ni = (int)qj[rn];
            return key;
        };
        bool more = __any(ro < nq_);
        static_for<K>([&](auto t) {
            if (more) {
                top.template insert_first<decltype(t)::value>(next_key());
                more = __any(ro < nq_);
            }
        });
        while (more) {
            top.insert(next_key());
            more = __any(ro < nq_);
        }
    } else {
        // a queue overflowed somewhere in this workgroup: the plain scan (every candidate through the sorted insert)
        for (int t = cs; t < ntiles; t += CS) {
            tile(t);
#pragma unroll
            for (int e = 0; e < 16; ++e)
                top.insert(knn_key(acc[e], t * KM_TILE + 4 * half + (e & 3) + 8 * (e >> 2)));      // (rows past n are +inf)
        }
    }

    // merge the sorted key lists of every query (K keys and a +inf behind them), through the query tile's share of the
    // queue area; the merging lanes of the four query tiles sit in waves 0, 5, 10, 15: one per SIMD
    __syncthreads();
    double *mk = reinterpret_cast<double *>(reinterpret_cast<char *>(qd_all) + qt * (32 * QPQ * 6));
    constexpr int KL = K + 1;
    constexpr bool ONE_STAGE = 2 * CS * KL * 32 * 8 <= 32 * QPQ * 6;
    constexpr int LISTS = ONE_STAGE ? 2 * CS : CS;
    static_assert(LISTS * KL * 32 * 8 <= 32 * QPQ * 6, "key lists must fit a query tile's queues");
    if constexpr (ONE_STAGE) {
#pragma unroll
        for (int p = 0; p < K; ++p)
            mk[(list * KL + p) * 32 + col] = top.key[p];
        mk[(list * KL + K) * 32 + col] = __builtin_inf();
        __syncthreads();
    } else {
        // first the two lane halves of a wave: the upper half's list goes through LDS into the lower half's
        if (half == 1) {
#pragma unroll
            for (int p = 0; p < K; ++p)
                mk[(cs * KL + p) * 32 + col] = top.key[p];
        }
        __syncthreads();
        if (half == 0) {
            for (int p = 0; p < K; ++p)
                top.insert(mk[(cs * KL + p) * 32 + col]);
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int p = 0; p < K; ++p)
                mk[(cs * KL + p) * 32 + col] = top.key[p];
            mk[(cs * KL + K) * 32 + col] = __builtin_inf();
        }
        __syncthreads();
    }
    if (cs == qt && half == 0 && qvalid) {
        const double *hp[LISTS];
#pragma unroll
        for (int l = 0; l < LISTS; ++l)
            hp[l] = mk + l * KL * 32 + col;
        int *dst = nn_idx + ((size_t)cloud * n + qi0) * k;
#pragma unroll
        for (int p = 0; p < K; ++p) {
            if (p < k) {
                double hk[LISTS];
#pragma unroll
                for (int l = 0; l < LISTS; ++l)
                    hk[l] = *hp[l];
                double best = hk[0];
#pragma unroll
                for (int l = 1; l < LISTS; ++l)
                    asm("v_min_f64 %0, %1, %2" : "=v"(best) : "v"(best), "v"(hk[l]));
#pragma unroll
                for (int l = 0; l < LISTS; ++l)
                    hp[l] += hk[l] == best ? 32 : 0;     // (keys are unique: a candidate is in one list)
                dst[p] = best < __builtin_inf() ? knn_key_low16(best) : 0;
            }
        }
    }
}

template <int K, int QPQ>
static hipError_t launch_knn3_wide_q(int b, int n, int ld, int k, const float *x, int *nn_idx, hipStream_t s)
{
    const size_t lds = knn3_wide_lds_bytes(n, K);
    static bool raised[64] = {};
    if (hipError_t e = raise_lds_limit(&knn3_wide_kernel<K, QPQ>, raised); e != hipSuccess)
        return e;
    hipLaunchKernelGGL((knn3_wide_kernel<K, QPQ>), dim3(ceil_div(n, KM_TILE * 4), b), dim3(1024), lds, s, n, ld, k, x, nn_idx);
    return hipSuccess;
}
template <int K>
static hipError_t launch_knn3_wide(int b, int n, int ld, int k, const float *x, int *nn_idx, hipStream_t s)
{
    const int q = knn3_wide_qpq(n, K);
    return q == 144   ? launch_knn3_wide_q<K, 144>(b, n, ld, k, x, nn_idx, s)
           : q == 124 ? launch_knn3_wide_q<K, 124>(b, n, ld, k, x, nn_idx, s)
                      : launch_knn3_wide_q<K, 112>(b, n, ld, k, x, nn_idx, s);
}
static bool knn3_wide_fits(int n, int k) { return k <= 20 && n >= 256 && knn3_wide_lds_bytes(n, k <= 10 ? 10 : 20) <= 160 * 1024; }

// ---- C = 3, second generation: the selection split into filter + queued drain ------------------
// knn3_kernel above runs the sorted insert for every candidate of every lane (a wave executes it
// whenever ANY lane needs it, i.e. always): ~45 instructions per candidate against 8 for the
// distance.  Here (as in knn64_scan_kernel) a candidate is compared with the lane's current k-th
// best and pushed on a per-lane LDS queue (4 instructions); the queue is drained through the
// insert in batches, max-over-lanes pops per batch.  The whole cloud (x, y, z, |.|^2) sits in LDS, a
// wave owns 64 queries and one of CS candidate ranges, the CS lists of a query are merged
// lexicographically by (d, j) at the end.  Same arithmetic and tie rule as knn3_kernel.
constexpr int K3_QCAP = 24, K3_STEP = 8;

template <int K, int CS>
__global__ __launch_bounds__(256) void knn3_scan_kernel(int n, int ld, int k, const float *__restrict__ x,
                                                        int *__restrict__ nn_idx)
{
    constexpr int WAVES = 4, QT = WAVES / CS;             // query tiles (of 64) per workgroup
    extern __shared__ __attribute__((aligned(16))) char k3_smem[];
    // layout: cloud float4[n] | queue d[WAVES][QCAP][64] | queue i[WAVES][QCAP][64]
    float4v *cand = reinterpret_cast<float4v *>(k3_smem);
    float *qd_all = reinterpret_cast<float *>(cand + n);
    int *qi_all = reinterpret_cast<int *>(qd_all + WAVES * K3_QCAP * 64);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qt = wave / CS, cs = wave % CS;
    int qgroup, cloud;
    xcd_cloud_tile(qgroup, cloud);
    const float *X = x + (size_t)cloud * n * ld;
    for (int j = tid; j < n; j += 256) {
        const float *row = X + (size_t)j * ld;
        const float cx = row[0], cy = row[1], cz = row[2];
        const float a = cx * cx, b = cy * cy, c = cz * cz;
        float sq = 0.0f + a;
        sq = sq + b;
        sq = sq + c;
        cand[j] = float4v{cx, cy, cz, sq};
    }
    __syncthreads();
    const int qi0 = (qgroup * QT + qt) * 64 + lane;
    const bool valid = qi0 < n;
    const float4v me = cand[valid ? qi0 : 0];
    const float qx = me.x, qy = me.y, qz = me.z, sqi = me.w;
    float *qd = qd_all + wave * K3_QCAP * 64;
    int *qi = qi_all + wave * K3_QCAP * 64;

    TopK<K> top;
    top.init();
    float thr = __builtin_inff();
    int cnt = 0;
    auto drain = [&]() {
        float nd = cnt > 0 ? qd[lane] : 0.0f;
        int ni = cnt > 0 ? qi[lane] : 0;
        for (int t = 0; __any(t < cnt); ++t) {
            const float cd = nd;
            const int ci = ni;
            if (t + 1 < cnt) {
                nd = qd[(t + 1) * 64 + lane];
                ni = qi[(t + 1) * 64 + lane];
            }
            if (t < cnt)
                top.insert(cd, ci);
        }
        cnt = 0;
        thr = top.d[K - 1];
    };

    const int per = (n + CS - 1) / CS;
    const int j_begin = min(cs * per, n), j_end = min(j_begin + per, n);
    for (int j0 = j_begin; j0 < j_end; j0 += K3_STEP) {
        if (__any(cnt > K3_QCAP - K3_STEP - 1))
            drain();
        float4v c[K3_STEP];
#pragma unroll
        for (int u = 0; u < K3_STEP; ++u)
            c[u] = cand[min(j0 + u, n - 1)];              // broadcast reads
#pragma unroll
        for (int u = 0; u < K3_STEP; ++u) {
            float inner = fmaf(qx, c[u].x, 0.0f);
            inner = fmaf(qy, c[u].y, inner);
            inner = fmaf(qz, c[u].z, inner);
            const float m2 = -2.0f * inner;
            const float t = sqi + m2;
            const float d = t + c[u].w;
            qd[cnt * 64 + lane] = d;                      // branch-free push
            qi[cnt * 64 + lane] = j0 + u;
            cnt += (j0 + u < j_end && d < thr) ? 1 : 0;
        }
    }
    drain();

    // merge the CS lists of every query: [list][p][lane], through the queue area of the tile's first wave
    __syncthreads();
    float *md = qd_all + (qt * CS) * K3_QCAP * 64;
    int *mi = qi_all + (qt * CS) * K3_QCAP * 64;
    static_assert(K <= K3_QCAP, "merge lists must fit the queue area");
#pragma unroll
    for (int p = 0; p < K; ++p) {
        md[(cs * K + p) * 64 + lane] = top.d[p];
        mi[(cs * K + p) * 64 + lane] = top.i[p];
    }
    __syncthreads();
    if (cs == 0 && valid) {
        int head[CS];
#pragma unroll
        for (int l = 0; l < CS; ++l)
            head[l] = 0;
        int *dst = nn_idx + ((size_t)cloud * n + qi0) * k;
        for (int p = 0; p < k; ++p) {
            float bd = __builtin_inff();
            int bi = 0x7fffffff, bl = 0;
#pragma unroll
            for (int l = 0; l < CS; ++l) {
                const int h = head[l];
                const float d = h < K ? md[(l * K + h) * 64 + lane] : __builtin_inff();
                const int i = h < K ? mi[(l * K + h) * 64 + lane] : 0x7fffffff;
                const bool better = d < bd || (d == bd && i < bi);
                bd = better ? d : bd;
                bi = better ? i : bi;
                bl = better ? l : bl;
            }
#pragma unroll
            for (int l = 0; l < CS; ++l)
                head[l] += (l == bl) ? 1 : 0;
            dst[p] = bi == 0x7fffffff ? 0 : bi;
        }
    }
}

template <int K, int CS>
static hipError_t launch_knn3_scan(int b, int n, int ld, int k, const float *x, int *nn_idx, hipStream_t s)
{
    const size_t lds = 16 * (size_t)n + 8 * 4 * K3_QCAP * 64;
    static bool raised[64] = {};
    if (hipError_t e = raise_lds_limit(&knn3_scan_kernel<K, CS>, raised); e != hipSuccess)
        return e;
    hipLaunchKernelGGL((knn3_scan_kernel<K, CS>), dim3(ceil_div(n, 64 * (4 / CS)), b), dim3(256), lds, s, n, ld, k, x,
                       nn_idx);
    return hipSuccess;
}

// ---- C = 64, fourth generation (round 5; NOT the default -- see the end of this comment): the scan on the bf16 matrix pipe -----
// knn64_wide_kernel computes every one of the N x N distances the way the oracle defines them (a k-ordered fp32 fma chain =
// v_mfma_f32_32x32x2_f32, 33 steps of 64 cycles per 32 x 32 tile) and is bound by exactly that: the fp32 matrix rate.  But
// the oracle's arithmetic is needed only for the candidates that can be among the k nearest.  Here:
//   planes  : every point once, minus a centre of its cloud, as two bf16 pieces per channel, h = bf16(y), m = bf16(y - h)
//             (y - h - m is below 2^-18 |y|); |x|^2 in the oracle's order (fp32, exact), |y|^2, and -|y|^2 / 2 as two bf16
//             pieces (knn64_planes_kernel).
//   scores  : s~_ij = y_i.y_j - (|y_i|^2 + |y_j|^2) / 2 = -d_ij / 2 from 13 v_mfma_f32_32x32x16_bf16 per tile (h.h, h.m, m.h
//             over four blocks of 16 channels, and one block that adds the norms): 416 matrix cycles instead of 2112, and
//             |s~ - s| <= E_i for the s the oracle's arithmetic gives (the bound is spelled out in the kernel).
//   pass A  : every tile; per lane the maximum of its rows of the tile goes into a sorted list of K values; the k-th largest
//             over the query's two lanes is tau~ (k DISTINCT candidates score at least that).
//   pass B  : every tile again; a candidate with s~ >= tau~ - 2 E_i goes to its lane's queue in LDS (the index only).  Every
//             candidate the oracle ranks among the k nearest is there: its s is >= the oracle's k-th largest >= tau~ - E_i.
//   select  : the oracle's distance (the fma chain over the 64 channels, (|x_i|^2 + -2 inner) + |x_j|^2) for every queued
//             candidate, sorted keys, a lane pair per query.
//   A queue that overflows flags the query group; a gated launch of knn64_wide_kernel (workgroups leave at once unless
//   flagged) recomputes those: correctness never depends on the margins being small, only on E_i being a bound.
// Bit-exact on every case of tests/test_00_ops_gpu.py (CLOUDAAE_KNN_SPLIT = 2 forces it).  Measured (profiles/
// notes_knn_split_r5.md): post-ReLU Gaussian features, [128, 1024, k = 10] 229 -> 163 us and [32, 4096, k = 20] 1036 -> 542 us with
// a first version of the selection (the oracle's distance only for neighbours the scores cannot order).  INSIDE a training
// step it loses: the features of a freshly initialised encoder sit in a ball a tenth of their own size (|x - centre|^2 =
// 0.01 |x|^2), neighbours' distances are 1e-3 |x|^2 and 5e-5 |x|^2 apart -- the level of the oracle's OWN rounding (its fma chain
// is good to 2e-6 |x|^2 a term), so half of all neighbours have to be settled by the oracle's arithmetic anyway; config 5's
// clouds repeat points (visible points re-drawn to 4096 rows), whose exact ties overflow the per-lane queues.
// B = 128: 241 us against 236; config 5: the queues overflow.  The fp32 kernel stays the default.
constexpr int KSP_T = 2;             // candidate tiles per round
constexpr int KSP_NB = 4;            // ring of round buffers: the tiles of round r + 3 travel while round r computes
constexpr int KSP_TILE_BYTES = KM_TILE * 256;
typedef __bf16 ksp_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ksp_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned ksp_bf16_rne(float v)           // finite v
{
    const unsigned u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

struct KspScratch {                  // one call's scratch (scratch_alloc): offsets in bytes
    size_t flags, rows, ext, sqx, sqc, total;
    int npad, gq;                    // rows per cloud in the planes (whole 256-query workgroups); 128-query groups per cloud
};
static KspScratch ksp_layout(int b, int n)
{
    KspScratch L;
    L.gq = ceil_div(n, 4 * KM_TILE);
    L.npad = ceil_div(n, 256) * 256;
    L.flags = 0;
    L.rows = ((size_t)b * L.gq + 255) / 256 * 256;
    L.ext = L.rows + (size_t)b * L.npad * 256;
    L.sqx = L.ext + (size_t)b * L.npad * 4;
    L.sqc = L.sqx + (size_t)b * L.npad * 4;
    L.total = L.sqc + (size_t)b * L.npad * 4;
    return L;
}

// grid (npad / 32, b), 256 threads: thread (row r = t / 8, unit u = t % 8) splits channels 8u .. 8u + 7 of its row -- of the row
// MINUS a centre of the cloud: distances do not change, and the error of a score is relative to the norms of what was
// split.  Features behind a ReLU are all positive: their spread around the centre is a fraction of their size (at the
// first training steps a twentieth), and neighbours whose distances differ by 1e-4 of |x|^2 -- most of them, there -- are
// told apart by scores on centred rows but not by scores on the rows themselves.  The centre is the mean of 64 rows spread
// over the cloud, summed in one fixed order by every workgroup (ANY centre is valid; it only has to be the same for the whole
// cloud).  The norm in the oracle's order (of the row itself: squares rounded, then summed channel by channel) is a chain
// through the row's eight threads: thread u continues the sum where thread u - 1 stopped (eight steps of eight additions;
// every thread runs all of them, one keeps the result); the norm of the centred row likewise.  Also clears the flag bytes
// of the query groups (a byte per 128 rows).
__global__ __launch_bounds__(256) void knn64_planes_kernel(int n, int ld, int npad, int gq, const float *__restrict__ x,
                                                          unsigned char *__restrict__ rows, unsigned *__restrict__ ext,
                                                          float *__restrict__ sqx, float *__restrict__ sqc,
                                                          unsigned char *__restrict__ flags)
{
    __shared__ float part[4][64];
    const int cloud = blockIdx.y, t = threadIdx.x, lane = t & 63;
    const int row = blockIdx.x * KM_TILE + (t >> 3), u = t & 7;
    const float *X = x + (size_t)cloud * n * ld;
    if (t == 0 && (blockIdx.x & 3) == 0 && (int)(blockIdx.x >> 2) < gq)
        flags[cloud * gq + (blockIdx.x >> 2)] = 0;
    {
        float sum = 0.0f;
        for (int i = 0; i < 16; ++i) {
            const int r = (int)(((long long)((t >> 6) * 16 + i) * n) >> 6);     // sample row (t / 64) * 16 + i of 64
            sum += X[(size_t)r * ld + lane];
        }
        part[t >> 6][lane] = sum;
    }
    __syncthreads();
    float mu[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = 8 * u + e;
        mu[e] = (((part[0][c] + part[1][c]) + part[2][c]) + part[3][c]) * (1.0f / 64.0f);
    }
    unsigned h[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0};
    float v2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, w2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < n) {
        const float4v a = *reinterpret_cast<const float4v *>(X + (size_t)row * ld + 8 * u);
        const float4v c = *reinterpret_cast<const float4v *>(X + (size_t)row * ld + 8 * u + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float w = v[e] - mu[e];
            const unsigned hb = ksp_bf16_rne(w);
            const unsigned mb = ksp_bf16_rne(w - __uint_as_float(hb << 16));         // (the difference is exact)
            h[e >> 1] |= hb << (16 * (e & 1));
            m[e >> 1] |= mb << (16 * (e & 1));
            v2[e] = v[e] * v[e];
            w2[e] = w * w;
        }
    }
    unsigned char *R = rows + ((size_t)cloud * npad + row) * 256;
    *reinterpret_cast<ksp_u4 *>(R + 16 * u) = ksp_u4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<ksp_u4 *>(R + 128 + 16 * u) = ksp_u4{m[0], m[1], m[2], m[3]};
    float sq = 0.0f, sc = 0.0f;
#pragma unroll
    for (int step = 0; step < 8; ++step) {
        float c = sq, d = sc;                               // (step 0: the chains start from +0, like the oracle's)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            c = c + v2[e];
            d = d + w2[e];
        }
        sq = __shfl(c, (lane & ~7) | step, 64);             // what the row's thread `step` computed goes on
        sc = __shfl(d, (lane & ~7) | step, 64);
    }
    if (u == 0) {
        unsigned e = 0xff80u;                               // (-inf, 0): a row past the end never scores
        if (row < n) {
            const float hs = -0.5f * sc;
            const unsigned a0 = ksp_bf16_rne(hs);
            const unsigned a1 = ksp_bf16_rne(hs - __uint_as_float(a0 << 16));
            e = a0 | (a1 << 16);
        }
        ext[(size_t)cloud * npad + row] = e;
        sqx[(size_t)cloud * npad + row] = row < n ? sq : __builtin_inff();
        sqc[(size_t)cloud * npad + row] = row < n ? sc : 0.0f;
    }
}

template <int K>
struct MaxK {                        // the K largest values seen, descending
    float d[K];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int p = 0; p < K; ++p)
            d[p] = -__builtin_inff();
    }
    __device__ __forceinline__ void insert(float nd)
    {
        d[K - 1] = fmaxf(d[K - 1], nd);
#pragma unroll
        for (int p = K - 1; p > 0; --p) {
            const float a = d[p - 1], b = d[p];
            d[p - 1] = fmaxf(a, b);
            d[p] = fminf(a, b);
        }
    }
};

// queue slots per LANE (a query's two lanes keep their own: no counters in LDS, no atomics) + one slot that takes every write
// that is not a hit
constexpr int KSP_QH = 32;
template <int QW>
static size_t ksp_lds_bytes()
{
    return KSP_NB * KSP_T * KSP_TILE_BYTES + KSP_NB * KSP_T * 64 * 4 + 64 * QW * (size_t)(KSP_QH + 2) * 2 + 64 * 4 + 16;
}

// QW waves per workgroup, a 32-query tile each (8 where that still fills the chip: two waves per SIMD, and the cloud passes
// through LDS half as often; every workgroup streams the whole cloud through its own LDS either way)
template <int K, int QW>
__global__ __launch_bounds__(64 * QW) void knn64_split_kernel(int n, int ld, int k, int npad, int gq,
                                                             const float *__restrict__ x,
                                                             const unsigned char *__restrict__ rows,
                                                             const unsigned *__restrict__ ext, const float *__restrict__ sqx,
                                                             const float *__restrict__ sqc, unsigned char *__restrict__ flags,
                                                             int *__restrict__ nn_idx)
{
    constexpr int QH = KSP_QH, T = KSP_T, NB = KSP_NB, NQ = 32 * QW;
    extern __shared__ __attribute__((aligned(16))) char ksp_smem[];
    char *tiles = ksp_smem;                                               // [NB][T][32 rows x 256 bytes], units XOR-swizzled
    unsigned *extb = reinterpret_cast<unsigned *>(tiles + NB * T * KSP_TILE_BYTES);   // [NB][T][64]
    unsigned short *qj = reinterpret_cast<unsigned short *>(extb + NB * T * 64);      // [64 QW lanes][QH + 2] candidate indices
    float *red = reinterpret_cast<float *>(qj + 64 * QW * (QH + 2));      // [64]
    int *flag = reinterpret_cast<int *>(red + 64);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, half = lane >> 5;
    int wg, cloud;
    xcd_cloud_tile(wg, cloud);
    const int ntiles = (n + KM_TILE - 1) / KM_TILE;
    const unsigned char *R = rows + (size_t)cloud * npad * 256;
    const unsigned *EX = ext + (size_t)cloud * npad;
    const float *SQ = sqx + (size_t)cloud * npad;
    const float *SC = sqc + (size_t)cloud * npad;
    if (tid == 0)
        *flag = 0;
    // the largest norms of the cloud, of the rows and of the centred rows (the error bound of a score needs them)
    float sqmax = 0.0f, scmax = 0.0f;
    for (int j = tid; j < n; j += 64 * QW) {
        sqmax = fmaxf(sqmax, SQ[j]);
        scmax = fmaxf(scmax, SC[j]);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        sqmax = fmaxf(sqmax, __shfl_xor(sqmax, off, 64));
        scmax = fmaxf(scmax, __shfl_xor(scmax, off, 64));
    }
    if (lane == 0) {
        red[wave] = sqmax;
        red[32 + wave] = scmax;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < QW; ++w) {
        sqmax = fmaxf(sqmax, red[w]);
        scmax = fmaxf(scmax, red[32 + w]);
    }

    const int qrow = wg * NQ + wave * 32 + col;             // (< npad)
    const bool qvalid = qrow < n;
    // B operands: the query's pieces (this lane's half of every block of 16 channels), and the block that adds the norms
    ksp_u4 bh[4], bm[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        bh[kb] = *reinterpret_cast<const ksp_u4 *>(R + (size_t)qrow * 256 + 16 * (2 * kb + half));
        bm[kb] = *reinterpret_cast<const ksp_u4 *>(R + (size_t)qrow * 256 + 128 + 16 * (2 * kb + half));
    }
    const ksp_u4 bext = half == 0 ? ksp_u4{0x3f803f80u, EX[qrow], 0u, 0u} : ksp_u4{0u, 0u, 0u, 0u};
    const float sq_i = SQ[qrow];
    // How far a score can be from the oracle's -d / 2, with y = x - centre the rows that were split:
    //   the score against y's exact -|y_i - y_j|^2 / 2: 1.6e-5 (|y_i|^2 + |y_j|^2) -- dropped piece products 1.15e-5 |y_i| |y_j|
    //     (m.m', r.y', y.r' with |m| <= 2^-9 |y|, |r| <= 2^-18 |y| per channel), the norms' pieces 2^-19 each and the norms' own
    //     fp32 sums (64 ulp), the matrix pipe's fp32 accumulation (13 instructions, taken as 4 ulp of the running magnitude
    //     each: 3e-6); y is x - centre ROUNDED, off by an ulp of itself: 1.2e-7 more;
    //   the oracle's fp32 distance against the exact one: its fma chain and sums are off by up to 64 ulp of |x_i| |x_j|, the size
    //     of the rows themselves: 2.1e-6 (|x_i|^2 + |x_j|^2).
    const float Es = 1.8e-5f * ((qvalid ? SC[qrow] : 0.0f) + scmax) + 2.1e-6f * ((qvalid ? sq_i : 0.0f) + sqmax);

    // staging: a tile = 32 rows of 256 bytes = 8 wave instructions of 1 KB (global_load_lds, 16 bytes per lane, lane-linear
    // in LDS).  The rows are read back a row per lane, so the 16-byte units of a row are XOR-swizzled with the row number: the
    // lane that fills physical unit p of row r fetches logical unit p ^ (r & 15).
    auto stage = [&](auto tile_of, int slot0, int nslots, int buf) {
#pragma unroll
        for (int mth = 0; mth < T * 8 / QW; ++mth) {
            const int ii = wave + QW * mth, t = ii >> 3, part = ii & 7;
            const int tile = tile_of(min(slot0 + t, nslots - 1));
            const int r = part * 4 + (lane >> 4), pu = lane & 15;
            const unsigned char *src = R + ((size_t)(tile * KM_TILE + r) * 256 + 16 * (pu ^ (r & 15)));
            __builtin_amdgcn_global_load_lds(src, tiles + (buf * T + t) * KSP_TILE_BYTES + part * 1024, 16, 0, 0);
        }
        static_assert(T <= 2 && QW >= 2, "a wave per tile brings the norms' pieces");
        if (wave < T) {
            const int tile = tile_of(min(slot0 + wave, nslots - 1));
            __builtin_amdgcn_global_load_lds(EX + tile * KM_TILE + col, extb + (buf * T + wave) * 64, 4, 0, 0);
        }
    };
    struct Ops {
        ksp_u4 a[9];                                        // the norms' block, then (h, m) of the four blocks of 16 channels
    };
    auto operands = [&](int buf, int t) {
        const char *tb = tiles + (buf * T + t) * KSP_TILE_BYTES + col * 256;
        Ops o;
        o.a[0] = half == 0 ? ksp_u4{extb[(buf * T + t) * 64 + col], 0x3f803f80u, 0u, 0u} : ksp_u4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            o.a[1 + 2 * kb] = *reinterpret_cast<const ksp_u4 *>(tb + 16 * ((2 * kb + half) ^ (col & 15)));
            o.a[2 + 2 * kb] = *reinterpret_cast<const ksp_u4 *>(tb + 16 * ((8 + 2 * kb + half) ^ (col & 15)));
        }
        return o;
    };
    auto mm = [](const ksp_u4 &a, const ksp_u4 &b, const f32x16 &c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ksp_bf16x8, a), __builtin_bit_cast(ksp_bf16x8, b), c, 0, 0, 0);
    };
    // matrix instruction `step` (0 .. 12) of a tile: the norms, then h.h, h.m, m.h of every block
    auto mstep = [&](const Ops &o, auto step, const f32x16 &c) {
        constexpr int st = decltype(step)::value;
        if constexpr (st == 0)
            return mm(o.a[0], bext, c);
        else {
            constexpr int kb = (st - 1) / 3, w = (st - 1) % 3;
            return w == 0 ? mm(o.a[1 + 2 * kb], bh[kb], c) : w == 1 ? mm(o.a[1 + 2 * kb], bm[kb], c) : mm(o.a[2 + 2 * kb], bh[kb], c);
        }
    };
    // A pass: `nslots` tiles in rounds of T through a ring of NB round buffers.  One barrier per round: behind it the round's
    // tiles have landed (every wave waited for its own part) and nobody reads the buffer of the round before any more -- the
    // tiles of round r + NB - 1 travel into it; the loads of the NB - 2 rounds in between stay in flight (loads are issued for
    // rounds past the end as well -- the last tile again -- so the count a wave may leave outstanding never changes).
    // A wave alone on its SIMD issues in order: an accumulating matrix instruction waits ~32 cycles for the one before it,
    // and nothing else of the wave issues meanwhile.  So the visitor's work on tile t - 1 is cut into 13 pieces that sit
    // BETWEEN the 13 matrix instructions of tile t in program order (pinned by scheduling barriers).
    constexpr int LPR = T * 8 / QW;                         // a wave's loads per round (+ 1 for the waves that bring the norms)
    f32x16 zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e)
        zero16[e] = 0.0f;
    asm volatile("" : "+v"(zero16));                        // (opaque: not rebuilt in front of every tile)
    auto pass = [&](auto tile_of, int nslots, auto piece) {
        const int rounds = (nslots + T - 1) / T;
#pragma unroll
        for (int r = 0; r < NB - 1; ++r)
            stage(tile_of, r * T, nslots, r);
        f32x16 cur = zero16;
        int cur_tile = 0;
        bool cur_valid = false;
        for (int r = 0; r < rounds; ++r) {
            if (wave < T)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * (LPR + 1)) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 2) * LPR) : "memory");
            __syncthreads();
            stage(tile_of, (r + NB - 1) * T, nslots, (r + NB - 1) % NB);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const Ops o = operands(r % NB, t);
                f32x16 nxt = zero16;                      // (a register set that stays zero: the first instruction's addend)
                static_for<13>([&](auto st) {
                    nxt = mstep(o, st, nxt);
                    piece(cur, cur_tile, cur_valid, st);
                    __builtin_amdgcn_sched_barrier(0);
                });
                cur = nxt;
                cur_tile = tile_of(min(r * T + t, nslots - 1));
                cur_valid = r * T + t < nslots;
            }
        }
        static_for<13>([&](auto st) { piece(cur, cur_tile, cur_valid, st); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                    // (the buffers are free for the next pass)
    };

    // ---------------- pass A: tau~ ----------------
    // every tile once: per lane the maximum of its sixteen rows of the tile (of two blocks of eight, of four units of four,
    // where the cloud has few tiles) goes into a sorted list of K; the k-th largest over the query's two lanes is reached by
    // k DISTINCT candidates, and with 2 ntiles blocks to choose from it sits just below the k-th best score (about 1.2 k
    // candidates pass it).  The pass costs a third of pass B per tile (no queue writes).
    MaxK<K> best;
    best.init();
    const int units = 2 * ntiles >= 4 * K ? 1 : (4 * ntiles >= 4 * K ? 2 : 4);
    float um[4];
    pass([](int s) { return s; }, ntiles, [&](const f32x16 &acc, int, bool valid, auto step) {
        constexpr int st = decltype(step)::value;
        if constexpr (st < 4) {
            um[st] = fmaxf(fmaxf(acc[4 * st], acc[4 * st + 1]), fmaxf(acc[4 * st + 2], acc[4 * st + 3])) +
                     (valid ? 0.0f : -__builtin_inff());
        } else if constexpr (st == 4) {
            if (units <= 2) {
                um[0] = fmaxf(um[0], um[1]);
                um[2] = fmaxf(um[2], um[3]);
            }
            if (units == 1)
                um[0] = fmaxf(um[0], um[2]);
        } else if constexpr (st == 5) {
            best.insert(um[0]);
        } else if constexpr (st == 9) {
            if (units >= 2)
                best.insert(um[2]);
        } else if constexpr (st == 7 || st == 11) {
            if (units == 4)
                best.insert(um[st == 7 ? 1 : 3]);
        }
    });
    {
        float other[K];
#pragma unroll
        for (int p = 0; p < K; ++p)
            other[p] = __shfl_xor(best.d[p], 32, 64);
#pragma unroll
        for (int p = 0; p < K; ++p)
            best.insert(other[p]);
    }
    float tau = best.d[K - 1];
#pragma unroll
    for (int p = 0; p < K; ++p)
        tau = p == k - 1 ? best.d[p] : tau;
    const float thr = qvalid ? tau - 2.0f * Es : __builtin_inff();          // (a row past the end asks for nothing)

    // ---------------- pass B: the candidates at or above the bound ----------------
    // branch free: every candidate's index is written -- a hit to the lane's next queue slot, anything else to the slot behind
    // the queue
    unsigned short *myqj = qj + (wave * 64 + lane) * (QH + 2);
    int cnt = 0;                                            // hits so far (the queue keeps the first QH)
    pass([](int s) { return s; }, ntiles, [&](const f32x16 &acc, int tile, bool valid, auto step) {
        constexpr int st = decltype(step)::value;
        const float th = valid ? thr : __builtin_inff();
        auto one = [&](auto ee) {
            constexpr int e = decltype(ee)::value;
            const bool hit = acc[e] >= th;
            const int slot = hit ? min(cnt, QH) : QH;
            myqj[slot] = (unsigned short)(tile * KM_TILE + 4 * half + (e & 3) + 8 * (e >> 2));
            cnt += hit ? 1 : 0;
        };
        one(std::integral_constant<int, st>{});
        if constexpr (st >= 10)
            one(std::integral_constant<int, st + 3>{});
    });
    const int cnt_pair = cnt + __shfl_xor(cnt, 32, 64);
    if (qvalid && (cnt > QH || cnt_pair < k))               // (fewer than k: only NaN scores do that)
        *flag = 1;

    // ---------------- select: the oracle's distance for every queued candidate, a lane pair per query ----------------
    // (Round 5 first ordered the queue by score and gave only neighbours closer than the two evaluations can differ the oracle's
    //  distance.  On features as a training step has them -- |x - centre|^2 a hundredth of |x|^2, neighbours' distances 1e-3 |x|^2
    //  and 5e-5 |x|^2 apart -- half of all neighbours are that close: the ORACLE's own fma chain is only good to 2e-6 |x|^2 a
    //  term.  So every candidate of the queue, about 1.2 k per query, gets the fma chain; the bound keeps the queue that short
    //  whatever the data.)
    TopKey<K> top;
    top.init();
    {
        const float *X = x + (size_t)cloud * n * ld;
        const float4v *xi = reinterpret_cast<const float4v *>(X + (size_t)(qvalid ? qrow : 0) * ld);
        const int L = min(cnt, QH);
        int e = 0;
        while (__any(e < L)) {
            const int j = e < L ? (int)myqj[min(e, QH - 1)] : (qvalid ? qrow : 0);
            const float4v *xj = reinterpret_cast<const float4v *>(X + (size_t)j * ld);
            float inner = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < 16; c4 += 4) {
                float4v a[4], b[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = xi[c4 + u];
                    b[u] = xj[c4 + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    inner = __builtin_fmaf(a[u].x, b[u].x, inner);
                    inner = __builtin_fmaf(a[u].y, b[u].y, inner);
                    inner = __builtin_fmaf(a[u].z, b[u].z, inner);
                    inner = __builtin_fmaf(a[u].w, b[u].w, inner);
                }
            }
            const float m2 = -2.0f * inner;
            const float tt = sq_i + m2;
            const float d = tt + SQ[j];
            top.insert(e < L ? knn_key(d, j) : __builtin_inf());
            e += 1;
        }
        double other[K];
#pragma unroll
        for (int p = 0; p < K; ++p)
            other[p] = __shfl_xor(top.key[p], 32, 64);
#pragma unroll
        for (int p = 0; p < K; ++p)
            top.insert(other[p]);
    }
    __syncthreads();
    if (*flag != 0) {
        if (tid < (NQ + 127) / 128 && (wg * NQ) / 128 + tid < gq)
            flags[cloud * gq + (wg * NQ) / 128 + tid] = 1;
        return;
    }
    if (qvalid && half == 0) {
        int *dst = nn_idx + ((size_t)cloud * n + qrow) * k;
#pragma unroll
        for (int p = 0; p < K; ++p)
            if (p < k)
                dst[p] = knn_key_low16(top.key[p]);
    }
}

template <int K>
static hipError_t launch_knn_split(int b, int n, int ld, int k, const float *x, int *nn_idx, hipStream_t s)
{
    const KspScratch L = ksp_layout(b, n);
    void *scratch = nullptr;
    if (hipError_t e = scratch_alloc(&scratch, L.total, s); e != hipSuccess)
        return e;
    unsigned char *base = (unsigned char *)scratch;
    hipLaunchKernelGGL(knn64_planes_kernel, dim3(L.npad / KM_TILE, b), dim3(256), 0, s, n, ld, L.npad, L.gq, x, base + L.rows,
                       (unsigned *)(base + L.ext), (float *)(base + L.sqx), (float *)(base + L.sqc), base + L.flags);
    // 256-query workgroups (two waves per SIMD, the cloud streamed through LDS half as often) where that still fills the
    // chip; 128-query ones otherwise (knob CLOUDAAE_KNN_SPLIT_QW)
    const long long groups = (long long)(L.npad / 128) * b;
    const int qw = CLOUDAAE_KNOB_SET("CLOUDAAE_KNN_SPLIT_QW") ? CLOUDAAE_KNOB("CLOUDAAE_KNN_SPLIT_QW", 4) : (groups >= 1024 ? 8 : 4);
    if (qw == 8) {
        static bool raised[64] = {};
        auto kern = &knn64_split_kernel<K, 8>;
        const size_t lds = ksp_lds_bytes<8>();
        if (hipError_t e = raise_lds_limit(kern, raised); e != hipSuccess)
            return e;
        hipLaunchKernelGGL((knn64_split_kernel<K, 8>), dim3(ceil_div(L.npad, 256), b), dim3(512), lds, s, n, ld, k, L.npad,
                           L.gq, x, base + L.rows, (const unsigned *)(base + L.ext), (const float *)(base + L.sqx),
                           (const float *)(base + L.sqc), base + L.flags, nn_idx);
    } else {
        static bool raised[64] = {};
        auto kern = &knn64_split_kernel<K, 4>;
        const size_t lds = ksp_lds_bytes<4>();
        if (hipError_t e = raise_lds_limit(kern, raised); e != hipSuccess)
            return e;
        hipLaunchKernelGGL((knn64_split_kernel<K, 4>), dim3(L.npad / 128, b), dim3(256), lds, s, n, ld, k, L.npad,
                           L.gq, x, base + L.rows, (const unsigned *)(base + L.ext), (const float *)(base + L.sqx),
                           (const float *)(base + L.sqc), base + L.flags, nn_idx);
    }
    // the flagged query groups once more, the oracle's arithmetic throughout (a byte per workgroup of THAT grid)
    // (development knob CLOUDAAE_KNN_SPLIT_FIXUP = 0 leaves them unwritten: tools/dev/chk_knn_split.py counts them)
    if (CLOUDAAE_KNOB("CLOUDAAE_KNN_SPLIT_FIXUP", 1) != 0)
        if (hipError_t e = launch_knn_wide<K>(b, n, ld, k, x, nn_idx, s, base + L.flags); e != hipSuccess)
            return e;
    return hipFreeAsync(scratch, s);
}


// Which C = 64 kernel for `tiles` 32-query tiles (measured, B x N = 1024 points, k = 10, us):
//   tiles      knn64_mfma   scan, 1 wave/tile   scan, 2 waves/tile   wide (bound pass, 16 waves; round 3 -> end of round 4)
//    256 (B=8)      55            152                100                  73 -> 58
//    512 (B=16)     66            152                101                  74 -> 58
//    640 (B=20)     85                                                    75 -> 59
//   1024 (B=32)    138            134                102                  76 -> 58
//   4096 (B=128)                  414                397                 289 -> 228
//   8192 (B=256)   780            631                788                 571 -> 455
// Return value / knob CLOUDAAE_KNN_SCAN (forces a choice; the tests cover all of them): 0 = knn64_mfma_kernel,
// 1 / 2 = knn64_scan_kernel with one / two waves per query tile, 5 = knn64_wide_kernel (where it applies: see
// knn_wide_fits; otherwise 5 means 1).  (3 / 4 were the 8-wave bound kernel of round 2, superseded by the wide one.)
static bool knn_wide_fits(int n, int k)
{
    return k <= 20 && n >= 256 && knn_wide_lds_bytes(n) <= 158 * 1024;
}

static int knn_scan_waves(long long tiles, int n, int k)
{
    if (CLOUDAAE_KNOB_SET("CLOUDAAE_KNN_SCAN"))
        return CLOUDAAE_KNOB("CLOUDAAE_KNN_SCAN", 0);
    // (re-measured at the end of round 4, k = 10, wide / first generation: 256 tiles 57.7 / 59.3 us at n = 1024, 22.5 / 23.6 at
    //  n = 256, 35.4 / 35.1 at n = 512; 512 tiles 58.2 / 70.1, 22.7 / 27.6, 35.7 / 42.9, 114.8 / 117.4 at n = 2048; 128 tiles
    //  57.5 / 53.5)
    if (tiles >= 256 && knn_wide_fits(n, k))
        return 5;
    return tiles >= 4096 ? 1 : tiles >= 1024 ? 2 : 0;
}

template <int K>
static hipError_t launch_knn(int b, int n, int c, int ld, int k, const float *x, int *nn_idx, hipStream_t s)
{
    const bool first_gen = CLOUDAAE_KNOB_SET("CLOUDAAE_KNN_SCAN") && CLOUDAAE_KNOB("CLOUDAAE_KNN_SCAN", 0) == 0;
    const bool vec = ld % 4 == 0 && ((uintptr_t)x & 15) == 0;
    if (c == 3 && K <= 20 && n <= 6144 && !first_gen) {
        if constexpr (K <= 20) {
            // the matrix-core form wherever it fits (n >= 256, the cloud and the queues in LDS); knob CLOUDAAE_KNN3_WIDE = 0
            // / 1 forces the choice (the tests cover both).  Measured, k = 10, continuous coordinates, wide / scan: n = 1024:
            // B = 1 23 / 39 us, B = 8 24 / 40, B = 32 24 / 47, B = 128 91 / 104, B = 256 180 / 201; n = 256: B = 16 ... 64
            // 12 / 21; n = 512: 17 / 29; [2, 4096] 61 / 88; [32, 4096, k = 20] 354 / 639.  (Clouds made of a few distinct
            // points overflow the queues and pay the rescan: 48 us at B = 32 -- the scan kernel's time.)
            const bool wide = CLOUDAAE_KNOB_SET("CLOUDAAE_KNN3_WIDE") ? CLOUDAAE_KNOB("CLOUDAAE_KNN3_WIDE", 0) != 0 : true;
            if (wide && knn3_wide_fits(n, K))
                return launch_knn3_wide<K>(b, n, ld, k, x, nn_idx, s);
            // candidate ranges per query tile: as few as still give every SIMD two waves (fewer ranges = fewer lists to
            // fill: measured, n = 1024, k = 10, ranges 4 / 2 / 1: B = 32 47 / 55 / - us, B = 64 89 / 67 / - us,
            // B = 128 170 / 127 / 105 us, B = 256 - / 243 / 202 us)
            if ((long long)ceil_div(n, 64) * b >= 2048)
                return launch_knn3_scan<K, 1>(b, n, ld, k, x, nn_idx, s);
            if ((long long)ceil_div(n, 64) * b * 2 >= 2048)
                return launch_knn3_scan<K, 2>(b, n, ld, k, x, nn_idx, s);
            return launch_knn3_scan<K, 4>(b, n, ld, k, x, nn_idx, s);
        }
    } else if (c == 3) {
        hipLaunchKernelGGL(knn3_kernel<K>, dim3(ceil_div(n, 64), b), dim3(KNN_THREADS), 0, s, n, ld, k, x, nn_idx);
    } else if (c == 64 && vec && K <= 20 && n <= 16384) {
        if constexpr (K <= 20) {
            const long long tiles = (long long)ceil_div(n, KM_TILE) * b;
            const int mode = knn_scan_waves(tiles, n, K);
            if (mode == 5 && knn_wide_fits(n, K)) {
                // development knob CLOUDAAE_KNN_SPLIT: 2 = the scan on the bf16 matrix pipe (knn64_split_kernel), 1 = the same where
                // 256-query workgroups fill the chip, 0 (default) = the fp32 matrix pipe throughout: see knn64_split_kernel
                const int split = CLOUDAAE_KNOB("CLOUDAAE_KNN_SPLIT", 0);
                if (n <= 65536 && k >= 2 && (split == 2 || (split == 1 && (long long)ceil_div(n, 128) * b >= 1024)))
                    return launch_knn_split<K>(b, n, ld, k, x, nn_idx, s);
                return launch_knn_wide<K>(b, n, ld, k, x, nn_idx, s);
            } else if (mode == 2) {
                return launch_knn_scan<K, 4, 2>(b, n, ld, k, x, nn_idx, s);
            } else if (mode > 0) {
                return launch_knn_scan<K, 4, 1>(b, n, ld, k, x, nn_idx, s);
            }
            hipLaunchKernelGGL(knn64_mfma_kernel<K>, dim3(ceil_div(n, KM_TILE), b), dim3(KNN_THREADS), 0, s, n, ld, k, x,
                               nn_idx);
        }
    } else {
        hipLaunchKernelGGL(knn_generic_kernel<K>, dim3(ceil_div(n, KNN_THREADS), b), dim3(KNN_THREADS), 0, s, n, c, ld, k, x,
                           nn_idx);
    }
    return hipSuccess;
}

} // namespace cloudaae

using namespace cloudaae;

CLOUDAAE_API int cloudaae_knn(int b, int n, int c, int ld, int k, const float *x, int *nn_idx,
                              cloudaae_stream_t stream)
{
    const char *name = "cloudaae_knn";
    CLOUDAAE_REQUIRE(b >= 0 && n >= 0 && c > 0 && ld >= c, name, "bad size");
    CLOUDAAE_REQUIRE(k >= 1 && k <= 32, name, "k must be in [1,32]");
    CLOUDAAE_REQUIRE(b <= 65535, name, "batch > 65535");
    if (b == 0 || n == 0)
        return 0;
    CLOUDAAE_REQUIRE(k <= n, name, "k > number of points (tf.nn.top_k would reject it)");
    hipStream_t s = (hipStream_t)stream;
    // (a failed launch preparation -- e.g. the LDS limit of a kernel cannot be raised on this device -- is an
    //  error of the call, never a silently skipped kernel)
    const hipError_t e = k <= 10 ? launch_knn<10>(b, n, c, ld, k, x, nn_idx, s)
                       : k <= 20 ? launch_knn<20>(b, n, c, ld, k, x, nn_idx, s)
                                 : launch_knn<32>(b, n, c, ld, k, x, nn_idx, s);
    CLOUDAAE_CHECK_HIP(e, name);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}
