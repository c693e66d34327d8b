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
// Candidate ranges of a query are scanned by different waves; their sorted lists are merged
// lexicographically by (distance, index), which preserves the tie rule.
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
// chain of the oracle, at the matrix-pipe rate and with almost no LDS traffic.  In the
// accumulator layout a lane holds query column (lane & 31) and 16 candidate rows, so lanes l
// and l+32 keep separate sorted k-lists for the same query over disjoint candidates; the
// lists of a query are merged lexicographically by (distance, index) at the end, which is
// exactly "ascending distance, ties -> lower index".
// (Round 6 retired the first generation of this file -- knn3_kernel and knn64_mfma_kernel, a sorted insert per candidate --
//  and round 5's knn64_split_kernel, the scan on the bf16 matrix pipe that lost inside a training step
//  (profiles/notes_knn_split_r5.md; the kernel is kept as profiles/r06_knn_retired_kernels.diff).  Five kernels remain:
//  knn3_wide / knn3_scan (xyz), knn64_wide / knn64_scan (64 channels), knn_generic (anything else).)
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KM_TILE = 32;

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
template <int K, int QPQ, bool REUSE, bool TWO, bool HINT = false>
__global__ __launch_bounds__(1024) void knn64_wide_kernel(int n, int ld, int k, const float *__restrict__ x,
                                                          int *__restrict__ nn_idx, const float *__restrict__ hint_tau = nullptr)
{
    // HINT (round 6): the bound comes with the call -- hint_tau[cloud][query] = the largest distance from the query to k
    // distinct points somebody already suspects of being near it (the previous layer's neighbours:
    // knn64_hint_bound_kernel) -- so there is no pass A: every tile goes through pass B's filter once, 1.0 x the products.
    static_assert(!HINT || (!REUSE && !TWO), "the hinted form has no sample");
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
    const int S = HINT ? 0 : min(ntiles, max((ntiles + (K > 10 ? 1 : 3)) / (K > 10 ? 2 : 4), 4));
    const int stride = HINT ? 1 : ntiles / S;
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
    stage_rows(HINT ? 0 : 1, 0);                           // (the first sample tiles -- HINT: pass B's first tiles -- travel with them)
    if (HINT && tid < QW * 32) {                           // the bounds that came with the call (rows past the end: nothing passes)
        const int q = qgroup * QW * KM_TILE + tid;
        tauv[tid] = q < n ? fminf(hint_tau[(size_t)cloud * n + q], 3.4028234664e38f) : -1.0f;
    }
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
    const int list = cs * 2 + half;
    const int roundsB = (nB + CS - 1) / CS;
    const int roundsA2 = (SA2 + CS - 1) / CS;
    if constexpr (!HINT) {
        // K-th smallest unit minimum over the query's 2*CS lists, through the query tile's share of the queue area
        float *md = reinterpret_cast<float *>(reinterpret_cast<char *>(qd_all) + qt * (32 * QPQ * 6));
        // (UL values and a +inf behind them per list: a head that has taken a whole list reads the sentinel)
        static_assert(2 * CS * (UL + 1) * 32 * 4 <= 32 * QPQ * 6 && (32 * QPQ * 6) % 8 == 0,
                      "scratch lists must fit the queue area of one query tile");
        __syncthreads();
        int slot0 = list * (UL + 1) * 32 + col;                // (opaque: keeps the compiler from deriving these addresses
        asm volatile("" : "+v"(slot0));                        //  before the scan loop and spilling them across it)
    #pragma unroll
        for (int p = 0; p < UL; ++p)
            md[slot0 + p * 32] = um.d[p];
        md[slot0 + UL * 32] = __builtin_inff();
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
static hipError_t launch_knn_wide_q(int b, int n, int ld, int k, const float *x, int *nn_idx, hipStream_t s)
{
    const size_t lds = knn_wide_lds_bytes(n);
    static bool raised[64] = {};
    if (hipError_t e = raise_lds_limit(&knn64_wide_kernel<K, QPQ, REUSE, TWO>, raised); e != hipSuccess)
        return e;
    hipLaunchKernelGGL((knn64_wide_kernel<K, QPQ, REUSE, TWO>), dim3(ceil_div(n, KM_TILE * 4), b), dim3(1024), lds, s, n, ld,
                       k, x, nn_idx);
    return hipSuccess;
}
template <int K>
static hipError_t launch_knn_wide(int b, int n, int ld, int k, const float *x, int *nn_idx, hipStream_t s)
{
    if constexpr (K <= 10) {
        // pass A of at most two rounds (n <= 1024): its distances are kept and pass B skips the sampled tiles
        if (ceil_div(n, KM_TILE) <= 32)
            return launch_knn_wide_q<K, 144, true, false>(b, n, ld, k, x, nn_idx, s);
    }
    // otherwise the bound in two stages (one stage, the sample scanned twice, was round 4's: 1.5 x instead of 1.25 x the products)
    return knn_wide_qpq(n) == 144 ? launch_knn_wide_q<K, 144, false, true>(b, n, ld, k, x, nn_idx, s)
                                  : launch_knn_wide_q<K, 128, false, true>(b, n, ld, k, x, nn_idx, s);
}

// ---- the bound from a hint (round 6) --------------------------------------------------------------------------------------
// tau[cloud][q] = max over the k hinted points j of D(q, j), D in the arithmetic of the kernels above (the channel-ordered fma
// chain of -2 x_q . x_j from +0, then + |x_q|^2, then + |x_j|^2; norms = un-fused sequential sums of squares).  The hinted
// points are k DISTINCT indices (somebody's neighbour list), so at least k candidates lie at or below tau: it bounds the
// k-th distance, and the filtered scan that takes it (knn64_wide_kernel<..., HINT>) returns exactly what every other kernel
// returns -- a bad hint only costs queue slots (and, when a queue overflows, the flagged rescan).  Half a wave per query,
// a lane per hinted point (the lanes past k take the query itself: distance ~0), the row gathers hit L2 (a cloud's features
// are 256 bytes x n: resident).
__global__ __launch_bounds__(256) void knn64_hint_bound_kernel(int n, int ld, int k, const float *__restrict__ x,
                                                               const int *__restrict__ hint, float *__restrict__ tau)
{
    const int cloud = blockIdx.y, q = blockIdx.x * 8 + ((int)threadIdx.x >> 5), j = threadIdx.x & 31;
    if (q >= n)
        return;                                            // (whole half-waves leave together)
    const float *X = x + (size_t)cloud * n * ld;
    int c = j < k ? hint[((size_t)cloud * n + q) * k + j] : q;
    c = min(max(c, 0), n - 1);
    const float4v *qr = reinterpret_cast<const float4v *>(X + (size_t)q * ld);
    const float4v *cr = reinterpret_cast<const float4v *>(X + (size_t)c * ld);
    float acc = 0.0f, sqq = 0.0f, sqc = 0.0f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const float4v a = qr[u], b = cr[u];
        acc = fmaf(b.x, -2.0f * a.x, acc);
        acc = fmaf(b.y, -2.0f * a.y, acc);
        acc = fmaf(b.z, -2.0f * a.z, acc);
        acc = fmaf(b.w, -2.0f * a.w, acc);
        const float a0 = a.x * a.x, a1 = a.y * a.y, a2 = a.z * a.z, a3 = a.w * a.w;
        const float b0 = b.x * b.x, b1 = b.y * b.y, b2 = b.z * b.z, b3 = b.w * b.w;
        sqq = sqq + a0;
        sqq = sqq + a1;
        sqq = sqq + a2;
        sqq = sqq + a3;
        sqc = sqc + b0;
        sqc = sqc + b1;
        sqc = sqc + b2;
        sqc = sqc + b3;
    }
    float d = (acc + sqq) + sqc;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
        d = fmaxf(d, __shfl_xor(d, o, 32));
    if (j == 0)
        tau[(size_t)cloud * n + q] = d;
}

template <int K>
static hipError_t launch_knn_wide_hinted(int b, int n, int ld, int k, const float *x, const int *hint, float *tau, int *nn_idx,
                                         hipStream_t s)
{
    hipLaunchKernelGGL(knn64_hint_bound_kernel, dim3(ceil_div(n, 8), b), dim3(256), 0, s, n, ld, k, x, hint, tau);
    const size_t lds = knn_wide_lds_bytes(n);
    static bool raised[64] = {};
    if (knn_wide_qpq(n) == 144) {
        if (hipError_t e = raise_lds_limit(&knn64_wide_kernel<K, 144, false, false, true>, raised); e != hipSuccess)
            return e;
        hipLaunchKernelGGL((knn64_wide_kernel<K, 144, false, false, true>), dim3(ceil_div(n, KM_TILE * 4), b), dim3(1024), lds, s,
                           n, ld, k, x, nn_idx, tau);
    } else {
        static bool raised128[64] = {};
        if (hipError_t e = raise_lds_limit(&knn64_wide_kernel<K, 128, false, false, true>, raised128); e != hipSuccess)
            return e;
        hipLaunchKernelGGL((knn64_wide_kernel<K, 128, false, false, true>), dim3(ceil_div(n, KM_TILE * 4), b), dim3(1024), lds, s,
                           n, ld, k, x, nn_idx, tau);
    }
    return hipSuccess;
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

// Which C = 64 kernel for `tiles` 32-query tiles (measured, B x N = 1024 points, k = 10, us; knn64_mfma = the retired first
// generation):
//   tiles      knn64_mfma   scan, 1 wave/tile   scan, 2 waves/tile   wide (bound pass, 16 waves; round 3 -> end of round 4)
//    128 (B=4)      53.5                                                  57.5
//    256 (B=8)      55            152                100                  73 -> 58
//    512 (B=16)     66            152                101                  74 -> 58
//   1024 (B=32)    138            134                102                  76 -> 58
//   4096 (B=128)                  414                397                 289 -> 228
//   8192 (B=256)   780            631                788                 571 -> 455
// The wide kernel wherever it fits (clouds of 256 points and more whose queues fit LDS), else the scan kernel with two
// waves per query tile (one from 4096 tiles).  Knob CLOUDAAE_KNN_SCAN forces a choice (the tests cover all of them):
// 1 / 2 = knn64_scan_kernel with one / two waves per query tile, 5 = knn64_wide_kernel (where it fits; otherwise 5 means 2).
static bool knn_wide_fits(int n, int k)
{
    return k <= 20 && n >= 256 && knn_wide_lds_bytes(n) <= 158 * 1024;
}

static int knn_scan_waves(long long tiles, int n, int k)
{
    if (CLOUDAAE_KNOB_SET("CLOUDAAE_KNN_SCAN")) {
        const int forced = CLOUDAAE_KNOB("CLOUDAAE_KNN_SCAN", 0);
        if (forced == 1 || forced == 2 || forced == 5)
            return forced;
    }
    if (knn_wide_fits(n, k))
        return 5;
    return tiles >= 4096 ? 1 : 2;
}

template <int K>
static hipError_t launch_knn(int b, int n, int c, int ld, int k, const float *x, int *nn_idx, hipStream_t s)
{
    const bool vec = ld % 4 == 0 && ((uintptr_t)x & 15) == 0;
    if (c == 3 && K <= 20 && n <= 6144) {
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
    } else if (c == 64 && vec && K <= 20 && n <= 16384) {
        if constexpr (K <= 20) {
            const long long tiles = (long long)ceil_div(n, KM_TILE) * b;
            const int mode = knn_scan_waves(tiles, n, K);
            if (mode == 5 && knn_wide_fits(n, K))
                return launch_knn_wide<K>(b, n, ld, k, x, nn_idx, s);
            if (mode == 1)
                return launch_knn_scan<K, 4, 1>(b, n, ld, k, x, nn_idx, s);
            return launch_knn_scan<K, 4, 2>(b, n, ld, k, x, nn_idx, s);
        }
    }
    // anything else (other channel counts, k above 20, clouds beyond the LDS-resident forms): one lane per query, exact, slow
    hipLaunchKernelGGL(knn_generic_kernel<K>, dim3(ceil_div(n, KNN_THREADS), b), dim3(KNN_THREADS), 0, s, n, c, ld, k, x,
                       nn_idx);
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

// cloudaae_knn over 64 channels with a HINT: hint[b, n, k] = k distinct indices per point that are likely to be near it (the
// neighbour lists of the layer before).  Same result as cloudaae_knn, whatever the hint holds; tau_scratch: b * n floats.
// Shapes the hinted kernel does not take (other channel counts, k above 20, clouds below 256 points or beyond the
// LDS-resident form) go through cloudaae_knn unchanged.
CLOUDAAE_API int cloudaae_knn_hinted(int b, int n, int c, int ld, int k, const float *x, const int *hint, float *tau_scratch,
                                     int *nn_idx, cloudaae_stream_t stream)
{
    const char *name = "cloudaae_knn_hinted";
    const bool vec = ld % 4 == 0 && ((uintptr_t)x & 15) == 0;
    if (hint == nullptr || tau_scratch == nullptr || c != 64 || !vec || k > 20 || b <= 0 || n <= 0 || k < 1 || k > n ||
        b > 65535 || !knn_wide_fits(n, k <= 10 ? 10 : 20))
        return cloudaae_knn(b, n, c, ld, k, x, nn_idx, stream);
    hipStream_t s = (hipStream_t)stream;
    const hipError_t e = k <= 10 ? launch_knn_wide_hinted<10>(b, n, ld, k, x, hint, tau_scratch, nn_idx, s)
                                 : launch_knn_wide_hinted<20>(b, n, ld, k, x, hint, tau_scratch, nn_idx, s);
    CLOUDAAE_CHECK_HIP(e, name);
    CLOUDAAE_CHECK_LAUNCH(name);
    return 0;
}
