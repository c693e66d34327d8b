/*
 * cloudaae_oracle.c -- CPU restatement of the CloudAAE native-op algorithms.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (cloudaae_amd/) never calls into it.
 *
 * Every function restates the arithmetic of one reference function and cites
 * it (paths relative to the reference checkout).  Index-producing distances
 * are evaluated un-fused, left to right, in fp32 -- build with
 * -ffp-contract=off (see oracle/Makefile) -- because the reference CPU object
 * does so.
 *
 * Pinning status:
 *   nn_distance fwd/bwd : pinned against the reference's own C++ lines
 *                         (oracle/_ref, built by oracle/build_ref.sh) and the
 *                         seeded known-answer input of tf_nndistance_cpu.py:28-46.
 *   fps / gather / knn  : the reference ships no CPU kernel, test or golden
 *                         vector for these -> "parity unpinned" by the
 *                         reference; pinned by property tests instead.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------
 * Chamfer nearest neighbour, one direction.
 * Follows tf_ops/nn_distance/tf_nndistance.cpp:21-43 (nnsearch):
 *   d = (dx*dx + dy*dy) + dz*dz with dx = q.x - p.x in fp32 (the reference
 *   stores the float result in a double and compares doubles, which orders
 *   exactly like the float compare); strict '<', first minimum wins;
 *   m == 0 leaves dist = 0, idx = 0.
 * ---------------------------------------------------------------------- */
static void nn_one_direction(int b, int n, int m, const float *from, const float *to,
                             float *dist, int32_t *idx, int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int c = 0; c < b; ++c) {
        const float *P = from + (size_t)c * n * 3;
        const float *Q = to + (size_t)c * m * 3;
        for (int j = 0; j < n; ++j) {
            const float px = P[3 * j], py = P[3 * j + 1], pz = P[3 * j + 2];
            float best = 0.0f;
            int32_t arg = 0;
            for (int k = 0; k < m; ++k) {
                const float dx = Q[3 * k] - px;
                const float dy = Q[3 * k + 1] - py;
                const float dz = Q[3 * k + 2] - pz;
                const float d = dx * dx + dy * dy + dz * dz;
                if (k == 0 || d < best) {
                    best = d;
                    arg = k;
                }
            }
            dist[(size_t)c * n + j] = best;
            idx[(size_t)c * n + j] = arg;
        }
    }
}

/* NnDistance forward: both directions (tf_nndistance.cpp:79-80). */
ORACLE_API void oracle_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2,
                                   float *dist1, int32_t *idx1, float *dist2, int32_t *idx2,
                                   int threads)
{
    nn_one_direction(b, n, m, xyz1, xyz2, dist1, idx1, threads);
    nn_one_direction(b, m, n, xyz2, xyz1, dist2, idx2, threads);
}

/* NnDistanceGrad (tf_nndistance.cpp:126-163): zero both outputs, then for each
 * cloud sweep direction 1 (own += , partner -=) and direction 2, sequentially,
 * so the fp32 summation order is the reference's. */
ORACLE_API void oracle_nn_distance_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                                        const float *grad_dist1, const int32_t *idx1,
                                        const float *grad_dist2, const int32_t *idx2,
                                        float *grad_xyz1, float *grad_xyz2, int threads)
{
    (void)threads;
    memset(grad_xyz1, 0, sizeof(float) * (size_t)b * n * 3);
    memset(grad_xyz2, 0, sizeof(float) * (size_t)b * m * 3);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int c = 0; c < b; ++c) {
        const float *A = xyz1 + (size_t)c * n * 3;
        const float *B = xyz2 + (size_t)c * m * 3;
        float *gA = grad_xyz1 + (size_t)c * n * 3;
        float *gB = grad_xyz2 + (size_t)c * m * 3;
        for (int j = 0; j < n; ++j) {
            const int32_t t = idx1[(size_t)c * n + j];
            const float g = grad_dist1[(size_t)c * n + j] * 2;
            for (int a = 0; a < 3; ++a) {
                const float v = g * (A[3 * j + a] - B[3 * t + a]);
                gA[3 * j + a] += v;
                gB[3 * t + a] -= v;
            }
        }
        for (int j = 0; j < m; ++j) {
            const int32_t t = idx2[(size_t)c * m + j];
            const float g = grad_dist2[(size_t)c * m + j] * 2;
            for (int a = 0; a < 3; ++a) {
                const float v = g * (B[3 * j + a] - A[3 * t + a]);
                gB[3 * j + a] += v;
                gA[3 * t + a] -= v;
            }
        }
    }
}

/* ------------------------------------------------------------------------
 * Farthest point sampling.
 * Follows tf_ops/sampling/tf_sampling_g.cu:105-170 (the only implementation the
 * reference has; there is no CPU kernel, tf_sampling.cpp:123).  The CUDA block
 * has 512 threads; thread t scans k = t, t+512, ... keeping its first maximum
 * (strict '>' from best = -1, besti = 0, :130-150), then a left-biased binary
 * tree over the 512 slots keeps the lower slot on ties (:153-163).  We replay
 * exactly that schedule so ties resolve identically: max value, then lowest
 * (k mod 512), then lowest k.  Running minimum starts at 1e38 (:117);
 * d = ((dx*dx + dy*dy) + dz*dz), un-fused fp32 (:142); first index is 0 (:113).
 * ---------------------------------------------------------------------- */
#define FPS_LANES 512
ORACLE_API void oracle_farthest_point_sample(int b, int n, int m, const float *inp, int32_t *out,
                                             int threads)
{
    (void)threads;
    if (m <= 0)
        return;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int c = 0; c < b; ++c) {
        const float *P = inp + (size_t)c * n * 3;
        float *running = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
        float slot_v[FPS_LANES];
        int32_t slot_i[FPS_LANES];
        for (int k = 0; k < n; ++k)
            running[k] = 1e38f;
        int32_t last = 0;
        out[(size_t)c * m] = last;
        for (int j = 1; j < m; ++j) {
            const float ox = P[3 * last], oy = P[3 * last + 1], oz = P[3 * last + 2];
            for (int t = 0; t < FPS_LANES; ++t) {
                float best = -1.0f;
                int32_t arg = 0;
                for (int k = t; k < n; k += FPS_LANES) {
                    const float dx = P[3 * k] - ox, dy = P[3 * k + 1] - oy, dz = P[3 * k + 2] - oz;
                    const float d = dx * dx + dy * dy + dz * dz;
                    const float d2 = d < running[k] ? d : running[k];
                    running[k] = d2;
                    if (d2 > best) {
                        best = d2;
                        arg = k;
                    }
                }
                slot_v[t] = best;
                slot_i[t] = arg;
            }
            for (int stride = 1; stride < FPS_LANES; stride <<= 1) {
                for (int lo = 0; lo + stride < FPS_LANES; lo += 2 * stride) {
                    const int hi = lo + stride;
                    if (slot_v[lo] < slot_v[hi]) {
                        slot_v[lo] = slot_v[hi];
                        slot_i[lo] = slot_i[hi];
                    }
                }
            }
            last = slot_i[0];
            out[(size_t)c * m + j] = last;
        }
        free(running);
    }
}

/* GatherPoint (tf_sampling_g.cu:172-181): out[c,j,:] = inp[c,idx[c,j],:]. */
ORACLE_API void oracle_gather_point(int b, int n, int m, const float *inp, const int32_t *idx,
                                    float *out)
{
    for (int c = 0; c < b; ++c)
        for (int j = 0; j < m; ++j) {
            const int32_t a = idx[(size_t)c * m + j];
            for (int d = 0; d < 3; ++d)
                out[((size_t)c * m + j) * 3 + d] = inp[((size_t)c * n + a) * 3 + d];
        }
}

/* GatherPointGrad (tf_sampling_g.cu:183-192, zero fill at tf_sampling.cpp:174):
 * inp_g[c,idx[c,j],:] += out_g[c,j,:], here in ascending j (the CUDA kernel
 * uses float atomics, i.e. an unspecified order). */
ORACLE_API void oracle_gather_point_grad(int b, int n, int m, const float *out_g,
                                         const int32_t *idx, float *inp_g)
{
    memset(inp_g, 0, sizeof(float) * (size_t)b * n * 3);
    for (int c = 0; c < b; ++c)
        for (int j = 0; j < m; ++j) {
            const int32_t a = idx[(size_t)c * m + j];
            for (int d = 0; d < 3; ++d)
                inp_g[((size_t)c * n + a) * 3 + d] += out_g[((size_t)c * m + j) * 3 + d];
        }
}

/* ------------------------------------------------------------------------
 * ProbSample = tf_sampling_g.cu:7-87 (cumsumKernel) + :89-104 (binarysearchKernel),
 * launched by probsampleLauncher (:198-201): for every row, an inclusive prefix
 * sum of the category weights, then for every uniform draw r the smallest index
 * whose prefix sum is >= r * total.
 * The prefix sum is NOT a left-to-right sum; its association order is part of
 * the result (it decides which index an r near a boundary gets), so it is
 * restated here exactly:
 *   - the row is cut into chunks of 8192 values (:8, BlockSize*4);
 *   - a chunk is cut into quads; quad prefix = a, a+b, c+(a+b), (d+c)+(a+b)
 *     (:20-33); a ragged last quad is summed left to right and padded with its
 *     total (:34-43);
 *   - the quad totals G[0..n2) get an in-place inclusive scan: up-sweep
 *     G[((2k+2)<<u)-1] += G[((2k+1)<<u)-1] for u = 0,1,.. while (2<<u) <= n2
 *     (:46-56), then down-sweep G[((2k+3)<<u)-1] += G[((2k+2)<<u)-1] for u back
 *     to 0 (:57-67);
 *   - quad g >= 1 adds G[g-1] to its four prefixes (:69-77);
 *   - the chunk adds the carry of the previous chunks, kept as a compensated
 *     pair: t = G[n2-1] + c2; r = c1 + t; c2 = t - (r - c1); c1 = r (:81-84).
 * Search (:89-104): q = r * cum[n-1]; idx = n-1; for k = pow2 >= n down to 1:
 * if (idx >= k && cum[idx-k] >= q) idx -= k.
 * `cum` (b*n floats) receives the prefix sums, as the reference's `temp`. */
ORACLE_API void oracle_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r,
                                   float *cum, int32_t *out)
{
    enum { CHUNK = 8192 };
    float *G = (float *)malloc(sizeof(float) * (CHUNK / 4));
    float *V = (float *)malloc(sizeof(float) * CHUNK);
    for (int row = 0; row < b; ++row) {
        const float *p = inp_p + (size_t)row * n;
        float *c = cum + (size_t)row * n;
        float carry = 0.0f, carry2 = 0.0f;
        for (int j = 0; j < n; j += CHUNK) {
            const int len = n - j < CHUNK ? n - j : CHUNK;
            const int padded = (len + 3) & ~3, n2 = padded >> 2;
            for (int g = 0; g < n2; ++g) {
                const int k = 4 * g;
                if (k + 3 < len) {
                    const float v1 = p[j + k];
                    float v2 = p[j + k + 1];
                    v2 = v2 + v1;
                    float v3 = p[j + k + 2];
                    float v4 = p[j + k + 3];
                    v4 = v4 + v3;
                    v3 = v3 + v2;
                    v4 = v4 + v2;
                    V[k] = v1; V[k + 1] = v2; V[k + 2] = v3; V[k + 3] = v4;
                    G[g] = v4;
                } else {
                    float v = 0.0f;
                    for (int t = k; t < len; ++t) {
                        v = v + p[j + t];
                        V[t] = v;
                    }
                    for (int t = len; t < padded; ++t)
                        V[t] = v;
                    G[g] = v;
                }
            }
            int u = 0;
            for (; (2 << u) <= n2; ++u)
                for (int k = 0; k < (n2 >> (u + 1)); ++k)
                    G[(((k << 1) + 2) << u) - 1] += G[(((k << 1) + 1) << u) - 1];
            for (--u; u >= 0; --u)
                for (int k = 0; k < ((n2 - (1 << u)) >> (u + 1)); ++k)
                    G[(((k << 1) + 3) << u) - 1] += G[(((k << 1) + 2) << u) - 1];
            for (int g = 1; g < n2; ++g)
                for (int t = 0; t < 4; ++t)
                    V[4 * g + t] = V[4 * g + t] + G[g - 1];
            for (int t = 0; t < len; ++t)
                c[j + t] = V[t] + carry;
            const float tt = G[n2 - 1] + carry2;
            const float r2 = carry + tt;
            carry2 = tt - (r2 - carry);
            carry = r2;
        }
        int base = 1;
        while (base < n)
            base <<= 1;
        for (int q = 0; q < m; ++q) {
            const float key = inp_r[(size_t)row * m + q] * c[n - 1];
            int r = n - 1;
            for (int k = base; k >= 1; k >>= 1)
                if (r >= k && c[r - k] >= key)
                    r -= k;
            out[(size_t)row * m + q] = r;
        }
    }
    free(G);
    free(V);
}

/* ------------------------------------------------------------------------
 * kNN grouping = utils/tf_util.py:597-618 (pairwise_xyz_distance) followed by
 * utils/tf_util.py:621-632 (knn = top_k of the negated matrix).
 *   D[i][j] = (sq[i] + (-2 * inner[i][j])) + sq[j]                  (:618)
 *   sq[i]   = sum_c x[i][c]^2       (square rounded, then summed; :615)
 *   inner   = X * X^T               (tf.matmul; :613)
 *   result  = the k smallest D[i][*], ascending, ties -> lower j   (TopKV2)
 * TensorFlow/Eigen's summation order inside matmul and reduce_sum is not
 * pinned by anything in the reference ("parity
 * unpinned").  We DEFINE: sq as a sequential un-fused sum over c, inner as a
 * sequential k-ordered fmaf chain starting from +0 -- which is bitwise what
 * gfx950's v_mfma_f32_* computes (MI355X_MICROARCH.md, matrix cores).
 * `x` is [b][n][ld] with the first `c` channels of each row used, so the
 * layer-1 call (xyz = first 3 of 24 channels, tf_util.py:608) needs no copy.
 * ---------------------------------------------------------------------- */
ORACLE_API void oracle_knn(int b, int n, int c, int ld, int k, const float *x, int32_t *nn_idx,
                           float *nn_dist /* may be NULL */, int threads)
{
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
    for (int cl = 0; cl < b; ++cl) {
        const float *X = x + (size_t)cl * n * ld;
        float *sq = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
        float *bd = (float *)malloc(sizeof(float) * (size_t)(k > 0 ? k : 1));
        int32_t *bi = (int32_t *)malloc(sizeof(int32_t) * (size_t)(k > 0 ? k : 1));
        for (int i = 0; i < n; ++i) {
            float s = 0.0f;
            for (int ch = 0; ch < c; ++ch) {
                const float v = X[(size_t)i * ld + ch];
                const float v2 = v * v;
                s = s + v2;
            }
            sq[i] = s;
        }
        for (int i = 0; i < n; ++i) {
            int filled = 0;
            for (int j = 0; j < n; ++j) {
                float inner = 0.0f;
                for (int ch = 0; ch < c; ++ch)
                    inner = fmaf(X[(size_t)i * ld + ch], X[(size_t)j * ld + ch], inner);
                const float m2 = -2.0f * inner;
                const float t = sq[i] + m2;
                const float d = t + sq[j];
                /* stable insertion: j ascends, so an equal distance never
                 * overtakes an earlier index */
                if (filled < k) {
                    int p = filled++;
                    while (p > 0 && bd[p - 1] > d) {
                        bd[p] = bd[p - 1];
                        bi[p] = bi[p - 1];
                        --p;
                    }
                    bd[p] = d;
                    bi[p] = j;
                } else if (k > 0 && d < bd[k - 1]) {
                    int p = k - 1;
                    while (p > 0 && bd[p - 1] > d) {
                        bd[p] = bd[p - 1];
                        bi[p] = bi[p - 1];
                        --p;
                    }
                    bd[p] = d;
                    bi[p] = j;
                }
            }
            for (int p = 0; p < k; ++p) {
                nn_idx[((size_t)cl * n + i) * k + p] = p < filled ? bi[p] : 0;
                if (nn_dist)
                    nn_dist[((size_t)cl * n + i) * k + p] = p < filled ? bd[p] : 0.0f;
            }
        }
        free(sq);
        free(bd);
        free(bi);
    }
}

/* Materialised pairwise matrix of one cloud (tf_util.py:597-618), same
 * definition as above; used by tests to check oracle_knn against a stable
 * argsort and by property tests of the HIP kernel. */
ORACLE_API void oracle_pairwise_distance(int n, int c, int ld, const float *X, float *D)
{
    for (int i = 0; i < n; ++i) {
        float si = 0.0f;
        for (int ch = 0; ch < c; ++ch) {
            const float v = X[(size_t)i * ld + ch];
            const float v2 = v * v;
            si = si + v2;
        }
        for (int j = 0; j < n; ++j) {
            float sj = 0.0f, inner = 0.0f;
            for (int ch = 0; ch < c; ++ch) {
                const float v = X[(size_t)j * ld + ch];
                const float v2 = v * v;
                sj = sj + v2;
                inner = fmaf(X[(size_t)i * ld + ch], v, inner);
            }
            const float m2 = -2.0f * inner;
            const float t = si + m2;
            D[(size_t)i * n + j] = t + sj;
        }
    }
}

ORACLE_API int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
