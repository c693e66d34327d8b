/*
 * cloudaae_hip.h -- C ABI of libcloudaae_hip.so, the MI355X (gfx950) drop-in for
 * the native hot path of GeeeG/CloudAAE.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer to contiguous row-major memory owned by
 *     the caller (outputs and workspaces included).  Two entry points take
 *     stream-ordered scratch of ONE call themselves (hipMallocAsync / hipFreeAsync
 *     on the caller's stream, nothing outlives the call): cloudaae_nn_distance*
 *     when it cuts a direction's candidates into ranges, and cloudaae_gemm_bf16x3
 *     for the planes of its second operand (cloudaae_gemm_bf16x3p takes them from
 *     the caller instead);
 *   - float = IEEE fp32, int = int32; sizes are element counts;
 *   - every function takes the HIP stream to launch on as its last argument
 *     (a hipStream_t passed as void*; NULL = the default stream), is re-entrant,
 *     never synchronises, and returns 0 or a hipError_t value
 *     (cloudaae_last_error() describes the most recent failure of the calling
 *     thread).  Process-wide state is limited to: the table of development knobs
 *     (cloudaae_set_knob; set them between launches, not concurrently with them),
 *     one low-priority side stream (cloudaae_side_stream), and per-device flags
 *     that record which kernels had their dynamic-LDS limit raised;
 *   - a workspace whose size comes from a cloudaae_*_workspace / *_partials query
 *     is only as large as the K split the knobs implied AT THE QUERY; the entry
 *     points that take one also take its size in floats and fail (no launch) when
 *     the cut they derive at launch time needs more -- e.g. after a change of
 *     CLOUDAAE_GEMM_SPLITS / CLOUDAAE_FC_FWD_SPLITS / CLOUDAAE_FC_FWD_BLOCKS /
 *     CLOUDAAE_DETERMINISTIC between the query and a launch or a replay;
 *   - gradient outputs are zero-filled by the callee.
 * The reference's launchers have C++ linkage, no stream and no status
 * (tf_nndistance.cpp:168,208; tf_sampling.cpp:65,94,125,150); each entry point
 * below names the one it replaces and keeps its argument order.
 */
#ifndef CLOUDAAE_HIP_H
#define CLOUDAAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *cloudaae_stream_t; /* hipStream_t */

/* The ABI revision this header describes: argument lists and struct layouts.  A caller built against another
 * revision must not call in -- check cloudaae_version() == CLOUDAAE_ABI_VERSION after loading (the Python host
 * does, cloudaae_amd/_lib.py).  500: round 5 (fully connected entry points take up to 128 rows; tickets / partials
 * queries take M; no y_zeroed argument).  600: round 6 (cloudaae_knn_hinted added; nothing else changed).
 * 601: cloudaae_selftest_div_by added. */
#define CLOUDAAE_ABI_VERSION 601
int cloudaae_version(void);
const char *cloudaae_last_error(void);
/* Development knobs (kernel A/B choices and launch shapes for tests and sweeps; none is needed in normal use):
 * an integer per name, initialised from the environment variable of the same name the first time the library
 * looks at it (the library never reads the environment again), changed with these two calls.  Names are listed
 * in DESIGN.md ("Development knobs"), e.g. "CLOUDAAE_KNN_SCAN", "CLOUDAAE_NN_FILTER".
 * One knob is a MODE rather than a tuning aid: "CLOUDAAE_DETERMINISTIC" = 1 keeps every product of cloudaae_gemm_* whole
 * over K (no slices added with atomics) and sorts the reverse neighbour lists of the edge convolution's backward pass;
 * together with cloudaae_nn_distance_grad_ordered and the GEMM + batch-norm route for the fully connected stack (the
 * host's choices: TrainGraph(deterministic=True)) a whole training step is then bit-reproducible from run to run, as
 * the reference's sequential CPU path is. */
int cloudaae_set_knob(const char *name, int value);
int cloudaae_unset_knob(const char *name);
/* HOST helper: CRC-32C (Castagnoli, reflected, init/final xor 0xffffffff) of n bytes of host memory -- the
 * checksum of the TFRecord framing (train_cloudAAE_ycbv.py:80-135) and of tf.train.Saver checkpoints
 * (:276, :418-430).  crc = 0 for a whole buffer, or the previous result to continue over the next piece. */
unsigned cloudaae_crc32c(const void *data, unsigned long long n, unsigned crc);

/* Two-stream plumbing.  cloudaae_side_stream(): a low-priority stream owned by the library (one per
 * process), for work off the critical path; NULL on failure.  cloudaae_stream_wait(waiter, signaller):
 * everything enqueued on `signaller` so far completes before anything enqueued on `waiter` from now on
 * (an event record + a stream wait; no host synchronisation).  A caller that hands a side stream to
 * cloudaae_edgeconv_backward, or launches on it itself, joins with cloudaae_stream_wait(main, side)
 * before it consumes the results. */
cloudaae_stream_t cloudaae_side_stream(void);
int cloudaae_stream_wait(cloudaae_stream_t waiter, cloudaae_stream_t signaller);

/* ---- tf_ops/nn_distance ------------------------------------------------- */

/* NnDistance forward, both directions in one launch.
 * Replaces: void NmDistanceKernelLauncher(int b,int n,const float* xyz,int m,
 *   const float* xyz2,float* result,int* result_i,float* result2,int* result2_i)
 *   (tf_ops/nn_distance/tf_nndistance.cpp:168, tf_nndistance_g.cu:128-131);
 * numerics of the CPU op (tf_nndistance.cpp:21-43): squared L2, un-fused fp32,
 * first minimum wins; m == 0 gives dist 0 / idx 0.
 * xyz1 [b,n,3], xyz2 [b,m,3] -> dist1 [b,n], idx1 [b,n], dist2 [b,m], idx2 [b,m].
 * Clouds of very unequal size take stream-ordered scratch (hipMallocAsync / hipFreeAsync on `stream`, nothing
 * outlives the call); the first such call raises the release threshold of the device's default memory pool so
 * that later calls reuse the memory instead of returning it to the driver at every synchronisation. */
int cloudaae_nn_distance(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1,
                         int *idx1, float *dist2, int *idx2, cloudaae_stream_t stream);
/* The same search when cloud c's xyz2 is `count2[c]` distinct points followed by copies of them -- the reference's own
 * Chamfer targets: the visible points, then random re-draws of visible points up to a fixed row count
 * (utils/hidden_point_removal.py:38-43, 72; sliced at train_cloudAAE_ycbv.py:211-214).  row_src2 [b,m] names, for every
 * row j >= count2[c], the row < count2[c] it is a bitwise copy of (cloudaae_hidden_point_removal_rows writes both).
 * Results are those of cloudaae_nn_distance bit for bit ("first index wins" puts every answer among the originals; a
 * copy's own answer is its original's), at the cost of the distinct points only.  count2[c] outside (0, m]: all m rows
 * are searched.  Both NULL: cloudaae_nn_distance.  A copy row whose row_src2 is not a distinct row (< 0, >= count2[c]) gets
 * dist2 = NaN and idx2 = 0 -- every output element is written; the development knob CLOUDAAE_NN_PREFIX_VERIFY = 1 also
 * compares every copy with its original bit for bit (NaN on a mismatch): the hint is only valid for targets that are still
 * what cloudaae_hidden_point_removal_rows wrote (no shuffle, slice beyond the rows, or augmentation in between). */
int cloudaae_nn_distance_prefix(int b, int n, const float *xyz1, int m, const float *xyz2, const long long *count2,
                                const int *row_src2, float *dist1, int *idx1, float *dist2, int *idx2,
                                cloudaae_stream_t stream);

/* Development / test entry: the search scores of cloudaae_nn_distance's large-cloud kernel (|b'|^2 - 2 a'.b' of coordinates
 * centred on candidates[0], as error-free three-piece bfloat16 split products on the bf16 matrix pipe) for nq <= 32 queries
 * [nq,3] against nc <= 32 candidates [nc,3]: scores[q * 32 + c], and R[q] = (|a'_q| + max_c |b'_c|)^2, the quantity the
 * kernel's decision margin 160 * 2^-24 * R is stated in.  Same operand construction and instructions as the kernel. */
int cloudaae_dev_nn_split_scores(int nq, int nc, const float *queries, const float *candidates, float *scores, float *R,
                                 cloudaae_stream_t stream);

/* NnDistanceGrad.
 * Replaces: void NmDistanceGradKernelLauncher(int b,int n,const float* xyz1,int m,
 *   const float* xyz2,const float* grad_dist1,const int* idx1,const float* grad_dist2,
 *   const int* idx2,float* grad_xyz1,float* grad_xyz2)
 *   (tf_nndistance.cpp:208, tf_nndistance_g.cu:152-157; CPU loops tf_nndistance.cpp:126-163).
 * Either gradient output may be NULL (not wanted).  The terms of a point are added in no fixed order (atomics, in LDS per
 * 2048-point chunk of an output; the reference's own GPU kernel adds with global atomics): fp32 round-off apart from the
 * sequential sweep.  Every element of a wanted output is written. */
int cloudaae_nn_distance_grad(int b, int n, const float *xyz1, int m, const float *xyz2,
                              const float *grad_dist1, const int *idx1, const float *grad_dist2,
                              const int *idx2, float *grad_xyz1, float *grad_xyz2,
                              cloudaae_stream_t stream);
/* The same gradients accumulated in the ORDER of the reference's sequential CPU loops (tf_nndistance.cpp:126-163:
 * sweep over xyz1, then over xyz2), without atomics: bit-identical to that sweep and reproducible from run to run
 * (the atomic kernel above agrees with it to fp32 round-off only, as the reference's own GPU kernel does).
 * uniform != NULL: every distance has the upstream gradient uniform[0] * uniform_scale (the mean of
 * losses/chamfer_loss.py:13-14) and grad_dist1 / grad_dist2 are not read.  O(n m) index comparisons per cloud:
 * about ten times the time of the atomic kernel -- the deterministic mode's choice. */
int cloudaae_nn_distance_grad_ordered(int b, int n, const float *xyz1, int m, const float *xyz2,
                                      const float *grad_dist1, const int *idx1, const float *grad_dist2,
                                      const int *idx2, const float *uniform, float uniform_scale,
                                      float *grad_xyz1, float *grad_xyz2, cloudaae_stream_t stream);
/* The same when every distance has the SAME upstream gradient grad[0] * scale (the Chamfer loss is a mean
 * over them, losses/chamfer_loss.py:13-14): no per-point gradient arrays.  outputs_zeroed: only read by the
 * global-atomic development form (CLOUDAAE_NND_GRAD_LDS=0), which adds into its outputs (!= 0: the caller zero-filled
 * them, else the call clears them first); the default form stores every output element. */
int cloudaae_nn_distance_grad_uniform(int b, int n, const float *xyz1, int m, const float *xyz2, const float *grad,
                                      float scale, const int *idx1, const int *idx2, float *grad_xyz1,
                                      float *grad_xyz2, int outputs_zeroed, cloudaae_stream_t stream);

/* ---- tf_ops/sampling ---------------------------------------------------- */

/* FarthestPointSample: inp [b,n,3] -> out [b,m] (first index 0).
 * Replaces: void farthestpointsamplingLauncher(int b,int n,int m,const float* inp,
 *   float* temp,int* out) (tf_ops/sampling/tf_sampling.cpp:94, tf_sampling_g.cu:203-205;
 *   kernel :105-170, whose tie-break -- max, then lowest k mod 512, then lowest k --
 *   is reproduced).  `temp` is the reference's 32*n-float workspace
 *   (tf_sampling.cpp:115); it is only touched when n > 16384 and may be NULL
 *   otherwise. */
int cloudaae_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out,
                                   cloudaae_stream_t stream);

/* GatherPoint: out[i,j,:] = inp[i,idx[i,j],:].
 * Replaces: void gatherpointLauncher(int b,int n,int m,const float* inp,const int* idx,
 *   float* out) (tf_sampling.cpp:125, tf_sampling_g.cu:172-181,206-208). */
int cloudaae_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out,
                          cloudaae_stream_t stream);

/* GatherPointGrad: inp_g[i,idx[i,j],:] += out_g[i,j,:]; inp_g is zero-filled HERE
 * (the reference zeroes it in the Op, tf_sampling.cpp:174).
 * Replaces: void scatteraddpointLauncher(int b,int n,int m,const float* out_g,
 *   const int* idx,float* inp_g) (tf_sampling.cpp:150, tf_sampling_g.cu:183-192,209-211). */
int cloudaae_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx,
                               float *inp_g, cloudaae_stream_t stream);

/* out[b,m] = for every draw inp_r[b,m] in [0,1) the smallest category whose inclusive prefix sum of
 * inp_p[b,n] reaches inp_r * total; temp[b*n] receives the prefix sums (the reference's workspace).
 * The prefix sums keep the reference's association order (quads, scan tree, compensated chunk
 * carry), so the indices are those of its kernels.  No gradient (tf_sampling.py:22).
 * Replaces: void probsampleLauncher(int b,int n,int m,const float* inp_p,const float* inp_r,
 *   float* temp,int* out) (tf_sampling.cpp:65, tf_sampling_g.cu:7-104,198-201). */
int cloudaae_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp,
                         int *out, cloudaae_stream_t stream);

/* ---- utils/tf_util.py: kNN grouping ------------------------------------- */

/* pairwise_xyz_distance + knn fused (utils/tf_util.py:597-632): for every point
 * the k nearest points of its own cloud (self included), ascending distance,
 * ties -> lower index; the [b,n,n] matrix is never materialised.
 * x [b,n,ld] of which the first c channels are the metric (c = 3: the xyz slice
 * of tf_util.py:608; c = 64: the later layers); nn_idx [b,n,k].  k <= 32.
 * D[i][j] = (|x_i|^2 + (-2 * <x_i,x_j>)) + |x_j|^2 with <,> a channel-ordered
 * fp32 fma chain and |.|^2 a sequential un-fused sum (oracle_knn). */
int cloudaae_knn(int b, int n, int c, int ld, int k, const float *x, int *nn_idx,
                 cloudaae_stream_t stream);
/* The same with a HINT (revision 600): hint [b,n,k] int32 = k DISTINCT indices per point that are likely to be near it -- the
 * neighbour lists of the layer before (models/pointnet_ycb_23_decoder_4.py:337-404 recomputes the lists layer by layer on
 * features that change little).  Their largest distance bounds the k-th distance, so the scan needs no bound pass of its
 * own.  The RESULT is cloudaae_knn's whatever the hint holds (a bad one costs time only).  tau_scratch: b * n floats.
 * Shapes outside the hinted kernel (c != 64, k > 20, clouds below 256 points, hint == NULL) take cloudaae_knn's path. */
int cloudaae_knn_hinted(int b, int n, int c, int ld, int k, const float *x, const int *hint, float *tau_scratch,
                        int *nn_idx, cloudaae_stream_t stream);

/* ---- utils/tf_util.py: dense layers ------------------------------------- */

/* C[M,N] (+)= op(A)[M,K] op(B)[K,N] (+ bias[N]); fp32 in / fp32 accumulate on the
 * matrix cores (v_mfma_f32_32x32x2_f32 = k-ordered fmaf chain).  Row-major.
 * trans_a: A is stored [K][M]; trans_b: B is stored [N][K].  accumulate: 0 overwrite C, 1 add to C,
 * 2 C already holds zeros (skips the clear pass of a split-K product).
 * Replaces tf.nn.conv2d 1x1 + bias_add (utils/tf_util.py:161-166) and tf.matmul +
 * bias_add (utils/tf_util.py:349-352) and their two gradient products. */
int cloudaae_gemm_f32(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                      const float *B, int ldb, float *C, int ldc, const float *bias, int accumulate,
                      cloudaae_stream_t stream);
/* cloudaae_gemm_f32 (overwrite mode) that also leaves, per tile row of the product, the column sums and
 * sums of squares of C in fp64: colstats[parts][2][N], parts = cloudaae_gemm_f32_colstats_parts(M, N, K)
 * (0: this shape is split over K and has no such variant).  Feeds cloudaae_bn_forward_colstats. */
int cloudaae_gemm_f32_colstats_parts(int M, int N, int K);
int cloudaae_gemm_f32_colstats(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                               const float *B, int ldb, float *C, int ldc, const float *bias, double *colstats,
                               cloudaae_stream_t stream);
/* K slices cloudaae_gemm_f32 will use for this shape (> 1: the output is combined with atomics and
 * must hold zeros first -- the call clears it itself unless accumulate is 1 or 2). */
int cloudaae_gemm_f32_splits(int M, int N, int K);
/* cloudaae_gemm_f32 (overwrite mode) whose result is BIT-REPRODUCIBLE from run to run: a product cut over K keeps
 * its slices apart in workspace (cloudaae_gemm_f32_ordered_workspace(M, N, K) floats; 0 = K stays whole and
 * workspace may be NULL) and a second kernel sums them in slice order, then adds the bias.  C need not be
 * cleared.  Used for every FORWARD product (the reference's CPU path is sequential and deterministic;
 * evaluate_cloudAAE_ycbv.py:421-477 returns the same reconstruction for the same frame). */
long long cloudaae_gemm_f32_ordered_workspace(int M, int N, int K);
int cloudaae_gemm_f32_ordered(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                              const float *B, int ldb, float *C, int ldc, const float *bias, float *workspace,
                              long long workspace_floats, cloudaae_stream_t stream);
/* Several independent weight-gradient products C_j += A_j^T B_j (A_j stored [K][M], B_j [K][N]: dW = x^T dy of
 * utils/tf_util.py:161-166 for several layers) in ONE launch: each is a single wave of short split-K workgroups on its
 * own, and nothing waits for them before the optimiser.  C_j is added to with atomics and must hold zeros: zeroed != 0
 * says the caller cleared it, else the call does.  fold_c: 0, or the power-of-two width at which C's logical columns
 * fold into stacked row blocks (the edge convolution's [2*cin, cout] kernel used as [cin, 2*cout]); then ldc == fold_c. */
typedef struct cloudaae_gemm_tn_job {
    int M, N, K;
    const float *A;
    int lda;
    const float *B;
    int ldb;
    float *C;
    int ldc;
    int fold_c;
    int zeroed;
} cloudaae_gemm_tn_job;
int cloudaae_gemm_f32_tn_group(int count, const cloudaae_gemm_tn_job *jobs, cloudaae_stream_t stream);
/* cloudaae_gemm_f32_ordered with C's logical columns folded into stacked row blocks of width fold_c (as in
 * cloudaae_gemm_tn_job; 0: none), no bias: the edge convolution's weight gradients in deterministic mode. */
int cloudaae_gemm_f32_ordered_fold(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                                   const float *B, int ldb, float *C, int ldc, int fold_c, float *workspace,
                                   long long workspace_floats,
                                   cloudaae_stream_t stream);
/* The same product with both operands rounded to bfloat16 (round to nearest even) on their way
 * to the matrix cores (v_mfma_f32_32x32x16_bf16), fp32 accumulate; A, B, C, bias stay fp32 in
 * memory, so the call is interchangeable with cloudaae_gemm_f32 (BASELINE configs[2]: bf16 MLPs). */
int cloudaae_gemm_bf16(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                       const float *B, int ldb, float *C, int ldc, const float *bias, int accumulate,
                       cloudaae_stream_t stream);
int cloudaae_gemm_bf16_splits(int M, int N, int K);
/* cloudaae_gemm_f32_ordered_workspace / _ordered for the bf16-operand product. */
long long cloudaae_gemm_bf16_ordered_workspace(int M, int N, int K);
int cloudaae_gemm_bf16_ordered(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                               const float *B, int ldb, float *C, int ldc, const float *bias, float *workspace,
                               long long workspace_floats,
                               cloudaae_stream_t stream);
/* cloudaae_gemm_f32_colstats_parts / _colstats for the bf16-operand product. */
int cloudaae_gemm_bf16_colstats_parts(int M, int N, int K);
int cloudaae_gemm_bf16_colstats(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                                const float *B, int ldb, float *C, int ldc, const float *bias, double *colstats,
                                cloudaae_stream_t stream);

/* fp32 products on the bf16 matrix cores by error-free splitting (opt-in; `gemm_dtype = "bf16x3"` in the Python host):
 * every operand element is split exactly into three bfloat16 pieces and the six piece products of weight >= 2^-16
 * are accumulated in fp32 -- what is dropped is below 2^-23 of each product, the size of one fp32 rounding, so the
 * result agrees with cloudaae_gemm_f32 at the level of an fp32 accumulation in another order -- at 2.7 x less matrix-pipe
 * time (six 32-cycle bf16 MFMAs per 16 k against eight 64-cycle fp32 MFMAs).  Arguments as cloudaae_gemm_f32 plus the
 * optional column sums of cloudaae_gemm_f32_colstats (cloudaae_gemm_bf16x3_colstats_parts tile rows).  Served: whole
 * tiles of 128 / 160, K % 32 == 0, 16-byte aligned rows, not both operands transposed
 * (cloudaae_gemm_bf16x3_supported); accumulate: 0 overwrite, 1 add, 2 add into a C the caller cleared. */
int cloudaae_gemm_bf16x3_supported(int trans_a, int trans_b, int M, int N, int K);
int cloudaae_gemm_bf16x3_colstats_parts(int M, int N, int K);
int cloudaae_gemm_bf16x3(int trans_a, int trans_b, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                         float *C, int ldc, const float *bias, int accumulate, double *colstats, cloudaae_stream_t stream);
/* The same split products with the second operand split ONCE by the caller (a weight multiplies every row tile of a
 * step's forward AND backward product): cloudaae_x3_split writes the three bfloat16 planes of a [rows][k] matrix
 * (src = [rows][k], or [k][rows] when transposed != 0; cloudaae_x3_planes_bytes(rows, k) = 6 rows k bytes, 16-byte
 * aligned, k % 32 == 0) in the order the matrix cores read them, and cloudaae_gemm_bf16x3p computes
 * C[M,N] (+)= A[M,K] P^T (+ bias) with P the planes of the [N][K] operand -- y = x W with the planes of W^T
 * (cloudaae_x3_split(N, K, W, ldw, 1, ..)), dx = dy W^T with the planes of W (cloudaae_x3_split(K_w, N_w, W, ldw, 0, ..)).
 * A stays fp32 and is split in registers, once per element.  Served: M % 128 == 0, N a multiple of 128 or 160,
 * K % 32 == 0, rows of A 16-byte aligned (cloudaae_gemm_bf16x3p_supported); colstats as cloudaae_gemm_f32_colstats with
 * cloudaae_gemm_bf16x3p_colstats_parts(M, N, K) tile rows.  cloudaae_gemm_bf16x3 itself takes this route for
 * trans_a == 0 with planes in stream-ordered scratch of the call. */
long long cloudaae_x3_planes_bytes(int rows, int k);
int cloudaae_x3_split(int rows, int k, const float *src, int ld, int transposed, void *planes, cloudaae_stream_t stream);
/* both plane sets of a weight W[K][N] in one launch: planes_fwd = cloudaae_x3_split(N, K, W, ldw, 1, ..) for y = x W,
 * planes_bwd = cloudaae_x3_split(K, N, W, ldw, 0, ..) for dx = dy W^T (6 K N bytes each; K, N multiples of 32) */
int cloudaae_x3_split_weight(int K, int N, const float *W, int ldw, void *planes_fwd, void *planes_bwd,
                             cloudaae_stream_t stream);
int cloudaae_gemm_bf16x3p_supported(int M, int N, int K);
int cloudaae_gemm_bf16x3p_colstats_parts(int M, int N, int K);
int cloudaae_gemm_bf16x3p(int M, int N, int K, const float *A, int lda, const void *planes, float *C, int ldc,
                          const float *bias, int accumulate, double *colstats, cloudaae_stream_t stream);

/* ---- activations kept as bfloat16 in HBM (BASELINE configs[2]: "bf16 MLPs") --------------------------------------
 * The same products as cloudaae_gemm_bf16 (conv2d 1x1 and its two gradient products, utils/tf_util.py:161-166) with
 * operands that already ARE bfloat16 in memory (uint16_t = the upper half of the fp32 pattern, round to nearest
 * even), fp32 accumulate; C is fp32, or bfloat16 when c_is_bf16 != 0.  Served: whole tiles only --
 * cloudaae_gemm_b16_supported() says whether (trans_a, trans_b, M, N, K) is (K % 64 == 0, M and N multiples of the
 * 128 / 160 tile sides, not both operands transposed); rows of A, B (and of a bf16 C) 16-byte aligned.
 * colstats (optional, as cloudaae_gemm_f32_colstats, cloudaae_gemm_b16_colstats_parts(M, N, K) tile rows): column
 * sums and sums of squares of the fp32 values C was rounded from.  accumulate: 0 overwrite, 1 add to C (fp32 C). */
int cloudaae_gemm_b16_supported(int trans_a, int trans_b, int M, int N, int K);
int cloudaae_gemm_b16_colstats_parts(int M, int N, int K);
int cloudaae_gemm_b16(int trans_a, int trans_b, int M, int N, int K, const uint16_t *A, int lda, const uint16_t *B,
                      int ldb, void *C, int ldc, int c_is_bf16, const float *bias, int accumulate, double *colstats,
                      cloudaae_stream_t stream);
/* dst[i] = bfloat16(src[i]) (round to nearest even), n a multiple of 8, both 16-byte aligned. */
int cloudaae_to_bf16(long long n, const float *src, uint16_t *dst, cloudaae_stream_t stream);
/* batch_norm_template + ReLU + reduce_mean over groups of pool_rows rows (models/pointnet_ycb_23_decoder_4.py:
 * 410-419) in training mode on a bfloat16 y[M,C]: the moments come from `colstats` (the fp32 column sums the product
 * left), the EMA shadows are updated, pooled[M/pool_rows, C] and pool_stats[M/pool_rows][3][C] are written as by
 * cloudaae_bn_forward_colstats(pool_mode 1, relu).  C % 256 == 0, pool_rows % 64 == 0. */
int cloudaae_bn_meanpool_forward16(int M, int C, const uint16_t *y, int ldy, const float *gamma, const float *beta,
                                   const float *decay, float *ema_mean, float *ema_var, float *save_mean,
                                   float *save_var, int pool_rows, float *pooled, double *pool_stats, void *workspace,
                                   const double *colstats, int colstats_parts, cloudaae_stream_t stream);
/* Its gradient: dy[M,C] (bfloat16) from dpooled[M/pool_rows, C]; dgamma / dbeta / dbias as cloudaae_bn_backward. */
int cloudaae_bn_meanpool_backward16(int M, int C, const uint16_t *y, int ldy, const float *gamma, const float *beta,
                                    const float *save_mean, const float *save_var, int pool_rows, const float *dpooled,
                                    uint16_t *dy, int lddy, float *dgamma, float *dbeta, float *dbias,
                                    int accumulate_param_grads, const double *pool_stats, void *workspace,
                                    cloudaae_stream_t stream);

/* batch_norm_template (utils/tf_util.py:473-511) on rows y[M,C] (+ ReLU), writing the
 * activation out[M,C] and/or its pool over groups of pool_rows consecutive rows
 * (pool_mode 0 none, 1 mean = models/pointnet_ycb_23_decoder_4.py:419, 2 max = :684,
 * :59-60; tie_count[M/pool_rows,C] receives the number of equal maxima).
 * training != 0: batch moments (biased variance), EMA shadows updated as
 * s -= (s - stat) * (1 - decay[0]) when ema_mean != NULL; training == 0: moments =
 * EMA shadows.  save_mean/save_var[C] receive the moments used.  eps = 1e-3.
 * workspace: cloudaae_bn_workspace_bytes(C) bytes.  * pool_stats (optional, mean pool + ReLU in training mode): 3 x C doubles per group -- rows passing the
 * ReLU, sum of their x_hat, sum of all x_hat -- which cloudaae_bn_backward(pool_stats=...) turns into its
 * column sums without a pass over y when the pooled value is the only consumer (dout == NULL). */
long long cloudaae_bn_workspace_bytes(int C);
int cloudaae_bn_forward(int M, int C, const float *y, int ldy, const float *gamma, const float *beta,
                        int training, const float *decay, float *ema_mean, float *ema_var,
                        float *save_mean, float *save_var, int relu, float *out, int ldo, int pool_rows,
                        int pool_mode, float *pooled, float *tie_count, double *pool_stats, void *workspace,
                        cloudaae_stream_t stream);
/* The same with the statistics pass skipped: colstats[colstats_parts][2][C] already holds per-row-tile
 * column sums / sums of squares of y, written by cloudaae_gemm_f32_colstats (the product that made y had
 * the tile in registers anyway: for dgcnn_agg this saves reading 134 MB). */
int cloudaae_bn_forward_colstats(int M, int C, const float *y, int ldy, const float *gamma, const float *beta,
                                 int training, const float *decay, float *ema_mean, float *ema_var,
                                 float *save_mean, float *save_var, int relu, float *out, int ldo, int pool_rows,
                                 int pool_mode, float *pooled, float *tie_count, double *pool_stats, void *workspace,
                                 const double *colstats, int colstats_parts, cloudaae_stream_t stream);
/* gradient of the above: upstream = dout[M,C] (may be NULL) and/or dpooled[M/pool_rows,C]
 * (mean: /pool_rows; max: shared among equal maxima, as tf.reduce_max does);
 * produces dy[M,C], dgamma[C], dbeta[C] (NULL = not wanted), and dbias[C] (NULL = not wanted): the
 * gradient of a bias added to y right before the batch norm (tf_util.py:166 / :352 followed by
 * :173 / :355), i.e. the column sums of dy, without another pass over dy. */
int cloudaae_bn_backward(int M, int C, const float *y, int ldy, const float *gamma, const float *beta,
                         const float *save_mean, const float *save_var, int training, int relu,
                         const float *dout, int lddo, int pool_rows, int pool_mode, const float *dpooled,
                         const float *pooled, const float *tie_count, float *dy, int lddy, float *dgamma,
                         float *dbeta, float *dbias, int accumulate_param_grads, const double *pool_stats,
                         void *workspace, cloudaae_stream_t stream);
/* ---- batch norm over a batch that is sharded across ranks (SyncBN) -------------------
 * The reference is single-GPU: tf.nn.moments (utils/tf_util.py:492) sees the WHOLE batch.  When the
 * batch is sharded data-parallel, the `_sync` variants below reproduce that: every rank reduces its rows
 * to per-channel fp64 sums, the HOST-SUPPLIED `allreduce` adds them across ranks (the library does not
 * link a communication library; the host passes RCCL, or anything else, through its own runtime), and
 * the moments / the backward means are taken over count x world rows.  dgamma, dbeta and the bias gradient
 * stay LOCAL sums (the gradient exchange adds them across ranks like every other parameter gradient);
 * the EMA shadows see the global moments, so they stay identical on every rank.
 *   allreduce(ctx, buf, count, stream): sum `count` doubles at device pointer `buf` over all ranks, in
 *     place, ordered after the work already enqueued on `stream` and before whatever is enqueued next;
 *     returns 0 on success.  Called once per forward and once per backward of a layer.
 *   buf: device scratch of at least 2*C doubles owned by the caller (the sums travel in it).
 *   world: number of ranks (every rank contributes the same number of rows).
 * sync == NULL: exactly the plain entry point. */
typedef int (*cloudaae_allreduce_fn)(void *ctx, double *buf, int count, cloudaae_stream_t stream);
typedef struct cloudaae_bn_sync {
    cloudaae_allreduce_fn allreduce;
    void *ctx;
    int world;
    double *buf;
} cloudaae_bn_sync;
/* cloudaae_bn_forward / _colstats (colstats may be NULL) with global moments */
int cloudaae_bn_forward_sync(int M, int C, const float *y, int ldy, const float *gamma, const float *beta,
                             int training, const float *decay, float *ema_mean, float *ema_var,
                             float *save_mean, float *save_var, int relu, float *out, int ldo, int pool_rows,
                             int pool_mode, float *pooled, float *tie_count, double *pool_stats, void *workspace,
                             const double *colstats, int colstats_parts, const cloudaae_bn_sync *sync,
                             cloudaae_stream_t stream);
/* cloudaae_bn_backward with the means of (dz, dz*x_hat) over the global batch */
int cloudaae_bn_backward_sync(int M, int C, const float *y, int ldy, const float *gamma, const float *beta,
                              const float *save_mean, const float *save_var, int training, int relu,
                              const float *dout, int lddo, int pool_rows, int pool_mode, const float *dpooled,
                              const float *pooled, const float *tie_count, float *dy, int lddy, float *dgamma,
                              float *dbeta, float *dbias, int accumulate_param_grads, const double *pool_stats,
                              void *workspace, const cloudaae_bn_sync *sync, cloudaae_stream_t stream);
/* cloudaae_edgeconv_forward / _backward (declared below) with the batch norm of the block over the edges of
 * every rank's clouds: same arguments plus `sync` in front of the stream(s). */
int cloudaae_edgeconv_forward_sync(int b, int n, int k, int cin, int cout, const float *x, int ldx,
                                   const int *nn_idx, const float *weights, const float *biases,
                                   const float *gamma, const float *beta, int training, const float *decay,
                                   float *ema_mean, float *ema_var, int pool_mode, float *pq, float *save_mean,
                                   float *save_var, float *out, int ldo, float *tie_count, float *edge_stats,
                                   int gemm_bf16, void *workspace, const cloudaae_bn_sync *sync,
                                   cloudaae_stream_t stream);
int cloudaae_edgeconv_backward_sync(int b, int n, int k, int cin, int cout, const float *x, int ldx,
                                    const int *nn_idx, const float *weights, const float *biases,
                                    const float *gamma, const float *beta, int training, int pool_mode,
                                    const float *pq, const float *save_mean, const float *save_var,
                                    const float *out, int ldo, const float *tie_count, const float *dout, int lddo,
                                    float *dpq, int *rev_scratch, int rev_ready, float *dx, int lddx,
                                    int accumulate_dx, float *dweights, int dweights_zeroed, float *dbiases,
                                    float *dgamma, float *dbeta, const float *edge_stats, int gemm_bf16,
                                    void *workspace, const cloudaae_bn_sync *sync, cloudaae_stream_t stream,
                                    cloudaae_stream_t side_stream);
/* out[c] (+)= sum_r x[r][c] (bias gradients); workspace as for bn (same C). */
int cloudaae_colsum_f32(int M, int C, const float *x, int ldx, float *out, int accumulate, void *workspace,
                        cloudaae_stream_t stream);

/* ---- a whole fully connected layer at small batch ------------------------ */

/* tf_util.fully_connected (utils/tf_util.py:321-365: tf.matmul :351, bias_add :352, batch_norm_for_fc
 * :355, activation :358) as ONE launch per direction when the rows are the clouds of a batch of at
 * most cloudaae_fc_max_rows() (= 128: four 32-row MFMA tiles): the decoder and pose heads of
 * models/pointnet_ycb_23_decoder_4.py:413-455.  Larger batches take cloudaae_gemm_f32 + cloudaae_bn_*.
 *
 * forward: y[M,N] = x[M,K] w[K,N] + bias (bias may be NULL); with gamma != NULL also the batch norm of
 * y (arguments as cloudaae_bn_forward) into out[M,N], y keeping the pre-normalisation values the
 * backward needs.  gamma == NULL: no batch norm, only y is written.
 * tickets: cloudaae_fc_forward_tickets(M, N) ints holding ZERO, left zero by the call (arrival counters:
 * the product is cut over K -- and over 32-row tiles of the batch -- across workgroups; the last to
 * arrive at a column tile sums the pieces and, with batch norm, normalises the column tile);
 * partials: cloudaae_fc_forward_partials(M, K, N, gamma != NULL) floats of scratch (any contents);
 * partials_floats: the floats behind it (the call fails if this launch's cut needs more -- the cut is derived again
 * at every launch, also from development knobs).
 * The pieces are summed in a FIXED order by the last to arrive: the layer is bit-reproducible from run
 * to run.  With tickets or partials NULL a layer keeps K whole in one workgroup per column tile and
 * row tile (slower; batch norm over more than 32 rows is then refused).
 * Two calls in flight at the same time (different streams) need separate counters and scratch. */
int cloudaae_fc_max_rows(void);
int cloudaae_fc_forward_tickets(int M, int N);
long long cloudaae_fc_forward_partials(int M, int K, int N, int batch_norm);
int cloudaae_fc_forward(int M, int K, int N, const float *x, int ldx, const float *w, const float *bias,
                        const float *gamma, const float *beta, int training, const float *decay,
                        float *ema_mean, float *ema_var, float *save_mean, float *save_var, int relu,
                        float *y, float *out, int *tickets, float *partials,
                        long long partials_floats, cloudaae_stream_t stream);
/* backward of the same layer from dout[M,N] (gradient of `out`, or of y when gamma == NULL):
 *   dx[M,K] += d(y) w^T      (ADDED with fp32 atomics: pass zeros, or a buffer that other consumers
 *                             of x add their gradients to as well; NULL = not wanted)
 *   dw[K,N] (+)= x^T d(y)    (accumulate_dw; NULL = not wanted)
 *   dgamma, dbeta, dbias (+)= as cloudaae_bn_backward (accumulate_param_grads; NULL = not wanted;
 *                             without batch norm dbias = column sums of dout). */
int cloudaae_fc_backward(int M, int K, int N, const float *x, int ldx, const float *w, const float *y,
                         const float *gamma, const float *beta, const float *save_mean,
                         const float *save_var, int training, int relu, const float *dout, int lddo,
                         float *dx, int lddx, float *dw, int accumulate_dw, float *dgamma, float *dbeta,
                         float *dbias, int accumulate_param_grads, cloudaae_stream_t stream);

/* Up to cloudaae_fc_max_group() (= 4) INDEPENDENT layers of the same batch in one launch per
 * direction: the decoder and the two pose heads are three chains of three layers
 * (models/pointnet_ycb_23_decoder_4.py:413-455), so depth by depth they are three launches forward and
 * three backward instead of nine each.  One record per layer, fields as the arguments of
 * cloudaae_fc_forward / cloudaae_fc_backward above (forward reads the first block, backward both).
 * Layers that share their input x may share dx: the gradients of all consumers add up in it. */
typedef struct cloudaae_fc_layer {
    int K, N;
    const float *x;
    int ldx;
    const float *w, *bias;
    const float *gamma, *beta;          /* gamma NULL: no batch norm */
    float *ema_mean, *ema_var, *save_mean, *save_var;
    int relu;
    float *y, *out;
    int *tickets;
    /* backward */
    const float *dout;
    int lddo;
    float *dx;
    int lddx;
    float *dw;
    int accumulate_dw;
    float *dgamma, *dbeta, *dbias;
    int accumulate_param_grads;
    /* forward */
    float *partials;
    long long partials_floats;          /* floats behind `partials` (>= cloudaae_fc_forward_partials at launch time) */
    const float *out_rowvec;            /* gamma NULL only: y[r][c] += out_rowvec[r * out_rowvec_d + c % out_rowvec_d] */
    int out_rowvec_d;                   /* (train_cloudAAE_ycbv.py:232-233: xyz_recon = recon_res + element_mean) */
} cloudaae_fc_layer;
int cloudaae_fc_max_group(void);
int cloudaae_fc_forward_group(int M, int count, const cloudaae_fc_layer *layers, int training,
                              const float *decay, cloudaae_stream_t stream);
int cloudaae_fc_backward_group(int M, int count, const cloudaae_fc_layer *layers, int training,
                               cloudaae_stream_t stream);

/* ---- the DGCNN edge-convolution block, fused ---------------------------- */

/* get_edge_feature + conv2d 1x1 + batch norm + ReLU + pool over k
 * (utils/tf_util.py:635-669,111-179; models/pointnet_ycb_23_decoder_4.py:337-350):
 * x[b*n, cin] (row stride ldx), nn_idx[b,n,k], weights[2*cin, cout] (the TF kernel
 * [1,1,2cin,cout]), pool_mode 1 mean / 2 max -> out[b*n, cout] (row stride ldo).
 * pq[b*n, 2*cout] is scratch that the backward pass reads again.  cout in {64,128}.
 * max pool: tie_count[b*n, cout] receives the number of equal maxima (tf.reduce_max shares
 * the gradient among them); backward then also needs the forward output.
 * backward scratch: dpq[b*n, 2*cout] floats, rev_scratch[b*(n+1) + b*n*k] ints (reverse
 * neighbour lists, built by a counting sort in LDS: no atomics on the gradient tensors).
 * gemm_bf16 != 0: the block's dense products round their operands to bfloat16 (cloudaae_gemm_bf16).
 * dweights_zeroed != 0: the caller already cleared dweights (skips the clear pass of the split-K products).  * edge_stats (optional, used with mean pool in training mode): [b*n][3][cout] floats the forward call
 * fills per point -- edges passing the ReLU, sum of their x_hat, sum of all x_hat -- and the backward
 * call, given the same buffer, turns into its column sums with a streaming pass instead of gathering
 * every neighbour again (24.7 -> 7 us per 64-channel layer at B=32, N=1024).  * pq: [b*n][2*cout] scratch of the call; on return its first cout columns hold U = X W_centre' + b
 * (centre term of every edge of the point) and the last cout columns Q = X W_neighbour, which is how
 * cloudaae_edgeconv_backward expects to find it. */
long long cloudaae_edgeconv_workspace_bytes(int cout);
/* Self-test of the gradient pass's division by the neighbour count (mean pooling, utils/tf_util.py reduce_mean over k:
 * models/pointnet_ycb_23_decoder_4.py:350): the kernels divide by a launch-wide constant d with the IEEE sequence's
 * d-only part (reciprocal and its refinement, scaling of d) computed once -- six or eight instructions per quotient
 * instead of eleven.  Walks ALL 2^32 float numerators x: count[0] = those whose quotient differs in its bits from x / d
 * (two NaNs are equal), count[1] = bits of the largest magnitude among them.  count: 2 x u64, device memory.
 * corrections: 1 or 2 remainder steps; 0 = what cloudaae_edgeconv_backward takes for k = d.  1 <= d <= 2^20. */
int cloudaae_selftest_div_by(float d, int corrections, unsigned long long *count, cloudaae_stream_t stream);
int cloudaae_edgeconv_forward(int b, int n, int k, int cin, int cout, const float *x, int ldx,
                              const int *nn_idx, const float *weights, const float *biases,
                              const float *gamma, const float *beta, int training, const float *decay,
                              float *ema_mean, float *ema_var, int pool_mode, float *pq, float *save_mean,
                              float *save_var, float *out, int ldo, float *tie_count, float *edge_stats,
                              int gemm_bf16, void *workspace, cloudaae_stream_t stream);
/* The same, the pooled output stored once more as bfloat16 (round to nearest even, the conversion of cloudaae_to_bf16):
 * out_bf16 [b*n] rows ldo_bf16 elements apart -- a column slice of the bfloat16 twin of the concat buffer the aggregation
 * product reads when activations are kept in bfloat16 (BASELINE configs[2]): no conversion pass over the concat. */
int cloudaae_edgeconv_forward_b16out(int b, int n, int k, int cin, int cout, const float *x, int ldx,
                                     const int *nn_idx, const float *weights, const float *biases,
                                     const float *gamma, const float *beta, int training, const float *decay,
                                     float *ema_mean, float *ema_var, int pool_mode, float *pq, float *save_mean,
                                     float *save_var, float *out, int ldo, float *tie_count, float *edge_stats,
                                     int gemm_bf16, void *workspace, void *out_bf16, int ldo_bf16,
                                     cloudaae_stream_t stream);
/* Reverse neighbour lists (for every point m: the points that have m among their k neighbours) of up to 8
 * layers in one launch: rev_scratch[i] (b*(n+1) + b*n*k ints, the buffer later passed to
 * cloudaae_edgeconv_backward with rev_ready = 1) from nn_idx[i] ([b,n,k]).  The encoder's layers all have
 * their neighbour lists by the end of the forward pass, so backward builds them together. */
int cloudaae_edgeconv_revlists(int count, int b, int n, int k, const int *const *nn_idx, int *const *rev_scratch,
                               cloudaae_stream_t stream);
int cloudaae_edgeconv_backward(int b, int n, int k, int cin, int cout, const float *x, int ldx,
                               const int *nn_idx, const float *weights, const float *biases,
                               const float *gamma, const float *beta, int training, int pool_mode,
                               const float *pq, const float *save_mean, const float *save_var,
                               const float *out, int ldo, const float *tie_count, const float *dout,
                               int lddo, float *dpq, int *rev_scratch, int rev_ready, float *dx, int lddx,
                               int accumulate_dx, float *dweights, int dweights_zeroed, float *dbiases,
                               float *dgamma, float *dbeta, const float *edge_stats, int gemm_bf16,
                               void *workspace, cloudaae_stream_t stream, cloudaae_stream_t side_stream);

/* ---- train_cloudAAE_ycbv.py:194-273: the step around the network --------- */

/* :206-226  pc[b,n,3+num_class] = [visible[:, :n] + noise - mean, one_hot(class_id)],
 * mean[b,3] over the n noisy points; noisy[b,n,3] optional; visible is [b,p,3], p >= n. */
int cloudaae_input_assemble(int b, int p, int n, int num_class, const float *visible, const float *noise,
                            const long long *class_id, float *pc, float *mean, float *noisy,
                            cloudaae_stream_t stream);
/* The same with the noise of :217 (tf.random.normal, stddev = noise_std) drawn INSIDE the kernel: Philox4x32-10 keyed by
 * `seed`, counter = (cloud, point), stream = draws[0] -- the number of launches that have drawn so far, kept in the two
 * 64-bit device words `draws` ({counter, arrival ticket}, both zero to start with; the last workgroup of a launch
 * advances the counter, so a recorded step replayed with the same arguments draws fresh noise every time and the noise
 * does not depend on the global-step variable).  Same distribution as the reference's generator, not the same stream.
 * draws == NULL: stream 0 at every launch. */
int cloudaae_input_assemble_noise(int b, int p, int n, int num_class, const float *visible, const long long *class_id,
                                  float *pc, float *mean, float *noisy, float noise_std, unsigned long long seed,
                                  unsigned long long *draws, cloudaae_stream_t stream);
/* :232-233  out[b,r,:] = x[b,r,:] + v[b,:] */
int cloudaae_add_rowvec(int b, int r, int d, const float *x, const float *v, float *out,
                        cloudaae_stream_t stream);
int cloudaae_add_f32(long long n, const float *a, const float *b, float *out, cloudaae_stream_t stream);
/* out = a + b*c elementwise (a may be NULL): the VAE reparameterisation, models/...:953 */
int cloudaae_mul_add_f32(long long n, const float *a, const float *b, const float *c, float *out,
                         cloudaae_stream_t stream);
/* get_edge_feature (utils/tf_util.py:635-669; with_center = 0: _wo_center, :672-706), unfused:
 * x[b*n, c] (row stride ldx), nn_idx[b,n,k] -> out[b,n,k,(1+with_center)*c]; and its gradient
 * (dx[b*n, c] is zero-filled here). */
int cloudaae_edge_feature(int b, int n, int k, int c, int with_center, const float *x, int ldx,
                          const int *nn_idx, float *out, cloudaae_stream_t stream);
int cloudaae_edge_feature_grad(int b, int n, int k, int c, int with_center, const float *g, const int *nn_idx,
                               float *dx, cloudaae_stream_t stream);
/* tf.reduce_mean (mode 1) / tf.reduce_max (mode 2) over groups of `rows` consecutive rows of
 * x[groups*rows, c] -> out[groups, c] (+ tie count for max), and the gradient. */
int cloudaae_pool_rows(int groups, int rows, int c, int mode, const float *x, float *out, float *ties,
                       cloudaae_stream_t stream);
int cloudaae_pool_rows_grad(int groups, int rows, int c, int mode, const float *x, const float *out,
                            const float *ties, const float *g, float *dx, cloudaae_stream_t stream);
/* out[i] = scalar[0] * scale (+ add[i]) : gradient of a mean */
int cloudaae_fill_scaled(long long n, const float *scalar, float scale, const float *add, float *out,
                         cloudaae_stream_t stream);
long long cloudaae_mean_workspace_bytes(void);
int cloudaae_mean_f32(long long n, const float *x, float *out, void *workspace, cloudaae_stream_t stream);
/* per[i] = a[i] + b[i] and out = mean(per) in one pass (chamfer_loss.py:13-14); workspace as for the mean. */
int cloudaae_add_mean_f32(long long n, const float *a, const float *b, float *per, float *out, void *workspace,
                          cloudaae_stream_t stream);
/* losses/trans_distance.py:4-9 */
int cloudaae_trans_error(int b, const float *pred, const float *label, float *per, cloudaae_stream_t stream);
int cloudaae_trans_error_grad(int b, const float *pred, const float *label, const float *per,
                              const float *gper, float *dpred, cloudaae_stream_t stream);
/* losses/angular_distance_taylor.py:30-116 in float64: per[b] = geodesic angle,
 * jac[b,3] = d per / d pred, loss = mean (fp32). */
int cloudaae_rotation_error(int b, const float *pred, const double *label, double *per, double *jac,
                            float *loss, cloudaae_stream_t stream);
/* exponential_map (angular_distance_taylor.py:30-66): axag[b,3] f64 -> rot[b,3,3] f64 */
int cloudaae_exponential_map(int b, const double *axag, double *rot, cloudaae_stream_t stream);
int cloudaae_rotation_error_grad(int b, const double *jac, const float *gloss, float *dpred,
                                 cloudaae_stream_t stream);
/* The whole loss tail of train_cloudAAE_ycbv.py:241-268 in one launch: translation error per sample
 * (trans_per[b]) and its mean, SO(3) geodesic error per sample in float64 (rot_per[b], with the
 * Jacobian rot_jac[b,3] w.r.t. rot_pred) and its mean, total = w_xyz*xyz_loss + w_trans*trans_loss +
 * w_rot*rot_loss.  Same arithmetic as cloudaae_trans_error / cloudaae_rotation_error /
 * cloudaae_loss_mix.  _grad: d(total) -> d(xyz_loss), d(trans_pred)[b,3], d(rot_pred)[b,3]. */
int cloudaae_pose_losses(int b, const float *trans_pred, const float *trans_label, const float *rot_pred,
                         const double *rot_label, const float *xyz_loss, float w_xyz, float w_trans, float w_rot,
                         float *trans_per, float *trans_loss, double *rot_per, double *rot_jac, float *rot_loss,
                         float *total, cloudaae_stream_t stream);
int cloudaae_pose_losses_grad(int b, const float *trans_pred, const float *trans_label, const float *trans_per,
                              const double *rot_jac, const float *g_total, float w_xyz, float w_trans,
                              float w_rot, float *d_xyz_loss, float *d_trans_pred, float *d_rot_pred,
                              cloudaae_stream_t stream);
/* :268  total = w0*a + w1*b + w2*c on device scalars */
int cloudaae_loss_mix(const float *a, const float *b, const float *c, float w0, float w1, float w2,
                      float *out, cloudaae_stream_t stream);
int cloudaae_loss_mix_grad(const float *g, float w0, float w1, float w2, float *ga, float *gb, float *gc,
                           cloudaae_stream_t stream);
/* :263-273  tf.train.AdamOptimizer (ApplyAdam form) over a flat buffer; beta powers are
 * device scalars (TF's beta1_power/beta2_power variables), multiplied when advance != 0. */
int cloudaae_adam_tf(long long n, float *param, const float *grad, float *m, float *v, float lr,
                     float beta1, float beta2, float eps, float *beta1_power, float *beta2_power,
                     float grad_scale, int advance, cloudaae_stream_t stream);
/* The same plus the end-of-step bookkeeping, done by the last workgroup of the kernel to finish: beta powers
 * advance, *step += step_inc (the `batch` counter, :192) and, if bn_decay != NULL, the batch-norm decay of the
 * NEXT step = min(bn_clip, 1 - bn_init * bn_rate^floor(step * batch_size / bn_decay_step)) (:194-202, what
 * cloudaae_bn_decay_schedule computes).  ticket: one int holding zero, left zero. */
int cloudaae_adam_tf_step(long long n, float *param, const float *grad, float *m, float *v, float lr, float beta1,
                          float beta2, float eps, float *beta1_power, float *beta2_power, float grad_scale,
                          float *step, float step_inc, float batch_size, float bn_init, float bn_decay_step,
                          float bn_rate, float bn_clip, float *bn_decay, int *ticket, cloudaae_stream_t stream);
int cloudaae_sgd(long long n, float *param, const float *grad, float lr, float grad_scale,
                 cloudaae_stream_t stream);
/* :194-202  out[0] = min(clip, 1 - init * rate^floor(step[0]*batch_size/decay_step)) */
int cloudaae_bn_decay_schedule(const float *step, float batch_size, float init, float decay_step,
                               float rate, float clip, float *out, cloudaae_stream_t stream);
int cloudaae_increment(float *x, float by, cloudaae_stream_t stream);

/* ---- on-line synthesis (train_cloudAAE_ycbv.py:79-117) ------------------------------------ */

/* transform_object_model (train...:88-93): out[b,j,:] = models[class_id[b], j, 0:3] R_b^T + t_b.
 * models [nmodels,npts,6] (xyz|rgb, obj_models.tfrecords), rot [b,3,3] f64 (cloudaae_exponential_map). */
int cloudaae_transform_object_model(int b, int npts, int nmodels, const float *models,
                                    const long long *class_id, const double *rot, const float *trans,
                                    float *out, cloudaae_stream_t stream);
/* get_random_spherical_occluder (utils/generate_occluder.py:38-81): two Gaussian blobs of per_blob
 * points (sigma), centres ~ N(0,wnear/10), N(0,hnear/10), N((near+z)/2,(z-near)/6), z = trans[:,2];
 * occluder [b, 2*per_blob, 3], blobs interleaved row by row as in the reference.  Counter-based RNG
 * (Philox4x32-10) keyed by `seed`: same distribution as tf.random.normal, not the same stream. */
int cloudaae_random_spherical_occluder(int b, int per_blob, const float *trans, float wnear, float hnear,
                                       float near_dist, float sigma, unsigned long long seed, float *occluder,
                                       cloudaae_stream_t stream);
/* sphericalFlip (utils/hidden_point_removal.py:6-24, 51-68): points = concat(a[b,na,3], bpts[b,nb,3])
 * - center; flipped = 2 (R - |p|) p / |p| + p, R = max|p| * 10^param; both outputs are
 * [b, na+nb+1, 3] with a zero last row (the viewpoint).  bpts may be NULL with nb = 0. */
int cloudaae_spherical_flip(int b, int na, const float *a, int nb, const float *bpts, const float *center,
                            float param, float *flipped, float *org, cloudaae_stream_t stream);
/* convexHull / hidden_point_removal (utils/hidden_point_removal.py:27-48): visible points = vertices of
 * conv(flipped[b,n1,3]) minus the two largest vertex indices (the reference's two `[:-1]`); visible
 * [b,n1,3] = org rows of the visible ids (ascending), padded with random re-draws of visible ids;
 * num_vis [b] int64; visible_id [b,n1] (optional; -1 in the padded rows).  qhull is replaced by an exact
 * per-point vertex test (2-variable LPs in fp64: a local problem over the point's neighbours in a spatial
 * order, then verification passes over bounding volumes of 64-point groups; the points of a cloud are handed
 * to waves from a per-cloud queue); workspace: cloudaae_hpr_workspace_bytes(b, n1) (vertex flags, the sorted
 * cloud, its permutation, the queues -- contents undefined afterwards).  The points of a cloud are taken to be
 * distinct (qhull reports one of several identical vertices; here an exact copy of a binding constraint tests
 * as violated by round-off and the answer for such points is unspecified). */
long long cloudaae_hpr_workspace_bytes(int b, int n1);
int cloudaae_hidden_point_removal(int b, int n1, const float *flipped, const float *org,
                                  unsigned long long seed, float *visible, long long *num_vis, int *visible_id,
                                  void *workspace, cloudaae_stream_t stream);
/* The same with a chosen number of output rows: visible [b,rows,3] (visible_id [b,rows]) = the visible points in
 * ascending index, then random re-draws of visible points up to `rows` rows -- the reference's rule
 * (hidden_point_removal.py:38-40, where rows == n1) for a Chamfer target of 4N rows when 4N exceeds the model's
 * point count (BASELINE configs[4]: N = 4096).  row_src [b,rows] (optional): for every output row the row < num_vis it
 * is equal to (itself for the visible points, the drawn one for a re-draw; -1 when nothing is visible) -- what
 * cloudaae_nn_distance_prefix needs to search the distinct target points only. */
int cloudaae_hidden_point_removal_rows(int b, int n1, const float *flipped, const float *org,
                                       unsigned long long seed, int rows, float *visible, long long *num_vis,
                                       int *visible_id, int *row_src, void *workspace, cloudaae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CLOUDAAE_HIP_H */
