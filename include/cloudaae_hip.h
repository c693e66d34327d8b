/*
 * cloudaae_hip.h -- C ABI of libcloudaae_hip.so, the MI355X (gfx950) drop-in for
 * the native hot path of GeeeG/CloudAAE.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer to contiguous row-major memory owned by
 *     the caller (outputs and workspaces included); nothing is allocated here;
 *   - float = IEEE fp32, int = int32; sizes are element counts;
 *   - every function takes the HIP stream to launch on as its last argument
 *     (a hipStream_t passed as void*; NULL = the default stream), is re-entrant,
 *     keeps no global state, never synchronises, and returns 0 or a hipError_t
 *     value (cloudaae_last_error() describes the most recent failure of the
 *     calling thread);
 *   - gradient outputs are zero-filled by the callee.
 * The reference's launchers have C++ linkage, no stream and no status
 * (tf_nndistance.cpp:168,208; tf_sampling.cpp:65,94,125,150); each entry point
 * below names the one it replaces and keeps its argument order.
 */
#ifndef CLOUDAAE_HIP_H
#define CLOUDAAE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void *cloudaae_stream_t; /* hipStream_t */

int cloudaae_version(void);
const char *cloudaae_last_error(void);

/* ---- tf_ops/nn_distance ------------------------------------------------- */

/* NnDistance forward, both directions in one launch.
 * Replaces: void NmDistanceKernelLauncher(int b,int n,const float* xyz,int m,
 *   const float* xyz2,float* result,int* result_i,float* result2,int* result2_i)
 *   (tf_ops/nn_distance/tf_nndistance.cpp:168, tf_nndistance_g.cu:128-131);
 * numerics of the CPU op (tf_nndistance.cpp:21-43): squared L2, un-fused fp32,
 * first minimum wins; m == 0 gives dist 0 / idx 0.
 * xyz1 [b,n,3], xyz2 [b,m,3] -> dist1 [b,n], idx1 [b,n], dist2 [b,m], idx2 [b,m]. */
int cloudaae_nn_distance(int b, int n, const float *xyz1, int m, const float *xyz2, float *dist1,
                         int *idx1, float *dist2, int *idx2, cloudaae_stream_t stream);

/* NnDistanceGrad.
 * Replaces: void NmDistanceGradKernelLauncher(int b,int n,const float* xyz1,int m,
 *   const float* xyz2,const float* grad_dist1,const int* idx1,const float* grad_dist2,
 *   const int* idx2,float* grad_xyz1,float* grad_xyz2)
 *   (tf_nndistance.cpp:208, tf_nndistance_g.cu:152-157; CPU loops tf_nndistance.cpp:126-163).
 * Either gradient output may be NULL (not wanted). */
int cloudaae_nn_distance_grad(int b, int n, const float *xyz1, int m, const float *xyz2,
                              const float *grad_dist1, const int *idx1, const float *grad_dist2,
                              const int *idx2, float *grad_xyz1, float *grad_xyz2,
                              cloudaae_stream_t stream);

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

#ifdef __cplusplus
}
#endif
#endif /* CLOUDAAE_HIP_H */
