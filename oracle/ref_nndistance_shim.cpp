// ref_nndistance_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// C-ABI shim around the reference's own dependency-free Chamfer lines.  The
// reference translation unit tf_ops/nn_distance/tf_nndistance.cpp as a whole
// cannot be built here (it includes TensorFlow headers that this image lacks);
// the two pieces below need nothing but the C++ core language:
//   * tf_nndistance.cpp:21-43   static void nnsearch(...)           (forward)
//   * tf_nndistance.cpp:126-163 body of NnDistanceGradOp::Compute    (backward:
//     the two zero-fill loops and the two accumulation sweeps)
// oracle/build_ref.sh line-extracts them from /root/reference at build time into
// oracle/_ref/ (git-ignored, deleted again after the compile) and passes the
// paths in through REF_NNSEARCH_INC / REF_NNGRAD_INC.  No reference source is
// kept in this repository; without /root/reference this file does not build and
// the oracle falls back to its own restatement (oracle/cloudaae_oracle.c).
#ifndef REF_NNSEARCH_INC
#error "build with oracle/build_ref.sh"
#endif

#include REF_NNSEARCH_INC

extern "C" __attribute__((visibility("default"))) void
ref_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1, int *idx1,
                float *dist2, int *idx2)
{
    // call pattern of tf_nndistance.cpp:79-80
    nnsearch(b, n, m, xyz1, xyz2, dist1, idx1);
    nnsearch(b, m, n, xyz2, xyz1, dist2, idx2);
}

extern "C" __attribute__((visibility("default"))) void
ref_nn_distance_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                     const float *grad_dist1, const int *idx1, const float *grad_dist2,
                     const int *idx2, float *grad_xyz1, float *grad_xyz2)
{
    // the extracted statements use exactly these local names
#include REF_NNGRAD_INC
}
