// ref_gpu_shim.hip -- TEST INFRASTRUCTURE ONLY.
//
// The reference's own GPU kernels, compiled for gfx950 by hipcc from where they lie under /root/reference, behind a C-ABI shim:
//   tf_ops/sampling/tf_sampling_g.cu       whole file (it includes nothing): cumsumKernel + binarysearchKernel (ProbSample),
//                                          farthestpointsamplingKernel, gatherpointKernel, scatteraddpointKernel and their
//                                          launchers (:198-211)
//   tf_ops/nn_distance/tf_nndistance_g.cu  lines 5-151, line-extracted at build time like the CPU lines of
//                                          ref_nndistance_shim.cpp: NmDistanceKernel, its launcher (:128-131) and
//                                          NmDistanceGradKernel.  NOT its gradient launcher (:152-157): that one calls
//                                          cudaMemset, a CUDA runtime function this image does not have and this file does
//                                          not stand in for; the wrapper below clears the two outputs with hipMemset and
//                                          issues the launcher's two launches (:155-156) itself.
// oracle/build_ref.sh passes the paths in as REF_SAMPLING_CU / REF_NNDISTANCE_INC; no reference source is kept in this
// repository, and without /root/reference this file does not build (the tests that use it skip).  hipcc is not the
// reference's toolchain (nvcc, CUDA 9), so this pins nothing FORMALLY -- what it does is run the reference's own lines: the
// 512-thread strided scan and the `dists[i1] < dists[i2]` tree of the FPS kernel decide its tie-break, the blocked scan of
// cumsumKernel decides ProbSample's prefix sums, and the oracle's restatements of both (oracle/cloudaae_oracle.c) and the
// product kernels are compared with them bit for bit (tests/test_11_reference_kernels_gpu.py).  Built with -ffp-contract=off:
// the arithmetic the oracle defines (the un-fused expression of the source; nvcc's default would contract it).
// The launchers use the legacy default stream (`<<<grid, block>>>`, as the reference does); every wrapper synchronises.
#if !defined(REF_SAMPLING_CU) || !defined(REF_NNDISTANCE_INC)
#error "build with oracle/build_ref.sh"
#endif
#include <hip/hip_runtime.h>

#include REF_SAMPLING_CU
#include REF_NNDISTANCE_INC

#define REF_API extern "C" __attribute__((visibility("default")))

static int ref_done() { return (int)hipDeviceSynchronize(); }

// device pointers throughout; temp: 32 * n floats (tf_sampling.cpp:115)
REF_API int ref_gpu_farthest_point_sample(int b, int n, int m, const float *inp, float *temp, int *out)
{
    farthestpointsamplingLauncher(b, n, m, inp, temp, out);
    return ref_done();
}
REF_API int ref_gpu_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out)
{
    gatherpointLauncher(b, n, m, inp, idx, out);
    return ref_done();
}
// inp_g must be zero on entry (the reference's op clears it: tf_sampling.cpp:174)
REF_API int ref_gpu_scatter_add_point(int b, int n, int m, const float *out_g, const int *idx, float *inp_g)
{
    scatteraddpointLauncher(b, n, m, out_g, idx, inp_g);
    return ref_done();
}
// temp: b * n floats (tf_sampling.cpp:86)
REF_API int ref_gpu_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out)
{
    probsampleLauncher(b, n, m, inp_p, inp_r, temp, out);
    return ref_done();
}
REF_API int ref_gpu_nn_distance(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int *result_i,
                                float *result2, int *result2_i)
{
    NmDistanceKernelLauncher(b, n, xyz, m, xyz2, result, result_i, result2, result2_i);
    return ref_done();
}
REF_API int ref_gpu_nn_distance_grad(int b, int n, const float *xyz1, int m, const float *xyz2, const float *grad_dist1,
                                     const int *idx1, const float *grad_dist2, const int *idx2, float *grad_xyz1,
                                     float *grad_xyz2)
{
    // tf_nndistance_g.cu:153-156 (the zero-fill through this platform's runtime, the two launches as written there)
    if (hipMemset(grad_xyz1, 0, (size_t)b * n * 3 * 4) != hipSuccess || hipMemset(grad_xyz2, 0, (size_t)b * m * 3 * 4) != hipSuccess)
        return -1;
    NmDistanceGradKernel<<<dim3(1, 16, 1), 256>>>(b, n, xyz1, m, xyz2, grad_dist1, idx1, grad_xyz1, grad_xyz2);
    NmDistanceGradKernel<<<dim3(1, 16, 1), 256>>>(b, m, xyz2, n, xyz1, grad_dist2, idx2, grad_xyz2, grad_xyz1);
    return ref_done();
}
