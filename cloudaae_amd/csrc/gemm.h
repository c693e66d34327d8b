// gemm.h -- internal (not part of the C ABI): the GEMM launchers with folded-operand addressing,
// shared by gemm.hip / gemm_bf16.hip and edgeconv.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace cloudaae {

// C[M,N] (+)= op(A) op(B) like cloudaae_gemm_f32 / cloudaae_gemm_bf16, plus:
// fold_b / fold_c = 0, or the power-of-two width at which the logical columns of B's / C's row-major
// storage fold into stacked row blocks: logical (r, c) -> physical row (c / width) * rows + r, column
// c % width, leading dimension == width.  With width = cout the edge convolution's [2*cin, cout]
// kernel IS the [cin, 2*cout] matrix [W_centre | W_neighbour].
// colstats (optional): per row tile the column sums and sums of squares of C.
int gemm_f32_launch(const char *name, int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                    const float *B, int ldb, float *C, int ldc, const float *bias, int accumulate, int fold_b,
                    int fold_c, hipStream_t stream, double *colstats = nullptr, float *ordered_ws = nullptr);
int gemm_bf16_launch(const char *name, int trans_a, int trans_b, int M, int N, int K, const float *A, int lda,
                     const float *B, int ldb, float *C, int ldc, const float *bias, int accumulate, int fold_b,
                     int fold_c, hipStream_t stream, double *colstats = nullptr, float *ordered_ws = nullptr);
// ordered_ws (optional): room for splits * M * N floats; a product cut over K then keeps its slices apart and sums
// them in slice order (bit-reproducible) instead of adding them with atomics.
int gemm_slices_sum(const char *name, int M, int N, int splits, const float *ws, float *C, int ldc, const float *bias,
                    hipStream_t stream, int fold_shift = -1, int fold_rows = 0);

} // namespace cloudaae
