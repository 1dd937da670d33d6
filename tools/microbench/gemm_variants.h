// variants under evaluation for tools/microbench/gemm_f32_bench.hip
#pragma once
#include "../../gptq_amd/csrc/gemm2_f32.h"
namespace gptq {
template <bool NT>
__global__ __launch_bounds__(GEMM_THREADS) void small_kernel(float* C, int ldc, const float* A, int lda, const float* B,
                                                             int ldb, int M, int N, int K, int mode) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM64_LDS_FLOATS];
  const long r0 = (long)blockIdx.y * SBM, c0 = (long)blockIdx.x * SBN;
  Operand<float> a{A + r0 * lda, lda, 1, (int)min((long)SBM, M - r0), true};
  Operand<float> b = NT ? Operand<float>{B + c0 * ldb, ldb, 1, (int)min((long)SBN, N - c0), true}
                        : Operand<float>{B + c0, 1, ldb, (int)min((long)SBN, N - c0), true};
  gemm_tile64<float, float, true, NT>(a, b, 0, K, smem, Epilogue{C + r0 * ldc + c0, ldc, 1, mode, TRI_ALL, 0.f, 0.f});
}
constexpr int GEMM2_VARIANTS = 1;
inline const char* gemm2_name(int) { return "gemm_tile64 64x64"; }
inline void gemm2_launch(int, float* C, int ldc, const float* A, int lda, const float* B, int ldb, int M, int N, int K,
                         bool nt, int mode, hipStream_t s) {
  dim3 grid((N + SBN - 1) / SBN, (M + SBM - 1) / SBM);
  if (nt) small_kernel<true><<<grid, GEMM_THREADS, 0, s>>>(C, ldc, A, lda, B, ldb, M, N, K, mode);
  else small_kernel<false><<<grid, GEMM_THREADS, 0, s>>>(C, ldc, A, lda, B, ldb, M, N, K, mode);
}
}  // namespace gptq
