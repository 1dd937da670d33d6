// Hessian running mean  H <- H*n/(n+b) + 2/(n+b) * X^T X   (replaces gptq.py:38-65).
//
// X [tokens, C] arrives in the model dtype (fp16/bf16/fp32) and is widened to fp32
// on load (gptq.py:62).  The product is an exact-fp32 MFMA SYRK: only the upper
// triangle tiles (ti <= tj) are formed -- H is symmetric -- which halves the
// dominant FLOP term of the whole pipeline; gptq_symmetrize mirrors it on demand.
#include "gemm_f32.h"

namespace gptq {

template <typename T>
__global__ __launch_bounds__(GEMM_THREADS) void hessian_kernel(float* __restrict__ H, int ldh,
                                                               const T* __restrict__ X, int ldx, int C,
                                                               int tokens, float alpha, float beta,
                                                               bool vec) {
  // upper-triangle tile pair from a linear id (row-major over ti <= tj)
  const int nt = (C + GBM - 1) / GBM;
  int rest = blockIdx.x, ti = 0;
  while (rest >= nt - ti) { rest -= nt - ti; ++ti; }
  const int tj = ti + rest;

  __shared__ __attribute__((aligned(16))) float smem[GEMM_LDS_FLOATS];
  Operand<T> a{X + (long)ti * GBM, 1, ldx, min(GBM, C - ti * GBM), vec};
  Operand<T> b{X + (long)tj * GBN, 1, ldx, min(GBN, C - tj * GBN), vec};
  float* Ht = H + (long)ti * GBM * ldh + (long)tj * GBN;
  const bool diag = ti == tj;
  gemm_tile<T, T, false, false>(a, b, 0, tokens, smem, [=](int r, int c, float v) {
    if (diag && r > c) return;
    float* h = Ht + (long)r * ldh + c;
    *h = alpha * *h + beta * v;   // contraction is off: fl(fl(alpha*h) + fl(beta*v))
  });
}

// A[r][c] = A[c][r] for r > c, through a 32x33 LDS tile so both sides stay coalesced.
__global__ __launch_bounds__(256) void symmetrize_kernel(float* __restrict__ A, int lda, int n) {
  __shared__ float t[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;   // source tile rows bi*32, cols bj*32 (upper: bi <= bj)
  if (bi > bj) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = bi * 32 + ty + 8 * k, c = bj * 32 + tx;
    t[ty + 8 * k][tx] = (r < n && c < n) ? A[(long)r * lda + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = bj * 32 + ty + 8 * k, c = bi * 32 + tx;   // destination (lower) element
    if (r < n && c < n && r > c) A[(long)r * lda + c] = t[tx][ty + 8 * k];
  }
}

}  // namespace gptq

using namespace gptq;

extern "C" int gptq_hessian_accum(float* H, int ldh, const void* X, int x_dtype, int ldx, int C,
                                  int tokens, int nsamples_before, int batch, gptq_stream_t stream) {
  GPTQ_CHECK_ARG(H && X, "gptq_hessian_accum: null pointer");
  GPTQ_CHECK_ARG(C > 0 && tokens > 0 && batch > 0 && nsamples_before >= 0, "gptq_hessian_accum: bad sizes");
  GPTQ_CHECK_ARG(ldh >= C && ldx >= C, "gptq_hessian_accum: leading dimension smaller than C");
  const int n_after = nsamples_before + batch;
  const float alpha = (float)((double)nsamples_before / (double)n_after);   // gptq.py:59
  const float beta = (float)(2.0 / (double)n_after);                        // gptq.py:62 squared
  const int nt = cdiv(C, GBM);
  const int blocks = nt * (nt + 1) / 2;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (x_dtype) {
    case GPTQ_F32: {
      const float* x = static_cast<const float*>(X);
      hessian_kernel<float><<<blocks, GEMM_THREADS, 0, s>>>(H, ldh, x, ldx, C, tokens, alpha, beta, vec_ok(x, ldx));
      break;
    }
    case GPTQ_F16: {
      const __half* x = static_cast<const __half*>(X);
      hessian_kernel<__half><<<blocks, GEMM_THREADS, 0, s>>>(H, ldh, x, ldx, C, tokens, alpha, beta, vec_ok(x, ldx));
      break;
    }
    case GPTQ_BF16: {
      const __hip_bfloat16* x = static_cast<const __hip_bfloat16*>(X);
      hessian_kernel<__hip_bfloat16><<<blocks, GEMM_THREADS, 0, s>>>(H, ldh, x, ldx, C, tokens, alpha, beta, vec_ok(x, ldx));
      break;
    }
    default:
      GPTQ_CHECK_ARG(false, "gptq_hessian_accum: unknown dtype %d", x_dtype);
  }
  GPTQ_CHECK_LAUNCH("hessian_kernel");
  return GPTQ_OK;
}

extern "C" int gptq_symmetrize(float* A, int lda, int n, gptq_stream_t stream) {
  GPTQ_CHECK_ARG(A && n > 0 && lda >= n, "gptq_symmetrize: bad arguments");
  const int nt = cdiv(n, 32);
  symmetrize_kernel<<<dim3(nt, nt), 256, 0, static_cast<hipStream_t>(stream)>>>(A, lda, n);
  GPTQ_CHECK_LAUNCH("symmetrize_kernel");
  return GPTQ_OK;
}
