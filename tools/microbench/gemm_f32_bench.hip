// Microbenchmark of the exact-fp32 MFMA GEMM tiles behind the chain / trailing kernels (diagnostic tool, not product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o gemm_f32_bench gemm_f32_bench.hip
// Shapes: C[M x N] (-)= A[M x K] * B, A k-contiguous ([M][K] row-major); B either [K][N] row-major ("nn", the trailing
// update W -= Err * U) or [N][K] row-major ("nt", SYRK / panel products).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <cmath>

#include "../../gptq_amd/csrc/gemm_f32.h"
#include "gemm_variants.h"

namespace gptq { void set_error(const char*, ...) {} }
using namespace gptq;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---- baseline: the product's gemm_tile -------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(GEMM_THREADS) void base_kernel(float* C, int ldc, const float* A, int lda, const float* B,
                                                            int ldb, int M, int N, int K, int mode) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM_LDS_FLOATS];
  const long r0 = (long)blockIdx.y * GBM, c0 = (long)blockIdx.x * GBN;
  Operand<float> a{A + r0 * lda, lda, 1, (int)min((long)GBM, M - r0), true};
  Operand<float> b = NT ? Operand<float>{B + c0 * ldb, ldb, 1, (int)min((long)GBN, N - c0), true}
                        : Operand<float>{B + c0, 1, ldb, (int)min((long)GBN, N - c0), true};
  gemm_tile<float, float, true, NT>(a, b, 0, K, smem, Epilogue{C + r0 * ldc + c0, ldc, 1, mode, TRI_ALL, 0.f, 0.f});
}

static void fill(std::vector<float>& v, unsigned seed) {
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
}

int main(int argc, char** argv) {
  struct Shape { int M, N, K; bool nt; int mode; const char* what; };
  std::vector<Shape> shapes = {
      {4096, 128, 128, false, EPI_SUB, "near-next R=4096 (32 tiles)"},
      {4096, 384, 128, false, EPI_SUB, "near R=4096 N=384"},
      {12288, 256, 128, false, EPI_SUB, "near R=12288 N=256"},
      {22016, 256, 128, false, EPI_SUB, "near R=22016 N=256"},
      {4096, 512, 512, false, EPI_SUB, "far-next R=4096 (K=512,N=512)"},
      {3968, 128, 128, true, EPI_STORE, "panel 31 tiles (nt, store)"},
      {10880, 128, 128, true, EPI_STORE, "panel 85 tiles (nt, store)"},
      {3968, 384, 128, true, EPI_SUB, "near-syrk 31x3 tiles"},
      {10880, 384, 128, true, EPI_SUB, "near-syrk 85x3 tiles"},
      {12288, 3584, 128, false, EPI_SUB, "trailing qkv (rank-128, nn, RMW)"},
      {22016, 2048, 128, false, EPI_SUB, "trailing gate/up (rank-128, nn, RMW)"},
      {4096, 8192, 128, false, EPI_SUB, "trailing down (rank-128, nn, RMW)"},
      {12288, 3584, 512, false, EPI_SUB, "trailing qkv (rank-512, nn, RMW)"},
      {4096, 8192, 512, false, EPI_SUB, "trailing down (rank-512, nn, RMW)"},
      {8192, 8192, 128, true, EPI_SUB, "syrk-like (rank-128, nt, RMW)"},
      {8192, 8192, 512, true, EPI_SUB, "syrk-like (rank-512, nt, RMW)"},
      {4096, 4096, 4096, true, EPI_STORE, "long-K nt store"},
      {4096, 4096, 4096, false, EPI_STORE, "long-K nn store"},
  };
  const char* only = argc > 1 ? argv[1] : nullptr;
  for (const Shape& sh : shapes) {
    const int M = sh.M, N = sh.N, K = sh.K;
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N), hC((size_t)M * N);
    fill(hA, 1); fill(hB, 2); fill(hC, 3);
    float *A, *B, *C, *C0;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&C, hC.size() * 4)); CK(hipMalloc(&C0, hC.size() * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(C0, hC.data(), hC.size() * 4, hipMemcpyHostToDevice));
    const int ldb = sh.nt ? K : N;
    const double flops = 2.0 * M * N * K;
    printf("== %s: M %d N %d K %d\n", sh.what, M, N, K);
    std::vector<float> ref((size_t)M * N), got((size_t)M * N);
    for (int variant = 0; variant < 1 + GEMM2_VARIANTS; ++variant) {
      auto launch = [&](hipStream_t s) {
        if (variant == 0) {
          dim3 grid((N + GBN - 1) / GBN, (M + GBM - 1) / GBM);
          if (sh.nt) base_kernel<true><<<grid, GEMM_THREADS, 0, s>>>(C, N, A, K, B, ldb, M, N, K, sh.mode);
          else base_kernel<false><<<grid, GEMM_THREADS, 0, s>>>(C, N, A, K, B, ldb, M, N, K, sh.mode);
        } else {
          gemm2_launch(variant - 1, C, N, A, K, B, ldb, M, N, K, sh.nt, sh.mode, s);
        }
      };
      if (only && variant > 0 && atoi(only) != variant) continue;
      CK(hipMemcpy(C, C0, hC.size() * 4, hipMemcpyDeviceToDevice));
      launch(0);
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      CK(hipMemcpy(got.data(), C, got.size() * 4, hipMemcpyDeviceToHost));
      double maxdiff = 0;
      if (variant == 0) ref = got;
      else for (size_t i = 0; i < got.size(); i += 97) maxdiff = fmax(maxdiff, fabs((double)got[i] - ref[i]));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int i = 0; i < 3; ++i) launch(0);
      CK(hipEventRecord(e0, 0));
      const int reps = 20;
      for (int i = 0; i < reps; ++i) launch(0);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      printf("   %-34s %9.1f us  %7.1f TFLOP/s  %6.2f TB/s(C rmw)  maxdiff %.2e\n",
             variant == 0 ? "base gemm_tile 128x128" : gemm2_name(variant - 1), us, flops / us / 1e6,
             (sh.mode == EPI_SUB ? 8.0 : 4.0) * M * N / us / 1e6, maxdiff);
      fflush(stdout);
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(C0));
  }
  return 0;
}
