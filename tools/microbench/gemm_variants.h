// variants under evaluation for tools/microbench/gemm_f32_bench.hip
#pragma once
#include "../../gptq_amd/csrc/gemm2_f32.h"
namespace gptq {
// The product's 128 x 128 tile with parts of its stage loop removed (TIMING ONLY, wrong results) or rearranged:
//   ABL 1: no global loads after the first stage      2: no LDS stores / barriers after the first stage      3: both
//   ABL 4: operand registers of TWO stages (stage kt + 2 in flight while kt + 1 is stored); results exact
template <typename TA, typename TB, bool AKC, bool BKC, int ABL>
__device__ __forceinline__ void gemm_tile_abl(const Operand<TA>& a, const Operand<TB>& b, int k_begin, int k_end,
                                              float* smem, const Epilogue& ep) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float* As = smem;
  float* Bs = smem + 2 * GBK * GLD;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float ra[2][GST][4], rb[2][GST][4];
  const int nk = (k_end - k_begin + GBK - 1) / GBK;
  float old[2][2][16];
  tile_load_old(old, ep, a.rem, b.rem, wm, wn, lane);
  stage_load<TA, AKC>(a, k_begin, k_end, ra[0]);
  stage_load<TB, BKC>(b, k_begin, k_end, rb[0]);
  stage_store<AKC>(As, ra[0]);
  stage_store<BKC>(Bs, rb[0]);
  if (ABL == 4 && nk > 1) {
    stage_load<TA, AKC>(a, k_begin + GBK, k_end, ra[1]);
    stage_load<TB, BKC>(b, k_begin + GBK, k_end, rb[1]);
  }
  __syncthreads();
  int cur = 0;
#pragma unroll 2
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    const int rs = (ABL == 4) ? (kt & 1) : 0;                     // register set that receives this iteration's loads
    if (ABL == 4) {
      if (kt + 2 < nk) {
        stage_load<TA, AKC>(a, k_begin + (kt + 2) * GBK, k_end, ra[rs]);
        stage_load<TB, BKC>(b, k_begin + (kt + 2) * GBK, k_end, rb[rs]);
      }
    } else if (more && !(ABL & 1)) {
      stage_load<TA, AKC>(a, k_begin + (kt + 1) * GBK, k_end, ra[0]);
      stage_load<TB, BKC>(b, k_begin + (kt + 1) * GBK, k_end, rb[0]);
    }
    constexpr int LDA = LdsStride<AKC>::v, LDB = LdsStride<BKC>::v;
    const float* Ac = As + cur * GBK * GLD + wm * 64 + (lane & 31);
    const float* Bc = Bs + cur * GBK * GLD + wn * 64 + (lane & 31);
    const int kq = lane >> 5;
    float a0 = Ac[kq * LDA], a1 = Ac[kq * LDA + 32];
    float b0 = Bc[kq * LDB], b1 = Bc[kq * LDB + 32];
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      const int kn = (kk + 2 < GBK ? kk + 2 : kk) + kq;
      const float na0 = Ac[kn * LDA], na1 = Ac[kn * LDA + 32];
      const float nb0 = Bc[kn * LDB], nb1 = Bc[kn * LDB + 32];
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    if (ABL == 4) {
      if (more) {                                                  // stage kt + 1 sits in the OTHER register set
        // (with the 2-unrolled loop kt & 1 is a compile-time constant per copy: no dynamic register indexing)
        stage_store<AKC>(As + (cur ^ 1) * GBK * GLD, ra[rs ^ 1]);
        stage_store<BKC>(Bs + (cur ^ 1) * GBK * GLD, rb[rs ^ 1]);
      }
      __syncthreads();
      cur ^= 1;
    } else if (!(ABL & 2)) {
      if (more) {
        stage_store<AKC>(As + (cur ^ 1) * GBK * GLD, ra[0]);
        stage_store<BKC>(Bs + (cur ^ 1) * GBK * GLD, rb[0]);
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  tile_finish(acc, old, ep, a.rem, b.rem, wm, wn, lane);
}
template <bool NT, int ABL>
__global__ __launch_bounds__(GEMM_THREADS) void abl_kernel(float* C, int ldc, const float* A, int lda, const float* B,
                                                           int ldb, int M, int N, int K, int mode) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM_LDS_FLOATS];
  const long r0 = (long)blockIdx.y * GBM, c0 = (long)blockIdx.x * GBN;
  Operand<float> a{A + r0 * lda, lda, 1, (int)min((long)GBM, M - r0), true};
  Operand<float> b = NT ? Operand<float>{B + c0 * ldb, ldb, 1, (int)min((long)GBN, N - c0), true}
                        : Operand<float>{B + c0, 1, ldb, (int)min((long)GBN, N - c0), true};
  gemm_tile_abl<float, float, true, NT, ABL>(a, b, 0, K, smem, Epilogue{C + r0 * ldc + c0, ldc, 1, mode, TRI_ALL, 0.f, 0.f});
}

// The product's tile with the next stage's LDS stores INSIDE the k loop (after k-pair `AT` of 16) instead of behind it:
// the store phase overlaps the last MFMAs of the stage and only the barrier is left between stages.  Exact.
template <typename TA, typename TB, bool AKC, bool BKC, int AT, bool PRE>
__device__ __forceinline__ void gemm_tile_early(const Operand<TA>& a, const Operand<TB>& b, int k_begin, int k_end,
                                                float* smem, const Epilogue& ep) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float* As = smem;
  float* Bs = smem + 2 * GBK * GLD;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float ra[GST][4], rb[GST][4];
  const int nk = (k_end - k_begin + GBK - 1) / GBK;
  float old[2][2][16];
  if (PRE) tile_load_old(old, ep, a.rem, b.rem, wm, wn, lane);
  if (nk > 0) {
    stage_load<TA, AKC>(a, k_begin, k_end, ra);
    stage_load<TB, BKC>(b, k_begin, k_end, rb);
    stage_store<AKC>(As, ra);
    stage_store<BKC>(Bs, rb);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      stage_load<TA, AKC>(a, k_begin + (kt + 1) * GBK, k_end, ra);
      stage_load<TB, BKC>(b, k_begin + (kt + 1) * GBK, k_end, rb);
    }
    constexpr int LDA = LdsStride<AKC>::v, LDB = LdsStride<BKC>::v;
    const float* Ac = As + cur * GBK * GLD + wm * 64 + (lane & 31);
    const float* Bc = Bs + cur * GBK * GLD + wn * 64 + (lane & 31);
    const int kq = lane >> 5;
    float a0 = Ac[kq * LDA], a1 = Ac[kq * LDA + 32];
    float b0 = Bc[kq * LDB], b1 = Bc[kq * LDB + 32];
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      const int kn = (kk + 2 < GBK ? kk + 2 : kk) + kq;
      const float na0 = Ac[kn * LDA], na1 = Ac[kn * LDA + 32];
      const float nb0 = Bc[kn * LDB], nb1 = Bc[kn * LDB + 32];
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
      if (kk == 2 * AT && more) {                                  // the other buffer: nobody reads it in this stage
        __builtin_amdgcn_sched_barrier(0);
        stage_store<AKC>(As + (cur ^ 1) * GBK * GLD, ra);
        stage_store<BKC>(Bs + (cur ^ 1) * GBK * GLD, rb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  if (PRE) tile_finish(acc, old, ep, a.rem, b.rem, wm, wn, lane);
  else tile_epilogue(acc, ep, a.rem, b.rem, wm, wn, lane);
}
template <bool NT, int AT, bool PRE>
__global__ __launch_bounds__(GEMM_THREADS) void early_kernel(float* C, int ldc, const float* A, int lda, const float* B,
                                                             int ldb, int M, int N, int K, int mode) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM_LDS_FLOATS];
  const long r0 = (long)blockIdx.y * GBM, c0 = (long)blockIdx.x * GBN;
  Operand<float> a{A + r0 * lda, lda, 1, (int)min((long)GBM, M - r0), true};
  Operand<float> b = NT ? Operand<float>{B + c0 * ldb, ldb, 1, (int)min((long)GBN, N - c0), true}
                        : Operand<float>{B + c0, 1, ldb, (int)min((long)GBN, N - c0), true};
  gemm_tile_early<float, float, true, NT, AT, PRE>(a, b, 0, K, smem, Epilogue{C + r0 * ldc + c0, ldc, 1, mode, TRI_ALL, 0.f, 0.f});
}
template <int AT, bool PRE>
inline void early_launch(float* C, int ldc, const float* A, int lda, const float* B, int ldb, int M, int N, int K, bool nt,
                         int mode, hipStream_t s) {
  dim3 grid((N + GBN - 1) / GBN, (M + GBM - 1) / GBM);
  if (nt) early_kernel<true, AT, PRE><<<grid, GEMM_THREADS, 0, s>>>(C, ldc, A, lda, B, ldb, M, N, K, mode);
  else early_kernel<false, AT, PRE><<<grid, GEMM_THREADS, 0, s>>>(C, ldc, A, lda, B, ldb, M, N, K, mode);
}

// ---- three-deep LDS ring: no bubble at the stage barrier ------------------------------------------------------------
// Stage kt + 2 is stored (from registers loaded one stage earlier) in the MIDDLE of stage kt's MFMAs, into the buffer that
// stage kt - 1 used; the first fragments of stage kt + 1 are read BEFORE the barrier that ends stage kt (that buffer has
// been complete since the barrier before).  One barrier per stage remains, with nothing but arrival skew behind it, and
// it waits for LDS only (the global loads of stage kt + 3 stay in flight across it).  Exact (same ascending k order).
template <int BK> struct Ring3 {
  static constexpr int PCS = BK / 8;                              // 4-element pieces per thread per operand stage
  static constexpr int BUF = 2 * BK * GLD;                        // floats per ring slot (A + B)
  static constexpr size_t LDS_BYTES = sizeof(float) * 3 * BUF;
};
template <typename T, bool KC, int BK>
__device__ __forceinline__ void r3_load(const Operand<T>& o, int k0, float (&r)[BK / 8][4]) {
  const int tid = threadIdx.x;
  if (!KC) {
    const int m4 = (tid & 31) * 4;
#pragma unroll
    for (int h = 0; h < BK / 8; ++h) load4_vec<T>(o.p + (long)m4 * o.st + (long)(k0 + (tid >> 5) + 8 * h) * o.sk, r[h]);
  } else {
    constexpr int KTH = BK / 4, RPP = 256 / KTH;                  // threads per tile row, rows per pass
    const int k = k0 + (tid % KTH) * 4;
#pragma unroll
    for (int h = 0; h < BK / 8; ++h) load4_vec<T>(o.p + (long)(tid / KTH + RPP * h) * o.st + (long)k * o.sk, r[h]);
  }
}
template <bool KC, int BK>
__device__ __forceinline__ void r3_store(float* S, const float (&r)[BK / 8][4]) {
  constexpr int LDS_ = LdsStride<KC>::v;
  const int tid = threadIdx.x;
  if (!KC) {
    const int m4 = (tid & 31) * 4;
#pragma unroll
    for (int h = 0; h < BK / 8; ++h)
      *reinterpret_cast<float4*>(S + ((tid >> 5) + 8 * h) * LDS_ + m4) = make_float4(r[h][0], r[h][1], r[h][2], r[h][3]);
  } else {
    constexpr int KTH = BK / 4, RPP = 256 / KTH;
    const int k = (tid % KTH) * 4;
#pragma unroll
    for (int h = 0; h < BK / 8; ++h) {
      const int m = tid / KTH + RPP * h;
#pragma unroll
      for (int e = 0; e < 4; ++e) S[(k + e) * LDS_ + m] = r[h][e];
    }
  }
}
// full tiles only (rem == 128, vec, K a multiple of BK): what the far updates bring
template <typename TA, typename TB, bool AKC, bool BKC, int BK>
__device__ __forceinline__ void gemm_tile_ring3(const Operand<TA>& a, const Operand<TB>& b, int k_begin, int k_end,
                                                float* smem, const Epilogue& ep) {
  constexpr int BUF = Ring3<BK>::BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, kq = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float ra[BK / 8][4], rb[BK / 8][4];
  const int nk = (k_end - k_begin) / BK;
  r3_load<TA, AKC, BK>(a, k_begin, ra);
  r3_load<TB, BKC, BK>(b, k_begin, rb);
  r3_store<AKC, BK>(smem, ra);
  r3_store<BKC, BK>(smem + BK * GLD, rb);
  if (nk > 1) {
    r3_load<TA, AKC, BK>(a, k_begin + BK, ra);
    r3_load<TB, BKC, BK>(b, k_begin + BK, rb);
    r3_store<AKC, BK>(smem + BUF, ra);
    r3_store<BKC, BK>(smem + BUF + BK * GLD, rb);
  }
  if (nk > 2) {
    r3_load<TA, AKC, BK>(a, k_begin + 2 * BK, ra);
    r3_load<TB, BKC, BK>(b, k_begin + 2 * BK, rb);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  constexpr int LDA = LdsStride<AKC>::v, LDB = LdsStride<BKC>::v;
  const int aoff = wm * 64 + (lane & 31), boff = BK * GLD + wn * 64 + (lane & 31);
  float a0 = smem[aoff + kq * LDA], a1 = smem[aoff + kq * LDA + 32];
  float b0 = smem[boff + kq * LDB], b1 = smem[boff + kq * LDB + 32];
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const float* Ac = smem + cur * BUF + aoff;
    const float* Bc = smem + cur * BUF + boff;
    const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;   // ring slots of stages kt + 1, kt + 2
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float na0, na1, nb0, nb1;
      if (kk + 2 < BK) {
        const int kn = kk + 2 + kq;
        na0 = Ac[kn * LDA]; na1 = Ac[kn * LDA + 32];
        nb0 = Bc[kn * LDB]; nb1 = Bc[kn * LDB + 32];
      } else {                                                     // the first fragments of stage kt + 1 (complete since the
        const float* An = smem + nxt * BUF + aoff;                 // previous barrier); a harmless read behind the last stage
        const float* Bn = smem + nxt * BUF + boff;
        na0 = An[kq * LDA]; na1 = An[kq * LDA + 32];
        nb0 = Bn[kq * LDB]; nb1 = Bn[kq * LDB + 32];
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
      if (kk == BK / 2 - 2) {                                      // the middle of the stage
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) {
          r3_store<AKC, BK>(smem + nn * BUF, ra);
          r3_store<BKC, BK>(smem + nn * BUF + BK * GLD, rb);
        }
        if (kt + 3 < nk) {
          r3_load<TA, AKC, BK>(a, k_begin + (kt + 3) * BK, ra);
          r3_load<TB, BKC, BK>(b, k_begin + (kt + 3) * BK, rb);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    cur = nxt;
  }
  tile_epilogue(acc, ep, a.rem, b.rem, wm, wn, lane);
}
template <bool NT, int BK>
__global__ __launch_bounds__(GEMM_THREADS) void ring3_kernel(float* C, int ldc, const float* A, int lda, const float* B,
                                                             int ldb, int M, int N, int K, int mode) {
  extern __shared__ __attribute__((aligned(16))) float dyn_smem[];
  const long r0 = (long)blockIdx.y * GBM, c0 = (long)blockIdx.x * GBN;
  Operand<float> a{A + r0 * lda, lda, 1, 128, true};
  Operand<float> b = NT ? Operand<float>{B + c0 * ldb, ldb, 1, 128, true} : Operand<float>{B + c0, 1, ldb, 128, true};
  gemm_tile_ring3<float, float, true, NT, BK>(a, b, 0, K, dyn_smem, Epilogue{C + r0 * ldc + c0, ldc, 1, mode, TRI_ALL, 0.f, 0.f});
}
template <int BK>
inline void ring3_launch(float* C, int ldc, const float* A, int lda, const float* B, int ldb, int M, int N, int K, bool nt,
                         int mode, hipStream_t s, size_t lds_min = 0) {
  dim3 grid(N / GBN, M / GBM);
  const size_t lds = Ring3<BK>::LDS_BYTES > lds_min ? Ring3<BK>::LDS_BYTES : lds_min;   // (lds_min: fewer workgroups per CU)
  if (nt) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ring3_kernel<true, BK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    ring3_kernel<true, BK><<<grid, GEMM_THREADS, lds, s>>>(C, ldc, A, lda, B, ldb, M, N, K, mode);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ring3_kernel<false, BK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    ring3_kernel<false, BK><<<grid, GEMM_THREADS, lds, s>>>(C, ldc, A, lda, B, ldb, M, N, K, mode);
  }
}
constexpr int GEMM2_VARIANTS = 3;
inline const char* gemm2_name(int v) {
  static const char* n[] = {"ring3 K16, 3 workgroups per CU", "ring3 K16, 67 KB: 2 per CU", "ring3 K32, 101 KB: 1 per CU"};
  return n[v];
}
template <int ABL>
inline void abl_launch(float* C, int ldc, const float* A, int lda, const float* B, int ldb, int M, int N, int K, bool nt,
                       int mode, hipStream_t s) {
  dim3 grid((N + GBN - 1) / GBN, (M + GBM - 1) / GBM);
  if (nt) abl_kernel<true, ABL><<<grid, GEMM_THREADS, 0, s>>>(C, ldc, A, lda, B, ldb, M, N, K, mode);
  else abl_kernel<false, ABL><<<grid, GEMM_THREADS, 0, s>>>(C, ldc, A, lda, B, ldb, M, N, K, mode);
}
inline void gemm2_launch(int v, float* C, int ldc, const float* A, int lda, const float* B, int ldb, int M, int N, int K,
                         bool nt, int mode, hipStream_t s) {
  if (M % GBM || N % GBN || K % 32) {                              // (the ring variants take full tiles only)
    early_launch<13, false>(C, ldc, A, lda, B, ldb, M, N, K, nt, mode, s);
    return;
  }
  switch (v) {
    case 0: ring3_launch<16>(C, ldc, A, lda, B, ldb, M, N, K, nt, mode, s); break;
    case 1: ring3_launch<16>(C, ldc, A, lda, B, ldb, M, N, K, nt, mode, s, 67584); break;
    default: ring3_launch<32>(C, ldc, A, lda, B, ldb, M, N, K, nt, mode, s); break;
  }
}
}  // namespace gptq
