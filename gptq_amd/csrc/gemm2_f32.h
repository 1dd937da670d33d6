// Small-tile companion of gemm_f32.h: the same exact-fp32 MFMA product on a 64x64 output tile per 256-thread workgroup
// (each wave owns ONE 32x32 accumulator).  For the latency-bound launches of the solve -- the panel solve, the rank-128
// updates inside a super-block / outer panel -- which bring a few dozen 128x128 tiles: four times the workgroups, a
// quarter of the MFMA time per workgroup, the same operand traffic per flop from L2.
#pragma once
#include "gemm_f32.h"

namespace gptq {

constexpr int SBM = 64;
constexpr int SBN = 64;
constexpr int SGST = GBK / 16;                       // 4-element pieces per thread per operand stage (64 x 32 floats)
constexpr int GEMM64_LDS_FLOATS = 2 * 2 * GBK * 68;  // A, B x double buffer, [k][64 + 4]

template <typename T, bool KC>
__device__ __forceinline__ void stage_load64(const Operand<T>& o, int k0, int k_end, float r[SGST][4]) {
  const int tid = threadIdx.x;
  if (!KC) {                                         // thread -> (k = tid >> 4 [+16h], 4 consecutive tile rows)
    const int m4 = (tid & 15) * 4;
#pragma unroll
    for (int h = 0; h < SGST; ++h) {
      const int k = k0 + (tid >> 4) + 16 * h;
      const T* p = o.p + (long)m4 * o.st + (long)k * o.sk;
      if (k < k_end && o.vec && m4 + 4 <= o.rem) {
        load4_vec<T>(p, r[h]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[h][e] = (k < k_end && m4 + e < o.rem) ? to_f32<T>(p[(long)e * o.st]) : 0.f;
      }
    }
  } else {                                           // thread -> (tile row = tid >> 3 [+32h], 4 consecutive k)
    const int k = k0 + (tid & 7) * 4;
#pragma unroll
    for (int h = 0; h < SGST; ++h) {
      const int m = (tid >> 3) + 32 * h;
      const T* p = o.p + (long)m * o.st + (long)k * o.sk;
      if (m < o.rem && o.vec && k + 4 <= k_end) {
        load4_vec<T>(p, r[h]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[h][e] = (m < o.rem && k + e < k_end) ? to_f32<T>(p[(long)e * o.sk]) : 0.f;
      }
    }
  }
}

template <bool KC>
struct LdsStride64 { static constexpr int v = KC ? 65 : 68; };

template <bool KC>
__device__ __forceinline__ void stage_store64(float* S, const float r[SGST][4]) {
  constexpr int LDS_ = LdsStride64<KC>::v;
  const int tid = threadIdx.x;
  if (!KC) {
    const int m4 = (tid & 15) * 4;
#pragma unroll
    for (int h = 0; h < SGST; ++h) {
      const int k = (tid >> 4) + 16 * h;
      *reinterpret_cast<float4*>(S + k * LDS_ + m4) = make_float4(r[h][0], r[h][1], r[h][2], r[h][3]);
    }
  } else {
    const int k = (tid & 7) * 4;
#pragma unroll
    for (int h = 0; h < SGST; ++h) {
      const int m = (tid >> 3) + 32 * h;
#pragma unroll
      for (int e = 0; e < 4; ++e) S[(k + e) * LDS_ + m] = r[h][e];
    }
  }
}

#ifdef GPTQ_DIAG   // diagnostic library only: s_memtime stamps of workgroup (0, 0) of the last 64-tile launch
static __device__ unsigned long long gemm64_stamps[8];   // (one copy per translation unit)
#define G64_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) gemm64_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G64_STAMP(i) do { } while (0)
#endif
// The old values of a read-modify-write epilogue.  gemm_tile64 loads them BEFORE the product: the tile belongs to this
// workgroup alone, and after the k loop the 16 loads were a whole L2 round trip on the critical path of every small
// launch (measured: 3.6 k of a rank-128 update's 15.5 k cycles).
__device__ __forceinline__ void tile_load_old64(float (&old)[16], const Epilogue& ep, int rem_m, int rem_n, int wm, int wn,
                                                int lane) {
  const bool rmw = ep.mode == EPI_SUB || ep.mode == EPI_AXPBY;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    const int col = wn * 32 + (lane & 31);
    old[e] = (rmw && row < rem_m && col < rem_n) ? ep.C[(long)row * ep.rs + (long)col * ep.cs] : 0.f;
  }
}
__device__ __forceinline__ void tile_finish64(const f32x16& acc, const float (&old)[16], const Epilogue& ep, int rem_m,
                                              int rem_n, int wm, int wn, int lane) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
    const int col = wn * 32 + (lane & 31);
    bool keep = row < rem_m && col < rem_n;
    if (ep.tri == TRI_LOWER) keep = keep && row >= col;
    if (ep.tri == TRI_UPPER) keep = keep && row <= col;
    const float v = acc[e];
    float out;
    if (ep.mode == EPI_STORE) out = v;
    else if (ep.mode == EPI_STORE_NEG) out = -v;
    else if (ep.mode == EPI_SUB) out = old[e] - v;
    else out = ep.alpha * old[e] + ep.beta * v;
    if (keep) {
      float* dst = ep.C + (long)row * ep.rs + (long)col * ep.cs;
      if (ep.wt) __hip_atomic_store(dst, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *dst = out;
    }
  }
  G64_STAMP(3);
}
__device__ __forceinline__ void tile_epilogue64(const f32x16& acc, const Epilogue& ep, int rem_m, int rem_n, int wm,
                                                int wn, int lane) {
  float old[16];
  tile_load_old64(old, ep, rem_m, rem_n, wm, wn, lane);
  tile_finish64(acc, old, ep, rem_m, rem_n, wm, wn, lane);
}

// acc(64 x 64 tile, this wave's 32 x 32 part) = sum_{k in [k_begin, k_end)} A(m,k) * B(n,k).  The k order of every
// output element is the same ascending fmaf chain as gemm_tile's: bit-identical results.  Ends with a workgroup
// barrier (every wave is done with `smem` and with its global reads).  `smem` must hold GEMM64_LDS_FLOATS floats.
template <typename TA, typename TB, bool AKC, bool BKC>
__device__ __forceinline__ void gemm_acc64(const Operand<TA>& a, const Operand<TB>& b, int k_begin, int k_end,
                                           float* smem, f32x16& acc) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float* As = smem;
  float* Bs = smem + 2 * GBK * 68;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float ra[SGST][4], rb[SGST][4];
  const int nk = (k_end - k_begin + GBK - 1) / GBK;
  G64_STAMP(0);
  if (nk > 0) {
    stage_load64<TA, AKC>(a, k_begin, k_end, ra);
    stage_load64<TB, BKC>(b, k_begin, k_end, rb);
    stage_store64<AKC>(As, ra);
    stage_store64<BKC>(Bs, rb);
  }
  __syncthreads();
  G64_STAMP(1);
  // (Tried: operand registers of two stages, i.e. stage kt + 2 in flight while kt + 1 is stored -- the k loop of a rank-128
  //  update went from 8.3 k to 7.2 k cycles, but the first stage and the epilogue grew by more: 16.3 k instead of 12.2 k.)
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      stage_load64<TA, AKC>(a, k_begin + (kt + 1) * GBK, k_end, ra);
      stage_load64<TB, BKC>(b, k_begin + (kt + 1) * GBK, k_end, rb);
    }
    constexpr int LDA = LdsStride64<AKC>::v, LDB = LdsStride64<BKC>::v;
    const float* Ac = As + cur * GBK * 68 + wm * 32 + (lane & 31);
    const float* Bc = Bs + cur * GBK * 68 + wn * 32 + (lane & 31);
    // all fragments of the stage first, then the MFMA chain: read one k-pair at a time straight in front of its MFMA (as
    // hipcc keeps such a loop) every 64-cycle MFMA waited for a ~100-cycle LDS round trip -- the k loop of a rank-128
    // update took 8.3 k cycles for 4.1 k cycles of MFMAs (round-2 stamps)
    float fa[GBK / 2], fb[GBK / 2];
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      const int k = kk + (lane >> 5);
      fa[kk / 2] = Ac[k * LDA];
      fb[kk / 2] = Bc[k * LDB];
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk / 2], fb[kk / 2], acc, 0, 0, 0);
    if (more) {
      stage_store64<AKC>(As + (cur ^ 1) * GBK * 68, ra);
      stage_store64<BKC>(Bs + (cur ^ 1) * GBK * 68, rb);
    }
    __syncthreads();
    cur ^= 1;
  }
  G64_STAMP(2);
}

// Two products that share the A operand:  acc0 = A B0^T over [k_begin, k0_end),  acc1 = A B1^T over [k_begin, k_end),
// k0_end <= k_end (a multiple of GBK past k_begin) -- the panel solve's two 64-column halves, of which the first needs only
// the first half of the triangular factor's k range.  One pass over A instead of two, one pipeline fill instead of two.
// `smem` must hold GEMM64X2_LDS_FLOATS floats.  Same ascending k order per output element as gemm_acc64.
constexpr int GEMM64X2_LDS_FLOATS = 3 * 2 * GBK * 68;
template <typename TA, typename TB, bool AKC, bool BKC>
__device__ __forceinline__ void gemm_acc64x2(const Operand<TA>& a, const Operand<TB>& b0, const Operand<TB>& b1,
                                             int k_begin, int k0_end, int k_end, float* smem, f32x16& acc0, f32x16& acc1) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  float* As = smem;
  float* B0s = smem + 2 * GBK * 68;
  float* B1s = smem + 4 * GBK * 68;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
  float ra[SGST][4], rb0[SGST][4], rb1[SGST][4];
  const int nk = (k_end - k_begin + GBK - 1) / GBK;
  const int nk0 = (k0_end - k_begin + GBK - 1) / GBK;               // stages in which acc0 takes part
  if (nk > 0) {
    stage_load64<TA, AKC>(a, k_begin, k_end, ra);
    if (nk0 > 0) stage_load64<TB, BKC>(b0, k_begin, k0_end, rb0);
    stage_load64<TB, BKC>(b1, k_begin, k_end, rb1);
    stage_store64<AKC>(As, ra);
    if (nk0 > 0) stage_store64<BKC>(B0s, rb0);
    stage_store64<BKC>(B1s, rb1);
  }
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk, more0 = kt + 1 < nk0;             // (workgroup-uniform)
    if (more) {
      stage_load64<TA, AKC>(a, k_begin + (kt + 1) * GBK, k_end, ra);
      if (more0) stage_load64<TB, BKC>(b0, k_begin + (kt + 1) * GBK, k0_end, rb0);
      stage_load64<TB, BKC>(b1, k_begin + (kt + 1) * GBK, k_end, rb1);
    }
    constexpr int LDA = LdsStride64<AKC>::v, LDB = LdsStride64<BKC>::v;
    const float* Ac = As + cur * GBK * 68 + wm * 32 + (lane & 31);
    const float* B0c = B0s + cur * GBK * 68 + wn * 32 + (lane & 31);
    const float* B1c = B1s + cur * GBK * 68 + wn * 32 + (lane & 31);
    float fa[GBK / 2], fb1[GBK / 2];                                  // (fragments first, see gemm_acc64)
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      const int k = kk + (lane >> 5);
      fa[kk / 2] = Ac[k * LDA];
      fb1[kk / 2] = B1c[k * LDB];
    }
    if (kt < nk0) {
      float fb0[GBK / 2];
#pragma unroll
      for (int kk = 0; kk < GBK; kk += 2) fb0[kk / 2] = B0c[(kk + (lane >> 5)) * LDB];
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < GBK; kk += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk / 2], fb0[kk / 2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk / 2], fb1[kk / 2], acc1, 0, 0, 0);
      }
    } else {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < GBK; kk += 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk / 2], fb1[kk / 2], acc1, 0, 0, 0);
    }
    if (more) {
      stage_store64<AKC>(As + (cur ^ 1) * GBK * 68, ra);
      if (more0) stage_store64<BKC>(B0s + (cur ^ 1) * GBK * 68, rb0);
      stage_store64<BKC>(B1s + (cur ^ 1) * GBK * 68, rb1);
    }
    __syncthreads();
    cur ^= 1;
  }
}

// C_tile(64 x 64) = the product above, combined with memory as `ep` says.
template <typename TA, typename TB, bool AKC, bool BKC>
__device__ __forceinline__ void gemm_tile64(const Operand<TA>& a, const Operand<TB>& b, int k_begin, int k_end,
                                            float* smem, const Epilogue& ep) {
  f32x16 acc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float old[16];
  tile_load_old64(old, ep, a.rem, b.rem, wave >> 1, wave & 1, lane);     // in flight under the whole product
  gemm_acc64<TA, TB, AKC, BKC>(a, b, k_begin, k_end, smem, acc);
  tile_finish64(acc, old, ep, a.rem, b.rem, wave >> 1, wave & 1, lane);
}

}  // namespace gptq
