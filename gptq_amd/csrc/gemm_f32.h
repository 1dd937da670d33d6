// Exact-fp32 MFMA GEMM tile for gfx950 (v_mfma_f32_32x32x2_f32), shared by the
// Hessian, Cholesky-chain and trailing-update kernels.
//
// One 256-thread workgroup computes a 128x128 output tile; the four waves own
// 64x64 quadrants (2x2 MFMA 32x32 accumulators each).  Operands are staged
// global -> registers -> LDS as [k][tile] images (k-major, leading dimension 132
// so that fragment reads `S[k][m0 + lane&31]` are bank-conflict free), double
// buffered: the next K-slab is fetched into registers while the current one feeds
// the matrix cores.  fp32-in MFMA is bit-for-bit an fmaf chain (no TF32 on gfx950),
// which is what gptq.py:18-19 (allow_tf32 = False) asks for.
#pragma once
#include "common.h"

namespace gptq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM = 128;
constexpr int GBN = 128;
constexpr int GBK = 32;     // a stage's MFMA time (16 k-pairs x 4 MFMA x 64 cyc) must cover the global-load latency
constexpr int GLD = 132;
constexpr int GEMM_THREADS = 256;
constexpr int GEMM_LDS_FLOATS = 2 * 2 * GBK * GLD;  // A,B x double buffer

// Operand element (t, k) lives at p[t * st + k * sk]; pointer already offset to the tile origin.
template <typename T>
struct Operand {
  const T* p;
  long st;
  long sk;
  int rem;   // valid tile rows (<= 128)
  bool vec;  // 4-element vector loads are legal (alignment + leading dimension)
};

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }
template <>
__device__ __forceinline__ float to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }

template <typename T>
__device__ __forceinline__ void load4_vec(const T* p, float out[4]);
template <>
__device__ __forceinline__ void load4_vec<float>(const float* p, float out[4]) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}
template <>
__device__ __forceinline__ void load4_vec<__half>(const __half* p, float out[4]) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  const __half2 lo = *reinterpret_cast<const __half2*>(&v.x);
  const __half2 hi = *reinterpret_cast<const __half2*>(&v.y);
  out[0] = __low2float(lo); out[1] = __high2float(lo);
  out[2] = __low2float(hi); out[3] = __high2float(hi);
}
template <>
__device__ __forceinline__ void load4_vec<__hip_bfloat16>(const __hip_bfloat16* p, float out[4]) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  out[0] = __uint_as_float(v.x << 16); out[1] = __uint_as_float(v.x & 0xffff0000u);
  out[2] = __uint_as_float(v.y << 16); out[3] = __uint_as_float(v.y & 0xffff0000u);
}

// Fetch this thread's share (GST x 4 elements) of a 128 x GBK operand slab starting at k0.
//   KC = false ("tile-contiguous", st == 1): thread -> (k = tid>>5 [+8h], 4 consecutive tile rows)
//   KC = true  ("k-contiguous",    sk == 1): thread -> (tile row = tid>>3 [+32h], 4 consecutive k)
constexpr int GST = GBK / 8;   // 4-element pieces per thread per operand stage

template <typename T, bool KC>
__device__ __forceinline__ void stage_load(const Operand<T>& o, int k0, int k_end, float r[GST][4]) {
  const int tid = threadIdx.x;
  if (!KC) {
    const int m4 = (tid & 31) * 4;
#pragma unroll
    for (int h = 0; h < GST; ++h) {
      const int k = k0 + (tid >> 5) + 8 * h;
      const T* p = o.p + (long)m4 * o.st + (long)k * o.sk;
      if (k < k_end && o.vec && m4 + 4 <= o.rem) {
        load4_vec<T>(p, r[h]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          r[h][e] = (k < k_end && m4 + e < o.rem) ? to_f32<T>(p[(long)e * o.st]) : 0.f;
      }
    }
  } else {
    const int k = k0 + (tid & 7) * 4;
#pragma unroll
    for (int h = 0; h < GST; ++h) {
      const int m = (tid >> 3) + 32 * h;
      const T* p = o.p + (long)m * o.st + (long)k * o.sk;
      if (m < o.rem && o.vec && k + 4 <= k_end) {
        load4_vec<T>(p, r[h]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          r[h][e] = (m < o.rem && k + e < k_end) ? to_f32<T>(p[(long)e * o.sk]) : 0.f;
      }
    }
  }
}

// LDS image stride per operand kind.  Tile-contiguous operands are written with 16-byte stores and need
// a stride that is a multiple of 4 (132).  K-contiguous operands are TRANSPOSED on the way in (4 scalar
// stores per loaded float4); with stride 132 those stores were 4-way bank conflicted (measured: 33-50 % of
// all LDS cycles of the chain GEMMs); an odd stride (129) makes them at most 2-way, which is free for
// ds_write_b32, while fragment reads (32 consecutive floats of one k row) are conflict free for any stride.
template <bool KC>
struct LdsStride { static constexpr int v = KC ? 129 : GLD; };

template <bool KC>
__device__ __forceinline__ void stage_store(float* S, const float r[GST][4]) {
  constexpr int LDS_ = LdsStride<KC>::v;
  const int tid = threadIdx.x;
  if (!KC) {
    const int m4 = (tid & 31) * 4;
#pragma unroll
    for (int h = 0; h < GST; ++h) {
      const int k = (tid >> 5) + 8 * h;
      *reinterpret_cast<float4*>(S + k * LDS_ + m4) = make_float4(r[h][0], r[h][1], r[h][2], r[h][3]);
    }
  } else {
    const int k = (tid & 7) * 4;
#pragma unroll
    for (int h = 0; h < GST; ++h) {
      const int m = (tid >> 3) + 32 * h;
#pragma unroll
      for (int e = 0; e < 4; ++e) S[(k + e) * LDS_ + m] = r[h][e];
    }
  }
}

// Epilogue of a tile: how the 128x128 result meets memory.  Element (row, col) lives at
// C[row * rs + col * cs].  Read-modify-write modes FIRST issue all 64 loads of a lane, then combine,
// then store: written naively (`C[..] -= v` per element) the compiler must keep each load behind the
// previous store and every lane pays 64 serialized memory round trips (measured: ~300 us of a 350 us
// Hessian launch was this).
enum EpiMode { EPI_STORE = 0, EPI_STORE_NEG = 1, EPI_SUB = 2, EPI_AXPBY = 3 };
enum EpiTri { TRI_ALL = 0, TRI_LOWER = 1, TRI_UPPER = 2 };   // keep row >= col / row <= col only

struct Epilogue {
  float* C;
  long rs, cs;
  int mode;
  int tri;
  float alpha, beta;   // EPI_AXPBY: C = alpha * C + beta * acc
  bool wt = false;     // write-through (`sc1`) stores: the tile is handed to another workgroup of the SAME launch behind
                       // a flag (chol_panel_kernel); honoured by the 64 x 64 tile's epilogue only
};

// The old values of a read-modify-write epilogue, loaded BEFORE the product (the tile belongs to this workgroup alone):
// after the k loop the 64 loads were a whole L2 round trip at the end of every tile (gemm2_f32.h measured the same on
// the 64 x 64 tile).
__device__ __forceinline__ void tile_load_old(float (&old)[2][2][16], const Epilogue& ep, int rem_m, int rem_n, int wm,
                                              int wn, int lane) {
  const bool rmw = ep.mode == EPI_SUB || ep.mode == EPI_AXPBY;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int col = wn * 64 + j * 32 + (lane & 31);
        old[i][j][e] = (rmw && row < rem_m && col < rem_n) ? ep.C[(long)row * ep.rs + (long)col * ep.cs] : 0.f;
      }
}
__device__ __forceinline__ void tile_finish(const f32x16 (&acc)[2][2], const float (&old)[2][2][16], const Epilogue& ep,
                                            int rem_m, int rem_n, int wm, int wn, int lane) {
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int col = wn * 64 + j * 32 + (lane & 31);
        bool keep = row < rem_m && col < rem_n;
        if (ep.tri == TRI_LOWER) keep = keep && row >= col;
        if (ep.tri == TRI_UPPER) keep = keep && row <= col;
        const float v = acc[i][j][e];
        float out;
        if (ep.mode == EPI_STORE) out = v;
        else if (ep.mode == EPI_STORE_NEG) out = -v;
        else if (ep.mode == EPI_SUB) out = old[i][j][e] - v;
        else out = ep.alpha * old[i][j][e] + ep.beta * v;       // contraction off: fl(fl(a*h) + fl(b*v))
        if (keep) ep.C[(long)row * ep.rs + (long)col * ep.cs] = out;
      }
}
__device__ __forceinline__ void tile_epilogue(const f32x16 (&acc)[2][2], const Epilogue& ep, int rem_m,
                                              int rem_n, int wm, int wn, int lane) {
  float old[2][2][16];
  tile_load_old(old, ep, rem_m, rem_n, wm, wn, lane);
  tile_finish(acc, old, ep, rem_m, rem_n, wm, wn, lane);
}

// C_tile = sum_{k in [k_begin, k_end)} A(m,k) * B(n,k), combined with memory as `ep` says.
// `smem` must hold GEMM_LDS_FLOATS floats.
template <typename TA, typename TB, bool AKC, bool BKC, bool PRELOAD_OLD = false>
__device__ __forceinline__ void gemm_tile(const Operand<TA>& a, const Operand<TB>& b, int k_begin,
                                          int k_end, float* smem, const Epilogue& ep) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
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
  if (PRELOAD_OLD) tile_load_old(old, ep, a.rem, b.rem, wm, wn, lane);   // in flight under the whole product
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
    // Operand fragments one k-pair AHEAD of the matrix cores: read straight in front of their four MFMAs (as this loop
    // was first written, and as hipcc kept it: ds_read, s_waitcnt lgkmcnt(0), 4 MFMAs) every k-pair exposed an LDS round
    // trip behind 256 cycles of MFMA issue -- the rank-4096 microbenchmark stopped at 75 % of the fp32 matrix peak.
    const int kq = lane >> 5;
    float a0 = Ac[kq * LDA], a1 = Ac[kq * LDA + 32];
    float b0 = Bc[kq * LDB], b1 = Bc[kq * LDB + 32];
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 2) {
      const int kn = (kk + 2 < GBK ? kk + 2 : kk) + kq;            // (last pair: a harmless re-read)
      const float na0 = Ac[kn * LDA], na1 = Ac[kn * LDA + 32];
      const float nb0 = Bc[kn * LDB], nb1 = Bc[kn * LDB + 32];
      asm volatile("" ::: "memory");                               // (or the reads sink back to just before their use)
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
    if (more) {
      stage_store<AKC>(As + (cur ^ 1) * GBK * GLD, ra);
      stage_store<BKC>(Bs + (cur ^ 1) * GBK * GLD, rb);
    }
    __syncthreads();
    cur ^= 1;
  }
  if (PRELOAD_OLD) tile_finish(acc, old, ep, a.rem, b.rem, wm, wn, lane);
  else tile_epilogue(acc, ep, a.rem, b.rem, wm, wn, lane);
}

template <typename T>
static inline bool vec_ok(const T* p, long ld) {
  return (reinterpret_cast<uintptr_t>(p) % (4 * sizeof(T)) == 0) && (ld % 4 == 0);
}

}  // namespace gptq
