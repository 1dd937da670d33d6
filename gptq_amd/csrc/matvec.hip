// Packed 3/4-bit dequant mat-vec  mul[col] += sum_k (scale[col]*q[k,col] - zero[col]) * vec[k]
// (replaces quant_cuda.vecquant3matmul / _faster, quant_cuda_kernel.cu:88-244).
//
// HBM-bound: every packed word is read exactly once.  Design for gfx950:
//   * a thread owns 4 adjacent output columns -> 16-byte loads, a wave reads 1 KiB per packed row;
//   * the four waves of a workgroup take different 32-input groups of the same 256 columns and
//     are reduced through LDS, so there is one atomic per column per workgroup (the reference
//     issues one per thread per 256-input slab);
//   * the dequant is algebraically hoisted:  scale * sum(q*x) - zero * sum(x), i.e. one FMA per
//     weight in the hot loop, fp32 accumulation throughout (also for fp16 `vec`, where the
//     reference accumulates in fp16);
//   * grid = column blocks x K-chunks of 16 groups (512 inputs): ~10 workgroups per CU on the FC2 shape;
//     the next group's packed words are prefetched while the current one is decoded.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace gptq {

constexpr int MV_KGROUPS_MAX = 128;   // upper bound on 32-input groups per workgroup (LDS x slab)

template <int BITS>
__device__ __forceinline__ void dot_group(const uint32_t (&w)[BITS], const float* __restrict__ x, float& acc) {
  // 32 codes from BITS words; x[0..31] is wave-uniform (LDS broadcast)
  if (BITS == 3) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int bit = 3 * j, word = bit >> 5, off = bit & 31;
      uint32_t v;
      if (off <= 29) v = (w[word] >> off) & 7u;
      else v = __builtin_amdgcn_alignbit(w[(word + 1) % BITS], w[word], off) & 7u;   // straddles two words
      acc = fmaf((float)v, x[j], acc);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const uint32_t v = (w[j >> 3] >> (4 * (j & 7))) & 15u;
      acc = fmaf((float)v, x[j], acc);
    }
  }
}

template <typename TV> __device__ __forceinline__ float mv_to_f32(TV v);
template <> __device__ __forceinline__ float mv_to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float mv_to_f32<__half>(__half v) { return __half2float(v); }

// VW = output columns per thread (4: 16-byte loads; 1: fallback for width % 4 != 0)
// groupsize == 0: one (scale, zero) per output column ([width]), applied once per workgroup.
// groupsize  > 0: grouped grids, tables [in/groupsize, width] (row = group), applied per 32-input group:
//                 y += s[g][col] * sum_{k in group}(q x) - z[g][col] * sum_{k in group}(x)   (SURVEY row f4).
template <int BITS, int VW, typename TV, bool GROUPED>
__global__ __launch_bounds__(256) void matvec_kernel(const TV* __restrict__ vec, const int32_t* __restrict__ mat,
                                                     float* __restrict__ mul, const float* __restrict__ scales,
                                                     const float* __restrict__ zeros, int ngroups, int width,
                                                     int kgroups, int groupsize) {
  __shared__ __attribute__((aligned(16))) float xs[MV_KGROUPS_MAX * 32];
  __shared__ float xs32[MV_KGROUPS_MAX];
  __shared__ float red[4][64 * VW + 1];
  __shared__ float xsum_s[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g0 = blockIdx.y * kgroups;
  const int ng = min(kgroups, ngroups - g0);
  const int col = (blockIdx.x * 64 + lane) * VW;

  // stage the x chunk (fp32) and its sum
  float part = 0.f;
  for (int k = tid; k < kgroups * 32; k += 256) {
    const float v = (k < ng * 32) ? mv_to_f32<TV>(vec[(long)g0 * 32 + k]) : 0.f;
    xs[k] = v;
    part += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) xsum_s[wave] = part;
  __syncthreads();
  if (GROUPED) {                                         // per-32-group sums of x for the grouped zero term
    for (int g = tid; g < kgroups; g += 256) {
      float sx = 0.f;
#pragma unroll
      for (int j = 0; j < 32; ++j) sx += xs[g * 32 + j];
      xs32[g] = sx;
    }
    __syncthreads();
  }

  float acc[VW];
#pragma unroll
  for (int v = 0; v < VW; ++v) acc[v] = 0.f;
  if (col < width) {
    // software pipeline: the packed words of group g+4 are in flight while group g is decoded
    auto fetch = [&](int g, uint32_t (&w)[BITS][VW]) {
      const int32_t* p = mat + ((long)(g0 + g) * BITS) * width + col;
#pragma unroll
      for (int r = 0; r < BITS; ++r) {
        if (VW == 4) {
          const uint4 q = *reinterpret_cast<const uint4*>(p + (long)r * width);
          w[r][0] = q.x; w[r][1] = q.y; w[r][2] = q.z; w[r][3] = q.w;
        } else {
          w[r][0] = (uint32_t)p[(long)r * width];
        }
      }
    };
    uint32_t wa[BITS][VW], wb[BITS][VW];
    int g = wave;
    if (g < ng) fetch(g, wa);
    float tot[VW];
#pragma unroll
    for (int v = 0; v < VW; ++v) tot[v] = 0.f;
    while (g < ng) {
      const bool more = g + 4 < ng;
      if (more) fetch(g + 4, wb);
      float sg[VW], zg[VW];
      if (GROUPED) {
        const long trow = (long)(((g0 + g) * 32) / groupsize) * width + col;
#pragma unroll
        for (int v = 0; v < VW; ++v) { sg[v] = scales[trow + v]; zg[v] = zeros[trow + v]; acc[v] = 0.f; }
      }
#pragma unroll
      for (int v = 0; v < VW; ++v) {
        uint32_t wc[BITS];
#pragma unroll
        for (int r = 0; r < BITS; ++r) wc[r] = wa[r][v];
        dot_group<BITS>(wc, xs + g * 32, acc[v]);
      }
      if (GROUPED) {
        const float sx = xs32[g];
#pragma unroll
        for (int v = 0; v < VW; ++v) tot[v] += sg[v] * acc[v] - zg[v] * sx;
      }
#pragma unroll
      for (int r = 0; r < BITS; ++r)
#pragma unroll
        for (int v = 0; v < VW; ++v) wa[r][v] = wb[r][v];
      g += 4;
    }
    if (GROUPED) {
#pragma unroll
      for (int v = 0; v < VW; ++v) acc[v] = tot[v];
    }
  }
#pragma unroll
  for (int v = 0; v < VW; ++v) red[wave][lane * VW + v] = acc[v];
  __syncthreads();
  if (tid < 64 * VW) {
    const int c = blockIdx.x * 64 * VW + tid;
    if (c < width) {
      const float q = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
      if (GROUPED) {
        atomicAdd(&mul[c], q);
      } else {
        const float sx = xsum_s[0] + xsum_s[1] + xsum_s[2] + xsum_s[3];
        atomicAdd(&mul[c], scales[c] * q - zeros[c] * sx);
      }
    }
  }
}

template <int BITS>
static int launch_matvec(const void* vec, int vec_dtype, const int32_t* mat, float* mul, const float* scales,
                         const float* zeros, int height, int width, int groupsize, hipStream_t s, const char* who) {
  GPTQ_CHECK_ARG(groupsize == 0 || (groupsize > 0 && groupsize % 32 == 0), "%s: groupsize must be a multiple of 32", who);
  GPTQ_CHECK_ARG(vec && mat && mul && scales && zeros, "%s: null pointer", who);
  GPTQ_CHECK_ARG(height > 0 && width > 0 && height % BITS == 0, "%s: height must be a positive multiple of %d", who, BITS);
  GPTQ_CHECK_ARG(vec_dtype == GPTQ_F32 || vec_dtype == GPTQ_F16, "%s: vec must be fp32 or fp16", who);
  const int ngroups = height / BITS;
  const bool v4 = (width % 4 == 0) && (reinterpret_cast<uintptr_t>(mat) % 16 == 0);
  const int vw = v4 ? 4 : 1;
  static const int kg_env = [] { const char* e = getenv("GPTQ_MV_KGROUPS"); return e ? atoi(e) : 0; }();
  int kgroups = kg_env > 0 ? std::min(kg_env, MV_KGROUPS_MAX) : 16;   // measured best on the 36864 x 9216 FC2 shape
  const dim3 grid(cdiv(width, 64 * vw), cdiv(ngroups, kgroups));
  GPTQ_CHECK_ARG(grid.y <= 65535, "%s: too many input groups", who);
#define MV_LAUNCH(VW, TV)                                                                                          \
  do {                                                                                                             \
    if (groupsize > 0)                                                                                             \
      matvec_kernel<BITS, VW, TV, true><<<grid, 256, 0, s>>>(static_cast<const TV*>(vec), mat, mul, scales, zeros, \
                                                             ngroups, width, kgroups, groupsize);                  \
    else                                                                                                           \
      matvec_kernel<BITS, VW, TV, false><<<grid, 256, 0, s>>>(static_cast<const TV*>(vec), mat, mul, scales,       \
                                                              zeros, ngroups, width, kgroups, groupsize);          \
  } while (0)
  if (vec_dtype == GPTQ_F32) { if (v4) MV_LAUNCH(4, float); else MV_LAUNCH(1, float); }
  else { if (v4) MV_LAUNCH(4, __half); else MV_LAUNCH(1, __half); }
#undef MV_LAUNCH
  GPTQ_CHECK_LAUNCH(who);
  return GPTQ_OK;
}

}  // namespace gptq

using namespace gptq;

extern "C" int gptq_vecquant3matmul(const void* vec, int vec_dtype, const int32_t* mat, float* mul,
                                    const float* scales, const float* zeros, int height, int width,
                                    gptq_stream_t stream) {
  return launch_matvec<3>(vec, vec_dtype, mat, mul, scales, zeros, height, width, 0,
                          static_cast<hipStream_t>(stream), "gptq_vecquant3matmul");
}

extern "C" int gptq_vecquant4matmul(const void* vec, int vec_dtype, const int32_t* mat, float* mul,
                                    const float* scales, const float* zeros, int height, int width,
                                    gptq_stream_t stream) {
  return launch_matvec<4>(vec, vec_dtype, mat, mul, scales, zeros, height, width, 0,
                          static_cast<hipStream_t>(stream), "gptq_vecquant4matmul");
}

extern "C" int gptq_vecquant_matmul_grouped(const void* vec, int vec_dtype, const int32_t* mat, float* mul,
                                            const float* scales, const float* zeros, int height, int width,
                                            int bits, int groupsize, gptq_stream_t stream) {
  GPTQ_CHECK_ARG(bits == 3 || bits == 4, "gptq_vecquant_matmul_grouped: bits must be 3 or 4");
  GPTQ_CHECK_ARG(groupsize > 0, "gptq_vecquant_matmul_grouped: groupsize must be positive");
  GPTQ_CHECK_ARG((height / bits * 32) % groupsize == 0, "gptq_vecquant_matmul_grouped: in_features must be a multiple of groupsize");
  if (bits == 3)
    return launch_matvec<3>(vec, vec_dtype, mat, mul, scales, zeros, height, width, groupsize,
                            static_cast<hipStream_t>(stream), "gptq_vecquant_matmul_grouped");
  return launch_matvec<4>(vec, vec_dtype, mat, mul, scales, zeros, height, width, groupsize,
                          static_cast<hipStream_t>(stream), "gptq_vecquant_matmul_grouped");
}
