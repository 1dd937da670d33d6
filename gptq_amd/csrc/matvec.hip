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

// ---------------------------------------------------------------------------------------------
// Second generation: TWO weights per decode instruction, no convert.
// A packed code c sitting at bit p of a 16-bit half, OR-ed into the mantissa of the fp16 constant 1024.0 (0x6400), IS the
// fp16 number 1024 + c * 2^p -- no extract, no convert.  One v_and_or_b32 on a 32-bit window of the packed stream
// therefore turns two codes (one per half) into a half2, and v_fma_mix_f32 multiplies either half (op_sel) with an fp32
// activation and adds to an fp32 accumulator: window + and_or + 2 fma = 2 full-rate instructions per weight instead
// of 3 (bfe + cvt + fma).  The activations are pre-scaled by 2^-p per slot while the x chunk is staged in LDS (once per
// workgroup, shared by all 256 columns, exact) and the constant part sum(1024 * 2^-p * x) is removed at the end;
// the codes sit at bits 5..7 of their halves (4-bit: 6), so that part is at most 8-32x (16x) the signal per unit of x.
// (Tried first: v_dot2c_f32_f16 on activation pairs split into two fp16 halves -- fewer instructions still, 1.4-1.9 per
// weight, but no faster than the old kernel: dot2 issues at a quarter of the fma rate.)
// Windows: 3-bit pairs (j, j+5) [15 bits apart] for j in 0..4 and 10..14, (j, j+6) [18 bits] for j in 20..25 -- together
// all 32 codes of a 96-bit group; 4-bit pairs are nibbles (k, k+4) of one word.
// ---------------------------------------------------------------------------------------------
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

struct PairSlot { int jlo, jhi, plo, phi, s; };     // codes, their bit positions inside the halves, window start bit
__host__ __device__ constexpr PairSlot pair3(int t) {
  // t in [0, 16): blocks A (0..4), B (5..9): (j, j+5), lo at p, hi at p - 1;  block C (10..15): (j, j+6), hi at p + 2
  // positions as high in the mantissa as they go: the constant part is 1024 * 2^-p per unit of x (8, 16 or 32 here)
  if (t < 5) { const int j = t, p = 7; return PairSlot{j, j + 5, p, p - 1, 3 * j - p}; }
  if (t < 10) { const int j = 10 + (t - 5), p = 7; return PairSlot{j, j + 5, p, p - 1, 3 * j - p}; }
  { const int j = 20 + (t - 10), p = 5; return PairSlot{j, j + 6, p, p + 2, 3 * j - p}; }
}
__host__ __device__ constexpr PairSlot pair4(int t) {
  // t in [0, 16): word t / 4, nibbles (k, k + 4), both at bit 6 of their halves: window = word shifted by 4k - 6
  const int word = t >> 2, k = t & 3;
  return PairSlot{8 * word + k, 8 * word + k + 4, 6, 6, 32 * word + 4 * k - 6};
}
template <int BITS> __host__ __device__ constexpr PairSlot pair_slot(int t) { return BITS == 3 ? pair3(t) : pair4(t); }
// the same table computed per lane while x is staged (a select cascade over the 16 constexpr entries costs ~100 vector
// instructions per wave, 15 % of the kernel): codes and bit positions only, checked against the table at compile time
template <int BITS> __host__ __device__ constexpr PairSlot pair_slot_rt(int t) {
  if (BITS == 3) {
    const bool c = t >= 10;
    const int j = t + (t >= 5 ? 5 : 0) + (c ? 5 : 0);
    return PairSlot{j, j + (c ? 6 : 5), c ? 5 : 7, c ? 7 : 6, 0};
  }
  const int j = 8 * (t >> 2) + (t & 3);
  return PairSlot{j, j + 4, 6, 6, 0};
}
template <int BITS> constexpr bool pair_slot_rt_ok() {
  for (int t = 0; t < 16; ++t) {
    const PairSlot a = pair_slot<BITS>(t), b = pair_slot_rt<BITS>(t);
    if (a.jlo != b.jlo || a.jhi != b.jhi || a.plo != b.plo || a.phi != b.phi) return false;
  }
  return true;
}
static_assert(pair_slot_rt_ok<3>() && pair_slot_rt_ok<4>(), "pair_slot_rt disagrees with the pair table");

// sum over the 16 lanes of a DPP row, result in every lane (row rotations: no LDS round trips)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// the 32-bit window of the group's bit stream that starts at bit s (may start before bit 0 or run past the end: zeros)
template <int BITS, int S>
__device__ __forceinline__ uint32_t stream_window(const uint32_t (&w)[BITS]) {
  constexpr int NBIT = 32 * BITS;
  if constexpr (S < 0) return w[0] << (-S);
  else if constexpr (S % 32 == 0) return w[S / 32];
  else if constexpr (S / 32 + 1 < BITS && S + 32 <= NBIT) return __builtin_amdgcn_alignbit(w[S / 32 + 1], w[S / 32], S % 32);
  else return w[S / 32] >> (S % 32);
}

template <int BITS, int T>
__device__ __forceinline__ void dot_pairs(const uint32_t (&w)[BITS], const float* __restrict__ xp, uint32_t magic,
                                          float& acc) {
  if constexpr (T < 16) {
    constexpr PairSlot ps = pair_slot<BITS>(T);
    constexpr uint32_t FM = (1u << BITS) - 1;
    constexpr uint32_t mask = (FM << ps.plo) | (FM << (16 + ps.phi));
    // one instruction (hipcc splits `(x & mask) | magic` into v_and + v_or: two 32-bit literals do not fit a VOP3 encoding;
    // with the mask in an SGPR and the magic in a VGPR it does)
    uint32_t q;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(q) : "v"(stream_window<BITS, ps.s>(w)), "s"(mask), "v"(magic));
    // fp16 (low / high half of q) x fp32 + fp32 in one instruction; the product is exact, one rounding at the add
    // (hipcc emits v_cvt_f32_f16 + v_fmac for fmaf((float)h, x, acc): hence the inline asm)
    const float x0 = xp[2 * T], x1 = xp[2 * T + 1];
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(q), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(q), "v"(x1));
    dot_pairs<BITS, T + 1>(w, xp, magic, acc);
  }
}

// Workgroup = 256 columns x KG groups of 32 inputs (KG = 8 by default); its four waves take groups wave, wave + 4, ...
// and keep the packed words of MV_PF groups in flight (1 KiB per load instruction and wave), issued before x is staged.
constexpr int MV_PF = 2;
constexpr int MV2_KG_MAX = 128;
static inline size_t matvec2_lds_bytes(int kg) { return sizeof(float) * ((size_t)kg * (32 + 32 + 2) + 4 * 257); }

template <int BITS, typename TV, bool GROUPED, int ABL = 0>   // ABL: timing-only diagnostic builds (results are wrong)
__global__ __launch_bounds__(256) void matvec2_kernel(const TV* __restrict__ vec, const int32_t* __restrict__ mat,
                                                      float* __restrict__ mul, const float* __restrict__ scales,
                                                      const float* __restrict__ zeros, int ngroups, int width,
                                                      int groupsize, int KG) {
  extern __shared__ __attribute__((aligned(16))) float mv_lds[];
  float* XP = mv_lds;                                     // [KG][16][2] activations of the pair slots, times 2^-p
  float* xs = XP + KG * 32;                               // [KG][32] the x chunk in stream order
  float* Tg = xs + KG * 32;                               // [KG] constant part per group: sum 1024 * 2^-p * x
  float* Sg = Tg + KG;                                    // [KG] sum of x per group (zero-point term)
  float* red = Sg + KG;                                   // [4][257]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: group indices and row bases stay scalar
  const int g0 = blockIdx.y * KG;
  const int ng = min(KG, ngroups - g0);
  const int col = (blockIdx.x * 64 + lane) * 4;

  // the packed words of this wave's first MV_PF groups go out first: their latency overlaps the staging of x below
  auto fetch = [&](int g, uint32_t (&w)[BITS][4]) {
    const int32_t* p = mat + ((long)(g0 + g) * BITS) * width + col;
#pragma unroll
    for (int r = 0; r < BITS; ++r) {
      if (ABL == 3) { w[r][0] = w[r][1] = w[r][2] = w[r][3] = (uint32_t)(g + r + lane); continue; }
      const uint4 q = *reinterpret_cast<const uint4*>(p + (long)r * width);
      w[r][0] = q.x; w[r][1] = q.y; w[r][2] = q.z; w[r][3] = q.w;
    }
  };
  uint32_t ring[MV_PF][BITS][4];
  if (col < width) {
#pragma unroll
    for (int i = 0; i < MV_PF; ++i)
      if (wave + 4 * i < ng) fetch(wave + 4 * i, ring[i]);
  }

  // ---- stage x in fp32, then the pre-scaled pair table ----
  for (int k = tid; k < KG * 32; k += 256) xs[k] = (k < ng * 32) ? mv_to_f32<TV>(vec[(long)g0 * 32 + k]) : 0.f;
  __syncthreads();
  {
    const int t = tid & 15;
    const PairSlot ps = pair_slot_rt<BITS>(t);
    const int jlo = ps.jlo, jhi = ps.jhi, plo = ps.plo, phi = ps.phi;
    for (int slot = tid; slot < KG * 16; slot += 256) {   // (KG * 16 is a multiple of 16: whole groups per 16 lanes)
      const int g = slot >> 4;
      const float a = ldexpf(xs[g * 32 + jlo], -plo), b = ldexpf(xs[g * 32 + jhi], -phi);   // exact
      XP[2 * slot] = a;
      XP[2 * slot + 1] = b;
      const float tpart = row16_sum(1024.f * (a + b));
      const float spart = row16_sum(xs[g * 32 + 2 * t] + xs[g * 32 + 2 * t + 1]);
      if (t == 0) { Tg[g] = tpart; Sg[g] = spart; }
    }
  }
  __syncthreads();

  uint32_t magic = 0x64006400u;                            // half2(1024, 1024), kept in a VGPR (see dot_pairs)
  asm volatile("" : "+v"(magic));
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float tot[4] = {0.f, 0.f, 0.f, 0.f};
  float twave = 0.f;
  if (col < width) {
    for (int base = wave; base < ng; base += 4 * MV_PF) {
#pragma unroll
      for (int i = 0; i < MV_PF; ++i) {
        const int g = base + 4 * i;
        if (g < ng) {                                        // wave-uniform
          if (GROUPED) {
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v] = 0.f;
          }
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            uint32_t wc[BITS];
#pragma unroll
            for (int r = 0; r < BITS; ++r) wc[r] = ring[i][r][v];
            if (ABL == 2) { for (int r = 0; r < BITS; ++r) acc[v] += __builtin_bit_cast(float, wc[r]); continue; }
            dot_pairs<BITS, 0>(wc, XP + g * 32, magic, acc[v]);
          }
          if (GROUPED) {
            const long trow = (long)(((g0 + g) * 32) / groupsize) * width + col;
            const float tg = Tg[g], sx = Sg[g];
#pragma unroll
            for (int v = 0; v < 4; ++v) tot[v] += scales[trow + v] * (acc[v] - tg) - zeros[trow + v] * sx;
          } else {
            twave += Tg[g];
          }
          if (g + 4 * MV_PF < ng) fetch(g + 4 * MV_PF, ring[i]);
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) red[wave * 257 + lane * 4 + v] = GROUPED ? tot[v] : acc[v] - twave;
  __syncthreads();
  const int c = blockIdx.x * 256 + tid;
  if (c < width) {
    const float q = red[tid] + red[257 + tid] + red[2 * 257 + tid] + red[3 * 257 + tid];
    if (GROUPED) {
      atomicAdd(&mul[c], q);
    } else {
      float sx = 0.f;
      for (int g = 0; g < ng; ++g) sx += Sg[g];
      if (ABL == 1) { if (blockIdx.y == (unsigned)(c & 127)) mul[c] = scales[c] * q - zeros[c] * sx; return; }
      atomicAdd(&mul[c], scales[c] * q - zeros[c] * sx);
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
  if (v4) {                                             // two-weights-per-instruction kernel (fp16 magic + dot2)
    // 8 groups of 32 inputs per workgroup (two per wave, both in flight from the first instruction on): measured best
    // on the FC2 shape, kernel time from rocprofv3 -- 8: 28.4 us, 16: 30.3, 32-80 with a 4-deep refill ring: 33-35
    // (3-bit, fp32 x).  What bounds it is vector issue, not HBM: window + and_or + 2 fma_mix = 4 + 4 + 8 + 8 cycles
    // per two weights and SIMD, 340 M weights -> 29 us; nontemporal loads were slower (34 us) on this MALL-resident shape.
    static const int kg2_env = tune_knob("GPTQ_MV_KGROUPS", 0);
    const int colblocks = cdiv(width, 256);
    int kg = kg2_env > 0 ? std::min(kg2_env, MV2_KG_MAX) : 8;
    kg = std::max(1, std::min(kg, ngroups));
    const dim3 grid2(colblocks, cdiv(ngroups, kg));
    GPTQ_CHECK_ARG(grid2.y <= 65535, "%s: too many input groups", who);
    const size_t lds = matvec2_lds_bytes(kg);
#ifdef GPTQ_DIAG   // timing-only ablation builds (wrong results): diagnostic library only
    static const int abl = [] { const char* e = getenv("GPTQ_MV_ABLATE"); return e ? atoi(e) : 0; }();
    if (abl && BITS == 3 && groupsize == 0 && vec_dtype == GPTQ_F32) {
      const float* v = static_cast<const float*>(vec);
      if (abl == 1) matvec2_kernel<3, float, false, 1><<<grid2, 256, lds, s>>>(v, mat, mul, scales, zeros, ngroups, width, 0, kg);
      else if (abl == 2) matvec2_kernel<3, float, false, 2><<<grid2, 256, lds, s>>>(v, mat, mul, scales, zeros, ngroups, width, 0, kg);
      else matvec2_kernel<3, float, false, 3><<<grid2, 256, lds, s>>>(v, mat, mul, scales, zeros, ngroups, width, 0, kg);
      GPTQ_CHECK_LAUNCH(who);
      return GPTQ_OK;
    }
#endif
#define MV2_LAUNCH(TV)                                                                                             \
  do {                                                                                                             \
    if (groupsize > 0)                                                                                             \
      matvec2_kernel<BITS, TV, true><<<grid2, 256, lds, s>>>(static_cast<const TV*>(vec), mat, mul, scales, zeros, \
                                                             ngroups, width, groupsize, kg);                       \
    else                                                                                                           \
      matvec2_kernel<BITS, TV, false><<<grid2, 256, lds, s>>>(static_cast<const TV*>(vec), mat, mul, scales,       \
                                                              zeros, ngroups, width, groupsize, kg);               \
  } while (0)
    if (vec_dtype == GPTQ_F32) MV2_LAUNCH(float); else MV2_LAUNCH(__half);
#undef MV2_LAUNCH
    GPTQ_CHECK_LAUNCH(who);
    return GPTQ_OK;
  }
  static const int kg_env = tune_knob("GPTQ_MV_KGROUPS", 0);
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
  // (only the scalar-width instantiation of the first-generation kernel remains: widths that are no multiple of 4, or
  //  an unaligned buffer; everything else took the kernel above)
  if (vec_dtype == GPTQ_F32) MV_LAUNCH(1, float); else MV_LAUNCH(1, __half);
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
