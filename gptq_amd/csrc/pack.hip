// int3 / int4 bit packing on the GPU  (replaces Quant3Linear.pack, quant.py:152-187 -- a host
// numpy loop the reference itself marks "TODO: perform packing on GPU", opt.py:361 -- and the
// int4 layout of zeroShot/models/quant.py:176-185).  HBM-bound: reads the [out, in] weights once
// (coalesced along `in`), transposes 64 x 128 tiles of integer codes through LDS, and writes the
// [in/32*bits, out] words coalesced along `out`.  Bit-exact, including the reference's uint32
// wrap-around for out-of-range codes.
#include "common.h"

namespace gptq {

constexpr int PT_O = 64;    // output features per tile
constexpr int PT_I = 128;   // input features per tile (4 groups of 32)

template <typename T> __device__ __forceinline__ float pk_to_f32(T v);
template <> __device__ __forceinline__ float pk_to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float pk_to_f32<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float pk_to_f32<__hip_bfloat16>(__hip_bfloat16 v) { return __bfloat162float(v); }

// 32 codes (uint32 each, reference semantics) -> 3 words: the 96-bit little-endian stream of
// quant.py:166-183, written with the same shifts / masks so that wrap-around matches bit for bit.
__device__ __forceinline__ void pack3_words(const uint32_t* v, int stride, uint32_t out[3]) {
  uint32_t w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
  for (int j = 0; j < 10; ++j) w0 |= v[j * stride] << (3 * j);
  w0 |= v[10 * stride] << 30;
  w1 |= (v[10 * stride] >> 2) & 1u;
#pragma unroll
  for (int j = 0; j < 10; ++j) w1 |= v[(11 + j) * stride] << (3 * j + 1);
  w1 |= v[21 * stride] << 31;
  w2 |= (v[21 * stride] >> 1) & 3u;
#pragma unroll
  for (int j = 0; j < 10; ++j) w2 |= v[(22 + j) * stride] << (3 * j + 2);
  out[0] = w0; out[1] = w1; out[2] = w2;
}

// SRC = 0: weights (T) + per-row scale / zero*scale -> codes;  SRC = 1: ready uint8 codes.
template <typename T, int BITS, int SRC>
__global__ __launch_bounds__(256) void pack_kernel(const T* __restrict__ src, int lds_, int n_out, int n_in,
                                                   const float* __restrict__ scales,
                                                   const float* __restrict__ zeros,
                                                   int32_t* __restrict__ qweight, bool vec4) {
  __shared__ uint32_t cs[PT_O][PT_I + 1];
  const int tid = threadIdx.x;
  const int o0 = blockIdx.y * PT_O, i0 = blockIdx.x * PT_I;

  // read phase: 32 lanes x 4 consecutive inputs = one 128-wide row slice; 8 rows per pass
  const int il = (tid & 31) * 4;
#pragma unroll
  for (int pass = 0; pass < PT_O / 8; ++pass) {
    const int ol = pass * 8 + (tid >> 5);
    const int o = o0 + ol;
    float s = 1.f, zs = 0.f;
    if (SRC == 0 && o < n_out) { s = scales[o]; zs = zeros[o]; }
    if (SRC == 1 && vec4 && o < n_out && i0 + il + 3 < n_in) {  // four codes in one 32-bit load
      const uint32_t q = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(src) + (long)o * lds_ + i0 + il);
#pragma unroll
      for (int e = 0; e < 4; ++e) cs[ol][il + e] = (q >> (8 * e)) & 0xffu;
      continue;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = i0 + il + e;
      uint32_t code = 0;
      if (o < n_out && i < n_in) {
        if (SRC == 0) {
          const float w = pk_to_f32<T>(src[(long)o * lds_ + i]);
          code = (uint32_t)(int32_t)rintf((w + zs) / s);     // quant.py:158
        } else {
          code = (uint32_t)reinterpret_cast<const uint8_t*>(src)[(long)o * lds_ + i];
        }
      }
      cs[ol][il + e] = code;
    }
  }
  __syncthreads();

  // pack phase: thread = (output ol, 32-group g); consecutive lanes -> consecutive outputs
  const int ol = tid & 63, g = tid >> 6;
  const int o = o0 + ol;
  const int gi = i0 / 32 + g;                    // global 32-group index
  if (o >= n_out || (gi + 1) * 32 > n_in) return;
  const uint32_t* v = &cs[ol][32 * g];
  if (BITS == 3) {
    uint32_t w[3];
    pack3_words(v, 1, w);
#pragma unroll
    for (int k = 0; k < 3; ++k) qweight[((long)gi * 3 + k) * n_out + o] = (int32_t)w[k];
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {                // zeroShot/models/quant.py:185
      uint32_t w = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) w |= v[8 * k + j] << (4 * j);
      qweight[((long)gi * 4 + k) * n_out + o] = (int32_t)w;
    }
  }
}

// pack_codes on wide tiles (the hot path packs the solver's uint8 codes): 64 outputs x 512 inputs per workgroup, every
// thread has eight 16-byte loads in flight before the first is used (the 64 x 128 tile above moves 4 bytes per lane and
// load: 2.0 TB/s = a quarter of the HBM roof on 9216 x 36864), the tile stays BYTES in LDS (row stride 528 B: the
// 16-lane groups of a 16-byte read hit 64 distinct banks), a thread packs four 32-groups of one output row and the
// stores of a wave are 256-byte runs along `out`.  Same shifts and masks as pack3_words / the int4 loop above: bit-exact
// including the uint32 wrap-around of out-of-range codes.
constexpr int PW_O = 64, PW_I = 512, PW_LD = PW_I + 16;
template <int BITS>
__global__ __launch_bounds__(256) void pack_codes_wide_kernel(const uint8_t* __restrict__ codes, int ldc, int n_out,
                                                             int n_in, int32_t* __restrict__ qweight) {
  __shared__ __attribute__((aligned(16))) uint8_t cs[PW_O * PW_LD];
  const int tid = threadIdx.x;
  const int o0 = blockIdx.y * PW_O, i0 = blockIdx.x * PW_I;
  uint4 v[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {                 // row = p * 8 + tid / 32, 16 codes at (tid % 32) * 16
    const int ol = p * 8 + (tid >> 5), il = (tid & 31) * 16;
    v[p] = make_uint4(0, 0, 0, 0);
    if (o0 + ol < n_out && i0 + il < n_in)
      v[p] = *reinterpret_cast<const uint4*>(codes + (long)(o0 + ol) * ldc + i0 + il);
  }
#pragma unroll
  for (int p = 0; p < 8; ++p)
    *reinterpret_cast<uint4*>(cs + (p * 8 + (tid >> 5)) * PW_LD + (tid & 31) * 16) = v[p];
  __syncthreads();
  const int ol = tid & 63, o = o0 + ol;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int g = (tid >> 6) + 4 * q;           // 32-group of the tile
    const int gi = i0 / 32 + g;
    if (o >= n_out || (gi + 1) * 32 > n_in) continue;
    const uint4 a = *reinterpret_cast<const uint4*>(cs + ol * PW_LD + 32 * g);
    const uint4 b = *reinterpret_cast<const uint4*>(cs + ol * PW_LD + 32 * g + 16);
    const uint32_t w8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t c[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) c[j] = (w8[j >> 2] >> (8 * (j & 3))) & 0xffu;
    if (BITS == 3) {
      uint32_t w[3];
      pack3_words(c, 1, w);
#pragma unroll
      for (int k = 0; k < 3; ++k) qweight[((long)gi * 3 + k) * n_out + o] = (int32_t)w[k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) w |= c[8 * k + j] << (4 * j);
        qweight[((long)gi * 4 + k) * n_out + o] = (int32_t)w;
      }
    }
  }
}

template <typename T, int SRC>
static int launch_pack(const T* src, int ld, int n_out, int n_in, const float* scales, const float* zeros,
                       int bits, int32_t* qweight, hipStream_t s) {
  const dim3 grid(cdiv(n_in, PT_I), cdiv(n_out, PT_O));
  // uint8 codes: 32-bit loads when every row starts 4-byte aligned
  const bool vec4 = SRC == 1 && ld % 4 == 0 && reinterpret_cast<uintptr_t>(src) % 4 == 0;
  if (bits == 3) pack_kernel<T, 3, SRC><<<grid, 256, 0, s>>>(src, ld, n_out, n_in, scales, zeros, qweight, vec4);
  else pack_kernel<T, 4, SRC><<<grid, 256, 0, s>>>(src, ld, n_out, n_in, scales, zeros, qweight, vec4);
  GPTQ_CHECK_LAUNCH("pack_kernel");
  return GPTQ_OK;
}

// Packed -> dense:  W[o][i] = scale[g][o] * (code - zero[g][o]),  g = i / groupsize  (tables [G, out], `zero`
// the INTEGER zero point, so the arithmetic is the solver's own `scale * (q - zero)`, quant.py:10, and every
// rank of a sharded run reconstructs bit-identical weights).  One thread per (32-input group, output).
template <typename T, int BITS>
__global__ __launch_bounds__(256) void dequant_kernel(const int32_t* __restrict__ qweight,
                                                      const float* __restrict__ scale, const float* __restrict__ zero,
                                                      int n_out, int n_in, int groupsize, T* __restrict__ W, int ldw) {
  const int o = blockIdx.x * 64 + (threadIdx.x & 63);
  const int gi = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (o >= n_out || (gi + 1) * 32 > n_in) return;
  uint32_t w[BITS];
#pragma unroll
  for (int k = 0; k < BITS; ++k) w[k] = (uint32_t)qweight[((long)gi * BITS + k) * n_out + o];
  const long trow = (long)((gi * 32) / groupsize) * n_out + o;
  const float s = scale[trow], z = zero[trow];
  T* dst = W + (long)o * ldw + gi * 32;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    uint32_t v;
    if (BITS == 3) {
      const int bit = 3 * j, word = bit >> 5, off = bit & 31;
      v = (off <= 29) ? (w[word] >> off) & 7u : __builtin_amdgcn_alignbit(w[(word + 1) % BITS], w[word], off) & 7u;
    } else {
      v = (w[j >> 3] >> (4 * (j & 7))) & 15u;
    }
    const float q = s * ((float)v - z);
    if constexpr (sizeof(T) == 4) dst[j] = q;
    else dst[j] = static_cast<T>(q);
  }
}

}  // namespace gptq

using namespace gptq;

static int check_pack_shape(const char* who, int out_features, int in_features, int bits) {
  GPTQ_CHECK_ARG(bits == 3 || bits == 4, "%s: bits must be 3 or 4", who);
  GPTQ_CHECK_ARG(out_features > 0 && in_features > 0, "%s: bad sizes", who);
  GPTQ_CHECK_ARG(in_features % 32 == 0, "%s: in_features must be a multiple of 32", who);
  return GPTQ_OK;
}

extern "C" int gptq_pack_weights(const void* weight, int w_dtype, int ldw, int out_features, int in_features,
                                 const float* scales, const float* zeros, int bits, int32_t* qweight,
                                 gptq_stream_t stream) {
  GPTQ_CHECK_ARG(weight && scales && zeros && qweight, "gptq_pack_weights: null pointer");
  if (int rc = check_pack_shape("gptq_pack_weights", out_features, in_features, bits)) return rc;
  GPTQ_CHECK_ARG(ldw >= in_features, "gptq_pack_weights: leading dimension too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (w_dtype) {
    case GPTQ_F32:
      return launch_pack<float, 0>(static_cast<const float*>(weight), ldw, out_features, in_features, scales, zeros, bits, qweight, s);
    case GPTQ_F16:
      return launch_pack<__half, 0>(static_cast<const __half*>(weight), ldw, out_features, in_features, scales, zeros, bits, qweight, s);
    case GPTQ_BF16:
      return launch_pack<__hip_bfloat16, 0>(static_cast<const __hip_bfloat16*>(weight), ldw, out_features, in_features, scales, zeros, bits, qweight, s);
  }
  GPTQ_CHECK_ARG(false, "gptq_pack_weights: unknown dtype %d", w_dtype);
}

extern "C" int gptq_pack_codes(const uint8_t* codes, int ldc, int out_features, int in_features, int bits,
                               int32_t* qweight, gptq_stream_t stream) {
  GPTQ_CHECK_ARG(codes && qweight, "gptq_pack_codes: null pointer");
  if (int rc = check_pack_shape("gptq_pack_codes", out_features, in_features, bits)) return rc;
  GPTQ_CHECK_ARG(ldc >= in_features, "gptq_pack_codes: leading dimension too small");
  if (ldc % 16 == 0 && reinterpret_cast<uintptr_t>(codes) % 16 == 0) {
    const dim3 grid(cdiv(in_features, PW_I), cdiv(out_features, PW_O));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (bits == 3) pack_codes_wide_kernel<3><<<grid, 256, 0, s>>>(codes, ldc, out_features, in_features, qweight);
    else pack_codes_wide_kernel<4><<<grid, 256, 0, s>>>(codes, ldc, out_features, in_features, qweight);
    GPTQ_CHECK_LAUNCH("pack_codes_wide_kernel");
    return GPTQ_OK;
  }
  return launch_pack<float, 1>(reinterpret_cast<const float*>(codes), ldc, out_features, in_features,
                               nullptr, nullptr, bits, qweight, static_cast<hipStream_t>(stream));
}

extern "C" int gptq_dequant_packed(const int32_t* qweight, const float* scale, const float* zero, int out_features,
                                   int in_features, int bits, int groupsize, void* weight, int w_dtype, int ldw,
                                   gptq_stream_t stream) {
  GPTQ_CHECK_ARG(qweight && scale && zero && weight, "gptq_dequant_packed: null pointer");
  if (int rc = check_pack_shape("gptq_dequant_packed", out_features, in_features, bits)) return rc;
  if (groupsize <= 0) groupsize = in_features;
  GPTQ_CHECK_ARG(groupsize % 32 == 0 && in_features % groupsize == 0 && ldw >= in_features,
                 "gptq_dequant_packed: groupsize must be a multiple of 32 dividing in_features");
  const dim3 grid(cdiv(out_features, 64), cdiv(in_features / 32, 4));
  hipStream_t s = static_cast<hipStream_t>(stream);
#define DQ(T, B) dequant_kernel<T, B><<<grid, 256, 0, s>>>(qweight, scale, zero, out_features, in_features, groupsize, static_cast<T*>(weight), ldw)
  if (w_dtype == GPTQ_F32) { if (bits == 3) DQ(float, 3); else DQ(float, 4); }
  else if (w_dtype == GPTQ_F16) { if (bits == 3) DQ(__half, 3); else DQ(__half, 4); }
  else if (w_dtype == GPTQ_BF16) { if (bits == 3) DQ(__hip_bfloat16, 3); else DQ(__hip_bfloat16, 4); }
  else GPTQ_CHECK_ARG(false, "gptq_dequant_packed: unknown dtype %d", w_dtype);
#undef DQ
  GPTQ_CHECK_LAUNCH("dequant_kernel");
  return GPTQ_OK;
}
