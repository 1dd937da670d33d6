/*
 * CPU oracle (plain C) for the integer / byte side of the GPTQ hot path.
 * TEST INFRASTRUCTURE -- only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call this.  Restates, line by line, the reference's
 *   - Quant3Linear.pack bit layout            (quant.py:161-186)
 *   - Quant4Linear nibble layout              (zeroShot/models/quant.py:181-185)
 *   - VecQuant3MatMulKernel<float>            (quant_cuda_kernel.cu:88-165)
 * The reference's CUDA sources cannot be compiled here (no nvcc; compiling them as HIP
 * would be a hipify port), so there is no oracle/_ref build: this restatement is pinned by
 * the pack goldens produced by the reference's own Python (tests/golden/g4_pack.npz) and
 * by the numpy oracle (oracle/gptq_oracle.py), which tests/test_oracle_c.py cross-checks.
 */
#include <stdint.h>
#include <string.h>

/* iw: [n_in, n_out] uint32 codes; qweight: [n_in/32*3, n_out] */
void oracle_pack3(const uint32_t* iw, int n_in, int n_out, int32_t* qweight) {
  uint32_t* q = (uint32_t*)qweight;
  memset(q, 0, (size_t)(n_in / 32 * 3) * n_out * sizeof(uint32_t));
  for (int c = 0; c < n_out; ++c) {
    int i = 0, row = 0;
    while (row < n_in / 32 * 3) {                                  /* quant.py:166 */
      for (int j = i; j < i + 10; ++j) q[(size_t)row * n_out + c] |= iw[(size_t)j * n_out + c] << (3 * (j - i));
      i += 10;
      q[(size_t)row * n_out + c] |= iw[(size_t)i * n_out + c] << 30;
      row += 1;
      q[(size_t)row * n_out + c] |= (iw[(size_t)i * n_out + c] >> 2) & 1u;
      i += 1;
      for (int j = i; j < i + 10; ++j) q[(size_t)row * n_out + c] |= iw[(size_t)j * n_out + c] << (3 * (j - i) + 1);
      i += 10;
      q[(size_t)row * n_out + c] |= iw[(size_t)i * n_out + c] << 31;
      row += 1;
      q[(size_t)row * n_out + c] |= (iw[(size_t)i * n_out + c] >> 1) & 3u;
      i += 1;
      for (int j = i; j < i + 10; ++j) q[(size_t)row * n_out + c] |= iw[(size_t)j * n_out + c] << (3 * (j - i) + 2);
      i += 10;
      row += 1;
    }
  }
}

void oracle_pack4(const uint32_t* iw, int n_in, int n_out, int32_t* qweight) {
  uint32_t* q = (uint32_t*)qweight;
  memset(q, 0, (size_t)(n_in / 8) * n_out * sizeof(uint32_t));
  for (int i = 0; i < n_in / 8 * 8; ++i)                           /* zeroShot/models/quant.py:184-185 */
    for (int c = 0; c < n_out; ++c) q[(size_t)(i / 8) * n_out + c] |= iw[(size_t)i * n_out + c] << (4 * (i % 8));
}

/* One output column, one 256-input slab (24 packed rows): the body of quant_cuda_kernel.cu:107-162. */
static float slab3(const int32_t* mat, int width, int row0, int col, const float* x, float scale, float zero) {
  float res = 0.f;
  size_t i = (size_t)width * row0 + col;
  int k = 0;
  uint32_t t1, t2, t;
  while (k < 256) {
    t1 = (uint32_t)mat[i];
    for (int j = 0; j < 10; ++j) res += (scale * (float)((t1 >> (3 * j)) & 7u) - zero) * x[k + j];
    i += width;
    t2 = (uint32_t)mat[i];
    t = (t1 >> 30) | ((t2 << 2) & 4u);
    t2 >>= 1;
    res += (scale * (float)t - zero) * x[k + 10];
    k += 11;
    for (int j = 0; j < 10; ++j) res += (scale * (float)((t2 >> (3 * j)) & 7u) - zero) * x[k + j];
    i += width;
    t1 = (uint32_t)mat[i];
    t = (t2 >> 30) | ((t1 << 1) & 6u);
    t1 >>= 2;
    res += (scale * (float)t - zero) * x[k + 10];
    k += 11;
    for (int j = 0; j < 10; ++j) res += (scale * (float)((t1 >> (3 * j)) & 7u) - zero) * x[k + j];
    i += width;
    k += 10;
  }
  return res;
}

/* mul[col] += sum over slabs; height = in/32*3 must be a multiple of 24 (the reference assumes in % 256 == 0). */
void oracle_vecquant3matmul_f32(const float* vec, const int32_t* mat, float* mul, const float* scales,
                                const float* zeros, int height, int width) {
  for (int row0 = 0; row0 + 24 <= height; row0 += 24) {            /* blockIdx.x */
    const float* x = vec + (size_t)(row0 / 24) * 256;
    for (int col = 0; col < width; ++col)                          /* blockIdx.y * 256 + threadIdx.x */
      mul[col] += slab3(mat, width, row0, col, x, scales[col], zeros[col]);   /* atomicAdd, :164 */
  }
}
