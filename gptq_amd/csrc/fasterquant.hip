// GPTQ solve: affine grids, the lazy-batch column loop and its trailing update
// (replaces quant.py:6-10, 37-77 and gptq.py:126-305, default branch).
//
// Bit-exactness contract: find_params, quantize and the in-block loop perform the same IEEE fp32
// operations in the same order as the reference's torch CPU ops (true division, round-half-even,
// separate multiply and subtract -- this file is compiled with -ffp-contract=off), so identical
// (W1, Hinv1, scale, zero) give identical bits.  Reductions (the trailing GEMM, the loss sum) are
// tolerance-level: their summation order differs from MKL's.
#include <stdlib.h>

#include <algorithm>

#include "gemm2_f32.h"

namespace gptq {

// ---------------------------------------------------------------------------------------------
// small utilities
// ---------------------------------------------------------------------------------------------
// gptq.py:143-145: dead = diag(H) == 0; H[dead, dead] = 1.  (W[:, dead] = 0 done by zero_dead_kernel.)
__global__ void dead_fix_kernel(float* __restrict__ H, int ldh, int C, int32_t* __restrict__ dead,
                                float* __restrict__ diag) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float* h = H + (long)c * ldh + c;
  const bool d = (*h == 0.f);
  if (d) *h = 1.f;
  dead[c] = d;
  diag[c] = d ? 1.f : *h;
}

__global__ void zero_dead_kernel(float* __restrict__ W, int ldw, int R, int C,
                                 const int32_t* __restrict__ dead) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C || !dead[c]) return;
  for (long r = blockIdx.y; r < R; r += gridDim.y) W[r * ldw + c] = 0.f;      // (grid.y strides over the rows)
}

// perm = argsort(diag, descending), stable on ties (gptq.py:166; torch leaves tie order unspecified).
// Rank by counting: 64 elements per workgroup, the four lanes of a quad split every 1024-value tile of `diag` between
// them (16-byte LDS reads, interleaved so that a wave's four addresses fall into different banks).
__global__ __launch_bounds__(256) void argsort_desc_kernel(const float* __restrict__ diag, int C,
                                                           int32_t* __restrict__ perm) {
  __shared__ __attribute__((aligned(16))) float tile[1024];
  const int i = blockIdx.x * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3;
  const float di = (i < C) ? diag[i] : 0.f;
  int rank = 0;
  for (int j0 = 0; j0 < C; j0 += 1024) {
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += 256) tile[t] = (j0 + t < C) ? diag[j0 + t] : -INFINITY;   // padding never counts
    __syncthreads();
#pragma unroll 4
    for (int q = part; q < 256; q += 4) {
      const float4 d = *reinterpret_cast<const float4*>(tile + 4 * q);
      const int j = j0 + 4 * q;
      rank += (d.x > di) || (d.x == di && j < i);
      rank += (d.y > di) || (d.y == di && j + 1 < i);
      rank += (d.z > di) || (d.z == di && j + 2 < i);
      rank += (d.w > di) || (d.w == di && j + 3 < i);
    }
  }
  rank += __shfl_xor(rank, 1);
  rank += __shfl_xor(rank, 2);
  if (i < C && part == 0) perm[rank] = i;
}

// dst[r][p] = src[r][perm[p]]  (gather = gptq.py:167)  or  dst[r][perm[p]] = src[r][p] (scatter = :301)
template <typename T, bool SCATTER>
__global__ void permute_cols_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd,
                                    int R, int C, const int32_t* __restrict__ perm) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= C) return;
  for (long r = blockIdx.y; r < R; r += gridDim.y) {
    if (SCATTER) dst[r * ldd + perm[p]] = src[r * lds_ + p];
    else dst[r * ldd + p] = src[r * lds_ + perm[p]];
  }
}

// The same through LDS, one workgroup per row at a time: the row is read and written with coalesced 16-byte accesses and
// the permutation happens inside LDS (the direct kernel above moves 4 bytes per 64-byte sector on its permuted side).
// C * 4 bytes of dynamic LDS (C <= 36864); C % 4 == 0 and 16-byte aligned rows.
template <bool SCATTER>
__global__ __launch_bounds__(512) void permute_rows_lds_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst,
                                                               int ldd, int R, int C, const int32_t* __restrict__ perm) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];
  for (int r = blockIdx.x; r < R; r += gridDim.x) {
    const float* s = src + (long)r * lds_;
    float* d = dst + (long)r * ldd;
    if (SCATTER) {
      for (int p = threadIdx.x * 4; p < C; p += 2048) {
        const float4 v = *reinterpret_cast<const float4*>(s + p);
        const int4 q = *reinterpret_cast<const int4*>(perm + p);
        rowbuf[q.x] = v.x; rowbuf[q.y] = v.y; rowbuf[q.z] = v.z; rowbuf[q.w] = v.w;
      }
    } else {
      for (int p = threadIdx.x * 4; p < C; p += 2048)
        *reinterpret_cast<float4*>(rowbuf + p) = *reinterpret_cast<const float4*>(s + p);
    }
    __syncthreads();
    if (SCATTER) {
      for (int p = threadIdx.x * 4; p < C; p += 2048)
        *reinterpret_cast<float4*>(d + p) = *reinterpret_cast<const float4*>(rowbuf + p);
    } else {
      for (int p = threadIdx.x * 4; p < C; p += 2048) {
        const int4 q = *reinterpret_cast<const int4*>(perm + p);
        *reinterpret_cast<float4*>(d + p) = make_float4(rowbuf[q.x], rowbuf[q.y], rowbuf[q.z], rowbuf[q.w]);
      }
    }
    __syncthreads();
  }
}

// col_group[p] = (perm ? perm[p] : p) / groupsize   (gptq.py:256-260)
__global__ void col_group_kernel(int32_t* __restrict__ cg, int C, const int32_t* __restrict__ perm, int g) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < C) cg[p] = (perm ? perm[p] : p) / g;
}

__global__ void gather_tab_col_kernel(const float* __restrict__ stab, const float* __restrict__ ztab,
                                      int tab_ld, const int32_t* __restrict__ cg, int last, int R,
                                      float* __restrict__ s, float* __restrict__ z) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const int g = cg ? cg[last] : 0;
  s[r] = stab[(long)r * tab_ld + g];
  z[r] = ztab[(long)r * tab_ld + g];
}

__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
  __shared__ double part[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)x[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = (float)part[0];
}

// ---------------------------------------------------------------------------------------------
// Quantizer.find_params (quant.py:37-77; perchannel, weight=True, mse=False): one wave per (row, group).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void find_params_kernel(const float* __restrict__ W, int ldw, int R,
                                                          int c0, int c1, int gsize, float maxq, int sym,
                                                          float* __restrict__ scale, float* __restrict__ zero,
                                                          int tab_ld, int g0) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int j = blockIdx.y;
  if (r >= R) return;
  const int lo = c0 + j * gsize, hi = min(lo + gsize, c1);
  const float* w = W + (long)r * ldw;
  float mn = INFINITY, mx = -INFINITY;
  for (int c = lo + lane; c < hi; c += 64) {
    const float v = w[c];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  if (lane != 0) return;
  float xmin = fminf(mn, 0.f), xmax = fmaxf(mx, 0.f);          // quant.py:56-58
  if (sym) {                                                   // quant.py:60-64
    xmax = fmaxf(fabsf(xmin), xmax);
    if (xmin < 0.f) xmin = -xmax;
  }
  if (xmin == 0.f && xmax == 0.f) { xmin = -1.f; xmax = 1.f; } // quant.py:65-67
  const float s = (xmax - xmin) / maxq;                        // quant.py:73
  const float z = sym ? (maxq + 1.f) / 2.f : rintf(-xmin / s); // quant.py:75,77
  scale[(long)r * tab_ld + g0 + j] = s;
  zero[(long)r * tab_ld + g0 + j] = z;
}

__device__ __forceinline__ float affine_code(float x, float s, float z, float maxq) {
  return fminf(fmaxf(rintf(x / s) + z, 0.f), maxq);            // quant.py:9
}

__global__ void quantize_rows_kernel(float* __restrict__ X, int ldx, int R, int C,
                                     const float* __restrict__ scale, const float* __restrict__ zero,
                                     float maxq) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  for (long r = blockIdx.y; r < R; r += gridDim.y) {
    const float s = scale[r], z = zero[r];
    float* x = X + r * ldx + c;
    *x = s * (affine_code(*x, s, z, maxq) - z);                // quant.py:10
  }
}

// ---------------------------------------------------------------------------------------------
// In-block loop (gptq.py:201-271).  Rows are independent; a quad of lanes shares one row, lane c
// holding block columns 4t + c in registers.  Per 4-column super-step the quad exchanges its four
// current columns by DPP quad broadcast and every lane runs the same (bitwise identical) quantize
// chain, so no cross-lane traffic sits on the sequential dependency chain.  The block is walked in
// NPH phases of 32 columns; U rows of a phase are staged in LDS as Us[row][c][t] = U[i][4t + c].
// ---------------------------------------------------------------------------------------------
template <int Q>
__device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), Q * 0x55, 0xf, 0xf, true));
}

struct QuantBlockArgs {
  float* W; int ldw; int R; int i1; int count;
  const float* U; int ldu;
  const float* scale_tab; const float* zero_tab; int tab_ld; const int32_t* col_group;
  float maxq;
  float* Err; uint8_t* codes; int ldc; const int32_t* col_map; float* loss;
  int lde;   // leading dimension of Err (>= blocksize): Err1 of this block is Err[r * lde + 0 .. blocksize)
  int errw;  // = blocksize: Err columns [count, errw) are zeroed (the kernel's own width 32 * NPH may be larger)
  const float* w0; int ldw0;   // nullable: the ORIGINAL weights (same layout as W); Err then receives Q1 - W0 instead of
                               // Err1 (factor form of the trailing updates, see gptq_fasterquant_rows)
  int wide;                    // bit 0: rows of W / Err / w0 allow 16-byte accesses and the block is full: a phase is
                               // retired through an LDS transpose with 32-byte row pieces per lane instead of 4-byte
                               // scatters; bit 1: the same for the codes (no column map, 8-byte aligned rows)
};

// One 32-column phase PH of the block (compile-time, so every "is there a later group" test and every
// register index below is static: no branches inside, no register shuffling between phases).
// U rows of phase PH for the LDS image Us[il][k & 3][k >> 2] = U[i][k], k >= i (upper), else 0: thread `tid`
// owns elements idx = tid + 256 j.  Columns past `count` (tail block) are padded: zero weights, identity U rows
// and a unit grid make their steps exact no-ops, so the hot loop carries no `count` branches.
template <int NPH, int PH, int ABL>
__device__ __forceinline__ void stage_fetch(const QuantBlockArgs& a, int tid, float (&v)[32 * 32 * NPH / 256]) {
  constexpr int B = 32 * NPH;
#pragma unroll
  for (int j = 0; j < 32 * B / 256; ++j) {
    const int idx = tid + 256 * j;
    const int il = idx / B, k = idx % B;
    const int i = 32 * PH + il;
    float x = (k == i) ? 1.f : 0.f;
    if (ABL != 3 && i < a.count && k >= i && k < a.count) x = a.U[(long)(a.i1 + i) * a.ldu + a.i1 + k];
    v[j] = x;
  }
}

template <int NPH>
__device__ __forceinline__ void stage_store(float* Us, int tid, const float (&v)[32 * 32 * NPH / 256]) {
  constexpr int B = 32 * NPH;
  constexpr int LDCL = 8 * NPH + 4;
#pragma unroll
  for (int j = 0; j < 32 * B / 256; ++j) {
    const int idx = tid + 256 * j;
    const int il = idx / B, k = idx % B;
    Us[(il * 4 + (k & 3)) * LDCL + (k >> 2)] = v[j];
  }
}

#ifdef GPTQ_DIAG   // diagnostic library only: s_memtime at the phase boundaries of workgroup 0 of the LAST launch
__device__ unsigned long long qb_stamps[32];
#define QB_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) qb_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QB_STAMP(i) do { } while (0)
#endif
template <int NPH, bool GROUPED, int PH, int ABL = 0>   // ABL: timing-only diagnostic builds (results are wrong)
__device__ __forceinline__ void quant_phase(const QuantBlockArgs& a, float (&w)[8 * NPH], float* Us, float* Us_next,
                                            const int* grp, const int* cmap, float sc, float zr, bool active,
                                            long rbase, float* wrow, int c, int tid, float& loss) {
  constexpr int B = 32 * NPH;
  constexpr int NREG = 8 * NPH;
  constexpr int LDCL = NREG + 4;
  constexpr int ph = PH;
  constexpr int NST = 32 * B / 256;              // staged elements per thread
  // The 32 U rows of this phase were staged into `Us` by the previous phase (or the kernel prologue); the rows
  // of the NEXT phase are fetched into registers now and stored to the other buffer at the end, so their
  // global / L2 latency hides under the column chain instead of preceding it.
  __syncthreads();
  QB_STAMP(1 + 3 * PH);

  float e[8], cd[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) { e[t] = 0.f; cd[t] = 0.f; }

  // grids of this phase's 32 columns: lane c fetches those of columns 4t + c up front (all loads in
  // flight together, off the sequential chain) and the quad shares them by DPP at each step
  float psc[8], pzr[8];
  if (GROUPED) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int col = 32 * ph + 4 * t + c;
      const int g = grp[min(col, B - 1)];
      psc[t] = (active && col < a.count) ? a.scale_tab[rbase * a.tab_ld + g] : 1.f;
      pzr[t] = (active && col < a.count) ? a.zero_tab[rbase * a.tab_ld + g] : 0.f;
    }
  }
  float w0p[8];                                             // factor form: this phase's original weights, for the retire below
  const bool wide = (a.wide & 1) != 0;                      // (block-uniform)
  if (wide) {                                               // columns 32 ph + 8 c ... + 7 of the row: the layout after the transpose
    const float4* p = reinterpret_cast<const float4*>(a.w0 + rbase * a.ldw0 + a.i1 + 32 * ph + 8 * c);
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
    if (a.w0 && active) { lo = p[0]; hi = p[1]; }
    w0p[0] = lo.x; w0p[1] = lo.y; w0p[2] = lo.z; w0p[3] = lo.w; w0p[4] = hi.x; w0p[5] = hi.y; w0p[6] = hi.z; w0p[7] = hi.w;
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int col = 32 * ph + 4 * t + c;
      w0p[t] = (a.w0 && active && col < a.count) ? a.w0[rbase * a.ldw0 + a.i1 + col] : 0.f;
    }
  }
  const bool tail = !GROUPED && (32 * ph + 32 > a.count);   // block-uniform, false except in a tail block
  // (issued after the grid loads above: vector-memory results return in order)
  float nxt[NST];
  if (PH + 1 < NPH) stage_fetch<NPH, PH + 1, ABL>(a, tid, nxt);

#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float* wt = &w[8 * ph];                                    // this phase's window of registers
    float cur[4];
    cur[0] = quad_bcast<0>(wt[t]);
    cur[1] = quad_bcast<1>(wt[t]);
    cur[2] = quad_bcast<2>(wt[t]);
    cur[3] = quad_bcast<3>(wt[t]);
    float gs4[4], gz4[4];
    if (GROUPED) {
      gs4[0] = quad_bcast<0>(psc[t]); gs4[1] = quad_bcast<1>(psc[t]);
      gs4[2] = quad_bcast<2>(psc[t]); gs4[3] = quad_bcast<3>(psc[t]);
      gz4[0] = quad_bcast<0>(pzr[t]); gz4[1] = quad_bcast<1>(pzr[t]);
      gz4[2] = quad_bcast<2>(pzr[t]); gz4[3] = quad_bcast<3>(pzr[t]);
    }
    // (a) the dependent chain of the four columns of this super-step: quantize, error, and the
    //     update of the not-yet-quantized columns of the SAME super-step (quad-uniform values)
    float err4[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int il = 4 * t + cc;
      float gsc = sc, gzr = zr;
      if (GROUPED) { gsc = gs4[cc]; gzr = gz4[cc]; }
      else if (tail && 32 * ph + il >= a.count) { gsc = 1.f; gzr = 0.f; }
      const float* urow = Us + il * 4 * LDCL;
      const float x = cur[cc];
      const float code = ABL == 2 ? fminf(fmaxf(rintf(x * gsc) + gzr, 0.f), a.maxq) : affine_code(x, gsc, gzr, a.maxq);     // gptq.py:262-264
      const float q = gsc * (code - gzr);
      const float d = urow[cc * LDCL + 8 * ph + t];            // Hinv1[i, i]
      const float err = ABL == 2 ? (x - q) * d : (x - q) / d;  // gptq.py:269
      err4[cc] = err;
      loss += err * err;                                       // (w-q)^2/d^2, gptq.py:267 (tolerance-level)
      if (c == cc) { wt[t] = q; e[t] = err; cd[t] = code; }
#pragma unroll
      for (int c2 = cc + 1; c2 < 4; ++c2)
        cur[c2] -= err * urow[c2 * LDCL + 8 * ph + t];         // gptq.py:270
    }
    // (b) the four rank-1 updates of every later column, applied per element in the reference's
    //     order (column i, then i+1, ...): independent across elements, so the LDS reads batch up
    //     and nothing here sits on the chain above.
#pragma unroll
    for (int cc = 0; cc < (ABL == 1 ? 0 : 4); ++cc) {
      const float* ul = Us + (4 * t + cc) * 4 * LDCL + c * LDCL;
      const float err = err4[cc];
#pragma unroll
      for (int j = 8 * ph + t + 1; j < NREG; ++j) w[j] -= err * ul[j];
    }
  }

  QB_STAMP(2 + 3 * PH);
  // retire the window: Q1 -> W, Err1 -> Err, codes
  if (wide) {
    // The lanes of a quad hold columns 4 t + c: written as they are, every store instruction of a wave touches 16 rows
    // with 16 bytes each (measured: 6 k of a phase's 25 k cycles went into 24 such stores per thread).  Through LDS (this
    // phase's U rows are dead) lane c gets columns 8 c ... 8 c + 7 of its row: two 16-byte stores per array.
    constexpr int TLD = 36;                                      // conflict-free for both access patterns
    float* T0 = Us;                                              // [64 rows][36]
    float* T1 = Us + 64 * TLD;                                   // Err1, or (factor form: Err is computed here) the codes
    const int rl = tid >> 2;
    const bool codes_wide = a.w0 && a.codes && (a.wide & 2);
    __syncthreads();                                             // every wave is done with this phase's U rows
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      T0[rl * TLD + 4 * t + c] = w[8 * ph + t];
      T1[rl * TLD + 4 * t + c] = a.w0 ? cd[t] : e[t];
    }
    __syncthreads();
    const float4 q0 = *reinterpret_cast<const float4*>(T0 + rl * TLD + 8 * c);
    const float4 q1 = *reinterpret_cast<const float4*>(T0 + rl * TLD + 8 * c + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(T1 + rl * TLD + 8 * c);
    const float4 s1 = *reinterpret_cast<const float4*>(T1 + rl * TLD + 8 * c + 4);
    if (active) {
      float4 x0 = s0, x1 = s1;
      if (a.w0) {                                                // factor form: Q1 - W0
        x0 = make_float4(q0.x - w0p[0], q0.y - w0p[1], q0.z - w0p[2], q0.w - w0p[3]);
        x1 = make_float4(q1.x - w0p[4], q1.y - w0p[5], q1.z - w0p[6], q1.w - w0p[7]);
      }
      float4* wq = reinterpret_cast<float4*>(wrow + 32 * ph + 8 * c);
      wq[0] = q0; wq[1] = q1;
      float4* eq = reinterpret_cast<float4*>(a.Err + rbase * a.lde + 32 * ph + 8 * c);
      eq[0] = x0; eq[1] = x1;
      if (codes_wide) {
        const uint32_t lo = (uint32_t)s0.x | ((uint32_t)s0.y << 8) | ((uint32_t)s0.z << 16) | ((uint32_t)s0.w << 24);
        const uint32_t hi = (uint32_t)s1.x | ((uint32_t)s1.y << 8) | ((uint32_t)s1.z << 16) | ((uint32_t)s1.w << 24);
        *reinterpret_cast<uint2*>(a.codes + rbase * a.ldc + a.i1 + 32 * ph + 8 * c) = make_uint2(lo, hi);
      } else if (a.codes) {
#pragma unroll
        for (int t = 0; t < 8; ++t) a.codes[rbase * a.ldc + cmap[32 * ph + 4 * t + c]] = (uint8_t)cd[t];
      }
    }
  } else if (active) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int col = 32 * ph + 4 * t + c;
      const float out = a.w0 ? w[8 * ph + t] - w0p[t] : e[t];         // factor form: Q1 - W0
      if (col < a.count) {
        wrow[col] = w[8 * ph + t];
        if (a.codes) a.codes[rbase * a.ldc + cmap[col]] = (uint8_t)cd[t];
      }
      if (col < a.errw) a.Err[rbase * a.lde + col] = (col < a.count) ? out : 0.f;
    }
  }
  if (PH + 1 < NPH) stage_store<NPH>(Us_next, tid, nxt);
  QB_STAMP(3 + 3 * PH);
}

// OCC = 2: at most 256 registers, so that two workgroups fit a compute unit -- for more than 256 workgroups (R > 16384),
// which otherwise run as two rounds (the kernel's natural allocation is ~290 registers: one wave per SIMD).
template <int NPH, bool GROUPED, int ABL = 0, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void quant_block_kernel(QuantBlockArgs a) {
  critical_path_priority();
  QB_STAMP(0);
  constexpr int B = 32 * NPH;
  constexpr int NREG = 8 * NPH;
  constexpr int LDCL = NREG + 4;               // padded class row (16-B aligned, conflict-free b128)
  __shared__ __attribute__((aligned(16))) float Usb[2][32 * 4 * LDCL];   // U rows of the current / next phase
  __shared__ int grp[B];
  __shared__ int cmap[B];

  const int tid = threadIdx.x;
  const int c = tid & 3;
  const int row = blockIdx.x * 64 + (tid >> 2);
  const bool active = row < a.R;
  const long rbase = (long)(active ? row : 0);
  float* wrow = a.W + rbase * a.ldw + a.i1;

  for (int k = tid; k < B; k += 256) {
    grp[k] = (a.col_group && k < a.count) ? a.col_group[a.i1 + k] : 0;
    cmap[k] = (k < a.count) ? (a.col_map ? a.col_map[a.i1 + k] : a.i1 + k) : 0;
  }

  float w[NREG];
#pragma unroll
  for (int j = 0; j < NREG; ++j) {
    const int col = 4 * j + c;
    w[j] = (active && col < a.count) ? wrow[col] : 0.f;
  }
  float sc = 1.f, zr = 0.f;
  if (!GROUPED && active) {
    sc = a.scale_tab[rbase * a.tab_ld];
    zr = a.zero_tab[rbase * a.tab_ld];
  }
  float loss = 0.f;

  {
    float first[32 * B / 256];
    stage_fetch<NPH, 0, ABL>(a, tid, first);
    stage_store<NPH>(Usb[0], tid, first);
  }
#define QPHASE(P) if constexpr (NPH > P) quant_phase<NPH, GROUPED, P, ABL>(a, w, Usb[(P) & 1], Usb[((P) + 1) & 1], grp, cmap, sc, zr, active, rbase, wrow, c, tid, loss)
  QPHASE(0); QPHASE(1); QPHASE(2); QPHASE(3); QPHASE(4); QPHASE(5); QPHASE(6); QPHASE(7);
#undef QPHASE
  if (active && c == 0) a.loss[row] += 0.5f * loss;               // gptq.py:274
  QB_STAMP(1 + 3 * NPH);
}

// W[:, c_begin:c_end] -= E[:, 0:K] @ U[u0 : u0 + K, c_begin:c_end]   (gptq.py:276), exact-fp32 MFMA.
// E [R, lde] holds the Err1 blocks of K consecutive columns u0 .. u0 + K (one lazy-batch block, K = blocksize, for the
// columns of the current super-block; a whole super-block, K = 4 * blocksize, for the columns beyond it).
__global__ __launch_bounds__(GEMM_THREADS) void trailing_kernel(float* __restrict__ W, int ldw, int R, int c_begin,
                                                                int c_end, const float* __restrict__ E, int lde,
                                                                int K, const float* __restrict__ U, int ldu, int u0,
                                                                bool bvec) {
  // 64 x 64 output tiles (gemm2_f32.h): 2x faster than 128 x 128 ones on the small launches inside a super-block,
  // never slower on the large ones, bit-identical
  critical_path_priority();
  __shared__ __attribute__((aligned(16))) float smem[GEMM64_LDS_FLOATS];
  const int tn = blockIdx.x, tm = blockIdx.y;
  const long r0 = (long)tm * SBM, c0 = (long)c_begin + (long)tn * SBN;
  Operand<float> a{E + r0 * lde, lde, 1, (int)min((long)SBM, R - r0),
                   (lde % 4) == 0 && reinterpret_cast<uintptr_t>(E) % 16 == 0};
  Operand<float> b{U + (long)u0 * ldu + c0, 1, ldu, (int)min((long)SBN, c_end - c0), bvec};
  float* Wt = W + r0 * ldw + c0;
  gemm_tile64<float, float, true, false>(a, b, 0, K, smem, Epilogue{Wt, ldw, 1, EPI_SUB, TRI_ALL, 0.f, 0.f});
}

// The same update with one workgroup per 128 x 128 tile (67 KB of LDS, two per compute unit): for the FAR updates on the
// helper stream -- one finishing workgroup then frees a slot that any kernel of the caller's stream fits into (see
// core.cpp, side_ctx).
__global__ __launch_bounds__(GEMM_THREADS) void trailing128_kernel(float* __restrict__ W, int ldw, int R, int c_begin,
                                                                   int c_end, const float* __restrict__ E, int lde,
                                                                   int K, const float* __restrict__ U, int ldu, int u0,
                                                                   bool bvec) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM_LDS_FLOATS];
  const long r0 = (long)blockIdx.y * GBM, c0 = (long)c_begin + (long)blockIdx.x * GBN;
  Operand<float> a{E + r0 * lde, lde, 1, (int)min((long)GBM, R - r0),
                   (lde % 4) == 0 && reinterpret_cast<uintptr_t>(E) % 16 == 0};
  Operand<float> b{U + (long)u0 * ldu + c0, 1, ldu, (int)min((long)GBN, c_end - c0), bvec};
  gemm_tile<float, float, true, false, true>(a, b, 0, K, smem, Epilogue{W + r0 * ldw + c0, ldw, 1, EPI_SUB, TRI_ALL, 0.f, 0.f});
}

// ---------------------------------------------------------------------------------------------
// Factor form of the cross-block compensation (no explicit inverse factor; used by gptq_fasterquant_rows).
// With x = w0 - q (total change of a row), e = its Err1 values and R = U^-1 (H + damp I = R R^T, R upper: what the
// reversed Cholesky factorization yields BEFORE any triangular inverse), the reference's updates e U = x sum up to
//     e = x R      and      W1[:, blk] = W0[:, blk] + (sum_{k < blk} x_k R[k, blk]) U_kk,    U_kk = R_kk^-1,
// because U[B, blk] = -R_BB^-1 R[B, blk] U_kk for the columns B before the block.  So the trailing updates run on an
// working weights in place,  W[:, j] -= (Q1 - W0)[:, blk] Rt[blk, j]  with the rows of  Rt = R blockdiag(U_kk)
// (gptq_rfactor_upper forms it with ONE pass of small products) -- the inverse form's kernels with other operands: the
// original weights are kept in a copy, from which quant_block_kernel forms Q1 - W0 when it retires a phase --
// and only the 128 x 128 diagonal blocks of U are ever needed (the factorization's
// diagonal-block kernel already forms them).  The in-block loop stays the reference's, bit for bit, given (W1, U_kk).
// ---------------------------------------------------------------------------------------------
}  // namespace gptq

using namespace gptq;

// W columns through the act-order permutation (gather: gptq.py:167, scatter back: gptq.py:301)
template <bool SCATTER>
static int permute_cols(const float* src, int lds_, float* dst, int ldd, int R, int C, const int32_t* perm, hipStream_t s) {
  const bool fast = C % 4 == 0 && C <= 36864 && lds_ % 4 == 0 && ldd % 4 == 0 &&
                    reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(perm) % 16 == 0;
  if (fast) {
    const size_t bytes = sizeof(float) * (size_t)C;
    GPTQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&permute_rows_lds_kernel<SCATTER>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    permute_rows_lds_kernel<SCATTER><<<std::min(R, 1024), 512, bytes, s>>>(src, lds_, dst, ldd, R, C, perm);
  } else {
    permute_cols_kernel<float, SCATTER><<<dim3(cdiv(C, 256), std::min(R, 65535)), 256, 0, s>>>(src, lds_, dst, ldd, R, C, perm);
  }
  GPTQ_CHECK_LAUNCH("permute_cols");
  return GPTQ_OK;
}

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" int gptq_find_params(const float* W, int ldw, int R, int c0, int c1, int gsize, int bits,
                                int sym, float* scale, float* zero, int tab_ld, int g0,
                                gptq_stream_t stream) {
  GPTQ_CHECK_ARG(W && scale && zero, "gptq_find_params: null pointer");
  GPTQ_CHECK_ARG(R > 0 && c0 >= 0 && c1 > c0 && gsize > 0 && ldw >= c1, "gptq_find_params: bad sizes");
  GPTQ_CHECK_ARG(bits >= 1 && bits <= 8, "gptq_find_params: bits must be in 1..8 (trits are out of scope)");
  const int ng = cdiv(c1 - c0, gsize);
  GPTQ_CHECK_ARG(tab_ld >= g0 + ng, "gptq_find_params: table too narrow");
  const float maxq = (float)((1 << bits) - 1);
  find_params_kernel<<<dim3(cdiv(R, 4), ng), 256, 0, static_cast<hipStream_t>(stream)>>>(
      W, ldw, R, c0, c1, gsize, maxq, sym, scale, zero, tab_ld, g0);
  GPTQ_CHECK_LAUNCH("find_params_kernel");
  return GPTQ_OK;
}

extern "C" int gptq_quantize_rows(float* X, int ldx, int R, int C, const float* scale, const float* zero,
                                  int bits, gptq_stream_t stream) {
  GPTQ_CHECK_ARG(X && scale && zero && R > 0 && C > 0 && ldx >= C, "gptq_quantize_rows: bad arguments");
  GPTQ_CHECK_ARG(bits >= 1 && bits <= 8, "gptq_quantize_rows: bits must be in 1..8");
  quantize_rows_kernel<<<dim3(cdiv(C, 256), std::min(R, 65535)), 256, 0, static_cast<hipStream_t>(stream)>>>(
      X, ldx, R, C, scale, zero, (float)((1 << bits) - 1));
  GPTQ_CHECK_LAUNCH("quantize_rows_kernel");
  return GPTQ_OK;
}

// QuantBlockArgs::wide for a launch (a full block whose rows allow 16-byte accesses; see quant_phase's retire)
static int quant_block_wide(const QuantBlockArgs& a, int blocksize) {
  static const int off = tune_knob("GPTQ_QB_WIDE", 1) == 0;
  auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  // (128 or 256 columns: the transpose tiles need the U-row buffer of at least four phases; full blocks only)
  if (off || (blocksize != 128 && blocksize != 256) || a.count != blocksize || a.errw != blocksize) return 0;
  if (a.ldw % 4 || a.lde % 4 || a.i1 % 4 || !al16(a.W) || !al16(a.Err)) return 0;
  if (a.w0 && (a.ldw0 % 4 || !al16(a.w0))) return 0;
  int wide = 1;
  if (a.codes && !a.col_map && a.ldc % 8 == 0 && a.i1 % 8 == 0 && reinterpret_cast<uintptr_t>(a.codes) % 8 == 0) wide |= 2;
  return wide;
}

static int launch_quant_block(const QuantBlockArgs& a, int blocksize, bool grouped, hipStream_t s) {
  const int grid = cdiv(a.R, 64);
#ifdef GPTQ_DIAG   // timing-only ablation builds (wrong results): diagnostic library only
  static const int abl = [] { const char* e = getenv("GPTQ_QB_ABLATE"); return e ? atoi(e) : 0; }();
  if (abl && blocksize == 128 && grouped) {
    if (abl == 1) quant_block_kernel<4, true, 1><<<grid, 256, 0, s>>>(a);
    else if (abl == 2) quant_block_kernel<4, true, 2><<<grid, 256, 0, s>>>(a);
    else quant_block_kernel<4, true, 3><<<grid, 256, 0, s>>>(a);
    GPTQ_CHECK_LAUNCH("quant_block_kernel");
    return GPTQ_OK;
  }
#endif
  // any blocksize up to 256 (gptq.py:127 takes any int): the kernel walks 1, 2, 4 or 8 phases of 32 columns (the next
  // power of two: fewer kernel variants to build) and treats the columns past `count` as padding, exactly like the tail
  // block of a matrix
#define QB_CASE(NPH)                                                                  \
  case NPH:                                                                           \
    if (grouped) quant_block_kernel<NPH, true><<<grid, 256, 0, s>>>(a);               \
    else quant_block_kernel<NPH, false><<<grid, 256, 0, s>>>(a);                      \
    break;
  static const int occ_env = tune_knob("GPTQ_QB_OCC", 0);
  if (blocksize > 64 && blocksize <= 128 && (occ_env == 2 || (occ_env == 0 && grid > 256))) {
    if (grouped) quant_block_kernel<4, true, 0, 2><<<grid, 256, 0, s>>>(a);
    else quant_block_kernel<4, false, 0, 2><<<grid, 256, 0, s>>>(a);
    GPTQ_CHECK_LAUNCH("quant_block_kernel");
    return GPTQ_OK;
  }
  int nph = 1;
  while (nph * 32 < blocksize) nph *= 2;
  switch (nph) {
    QB_CASE(1) QB_CASE(2) QB_CASE(4) QB_CASE(8)
    default:
      set_error("blocksize %d unsupported (1 ... 256)", blocksize);
      return GPTQ_ERR_UNSUPPORTED;
  }
#undef QB_CASE
  GPTQ_CHECK_LAUNCH("quant_block_kernel");
  return GPTQ_OK;
}

extern "C" int gptq_quant_block(float* W, int ldw, int R, int C, int i1, int count, int blocksize,
                                const float* U, int ldu, const float* scale_tab, const float* zero_tab,
                                int tab_ld, const int32_t* col_group, int bits, float* Err,
                                uint8_t* codes, int ldc, const int32_t* col_map, float* loss,
                                gptq_stream_t stream) {
  GPTQ_CHECK_ARG(W && U && scale_tab && zero_tab && Err && loss, "gptq_quant_block: null pointer");
  GPTQ_CHECK_ARG(R > 0 && C > 0 && i1 >= 0 && count > 0 && count <= blocksize && i1 + count <= C,
                 "gptq_quant_block: bad block range");
  GPTQ_CHECK_ARG(blocksize >= 1 && blocksize <= 256, "gptq_quant_block: blocksize must be in 1 ... 256");
  GPTQ_CHECK_ARG(ldw >= C && ldu >= C && tab_ld >= 1, "gptq_quant_block: bad leading dimension");
  GPTQ_CHECK_ARG(bits >= 1 && bits <= 8, "gptq_quant_block: bits must be in 1..8");
  QuantBlockArgs a{W, ldw, R, i1, count, U, ldu, scale_tab, zero_tab, tab_ld, col_group,
                   (float)((1 << bits) - 1), Err, codes, ldc, col_map, loss, blocksize, blocksize, nullptr, 0, 0};
  a.wide = quant_block_wide(a, blocksize);
  return launch_quant_block(a, blocksize, col_group != nullptr, static_cast<hipStream_t>(stream));
}

// quant_super.hip: the column loop of a whole super-block in one launch (factor form, static or no groups)
namespace gptq {
struct QuantSuperArgs {
  float* W; int ldw; int R; int s0; int nb;
  const float* U; int ldu;
  const float* scale_tab; const float* zero_tab; int tab_ld; const int32_t* col_group;
  float maxq;
  float* Err; int lde;
  uint8_t* codes; int ldc; const int32_t* col_map;
  float* loss;
  const float* w0; int ldw0;
  int codes_wide;
};
}
int quant_super_lanes(int R);
int launch_quant_super(const gptq::QuantSuperArgs& a, bool grouped, int lanes, hipStream_t s);

namespace {
// The column loop keeps gptq.py's lazy-batch blocks (blocksize columns: quantize, compensate inside the block) but applies
// their trailing updates on two levels: after every block only to the rest of its SUPER-block of SUPER * blocksize
// columns (rank-blocksize products, on the critical path), and once per super-block to everything beyond it (ONE
// rank-(SUPER * blocksize) product: a quarter of the passes over W and a 1.5x better MFMA rate than four rank-128
// updates, measured 95 vs 63 TFLOP/s).  Sums are regrouped, values are the same up to fp32 rounding (the reference's own
// BLAS regroups them too).
constexpr int SUPER = 4;
struct SolveWs {
  float* Wp; float* W0; float* Err; float* loss; float* diag; float* stab; float* ztab;
  int32_t* dead; int32_t* perm; int32_t* cgroup; void* hinv; size_t hinv_bytes; size_t total;
};
// Factor form: needs the 128-blocks of the column loop and of the factorization to coincide, and every dynamic group
// inside one block (columns of LATER blocks hold partial sums that differ from the inverse form's until all the
// blocks before them have been applied, and a dynamic group's grid is taken from them, gptq.py:253-255).
bool use_rform(int C, int blocksize, int groupsize, int static_groups) {
  static const int off = [] { const char* e = getenv("GPTQ_RFORM"); return e && atoi(e) == 0; }();   // 0: inverse form
  if (off || C % 128 != 0 || blocksize != 128) return false;
  return !(groupsize > 0 && !static_groups) || 128 % groupsize == 0;
}
SolveWs carve_solve(void* base, int R, int C, int blocksize, int groupsize, int actorder, int static_groups) {
  Carver cv(base);
  SolveWs w{};
  const int G = groupsize > 0 ? cdiv(C, groupsize) : 1;
  w.Wp = actorder ? cv.take<float>((size_t)R * C) : nullptr;
  w.W0 = use_rform(C, blocksize, groupsize, static_groups) ? cv.take<float>((size_t)R * C) : nullptr;
  w.Err = cv.take<float>((size_t)2 * R * SUPER * blocksize);   // Err1 of a whole super-block, double buffered
  w.loss = cv.take<float>(R);
  w.diag = cv.take<float>(C);
  w.stab = cv.take<float>((size_t)R * G);
  w.ztab = cv.take<float>((size_t)R * G);
  w.dead = cv.take<int32_t>(C);
  w.perm = cv.take<int32_t>(C);
  w.cgroup = cv.take<int32_t>(C);
  w.hinv_bytes = gptq_hinv_workspace_bytes(C);
  w.hinv = cv.take<char>(w.hinv_bytes);
  w.total = cv.used();
  return w;
}
}  // namespace

#ifdef GPTQ_DIAG
extern "C" int gptq_diag_trailing_stamps(unsigned long long* out4) {   // the last 64-tile launch of THIS file: trailing_kernel
  GPTQ_CHECK_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(gemm64_stamps), sizeof(unsigned long long) * 4));
  return GPTQ_OK;
}
#endif
#ifdef GPTQ_DIAG
extern "C" int gptq_diag_qb_stamps(unsigned long long* out32) {
  GPTQ_CHECK_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(qb_stamps), sizeof(unsigned long long) * 32));
  return GPTQ_OK;
}
#endif

extern "C" int gptq_fasterquant_factor_form(int C, int blocksize, int groupsize, int static_groups) {
  return use_rform(C, blocksize, groupsize, static_groups) ? 1 : 0;
}

extern "C" size_t gptq_fasterquant_workspace_bytes(int R, int C, int blocksize, int groupsize,
                                                   int actorder, int static_groups) {
  if (R <= 0 || C <= 0 || blocksize <= 0) return 0;
  return carve_solve(nullptr, R, C, blocksize, groupsize, actorder, static_groups).total;
}

// The head of the solve alone (gptq.py:143-145, 166): the dead-column fix of H's diagonal and, with act-order, the
// permutation -- what a factorization spread over several GPUs (gptq_chol_*) needs BEFORE it starts.  dead_out / perm_out:
// [C] int32; scratch: [C] floats.
extern "C" int gptq_solve_prepare(float* H, int ldh, int C, int actorder, int32_t* dead_out, int32_t* perm_out,
                                  float* scratch, gptq_stream_t stream) {
  GPTQ_CHECK_ARG(H && dead_out && scratch && C > 0 && ldh >= C && (!actorder || perm_out), "gptq_solve_prepare: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  dead_fix_kernel<<<cdiv(C, 256), 256, 0, s>>>(H, ldh, C, dead_out, scratch);
  if (actorder) argsort_desc_kernel<<<cdiv(C, 64), 256, 0, s>>>(scratch, C, perm_out);
  GPTQ_CHECK_LAUNCH("gptq_solve_prepare");
  return GPTQ_OK;
}

static int fasterquant_rows_impl(float* W, int ldw, float* H, int ldh, int R, int C, int bits, int sym,
                                 int blocksize, float percdamp, int groupsize, int actorder,
                                 int static_groups, float* scale_io, float* zero_io, int preset,
                                 float* group_scale, float* group_zero, int32_t* perm_out,
                                 uint8_t* codes, float* error_out, float* row_loss, int32_t* info,
                                 void* workspace, size_t workspace_bytes, gptq_stream_t stream,
                                 const int32_t* dead_in, const int32_t* perm_in);

extern "C" int gptq_fasterquant_rows(float* W, int ldw, float* H, int ldh, int R, int C, int bits, int sym,
                                     int blocksize, float percdamp, int groupsize, int actorder,
                                     int static_groups, float* scale_io, float* zero_io, int preset,
                                     float* group_scale, float* group_zero, int32_t* perm_out,
                                     uint8_t* codes, float* error_out, float* row_loss, int32_t* info,
                                     void* workspace, size_t workspace_bytes, gptq_stream_t stream) {
  return fasterquant_rows_impl(W, ldw, H, ldh, R, C, bits, sym, blocksize, percdamp, groupsize, actorder, static_groups,
                               scale_io, zero_io, preset, group_scale, group_zero, perm_out, codes, error_out, row_loss, info,
                               workspace, workspace_bytes, stream, nullptr, nullptr);
}

// gptq_fasterquant_rows for an H that ALREADY holds what gptq_rfactor_upper leaves (factor form only: see
// gptq_fasterquant_factor_form), factorized elsewhere from the H that gptq_solve_prepare fixed, with its `dead` flags and
// (act-order) its permutation: everything but the dead-column fix, the argsort and the factorization.
extern "C" int gptq_fasterquant_rows_factored(float* W, int ldw, float* H, int ldh, int R, int C, int bits, int sym,
                                              int blocksize, int groupsize, int actorder, int static_groups,
                                              float* scale_io, float* zero_io, int preset, float* group_scale,
                                              float* group_zero, const int32_t* dead_in, const int32_t* perm_in,
                                              uint8_t* codes, float* error_out, float* row_loss, int32_t* info,
                                              void* workspace, size_t workspace_bytes, gptq_stream_t stream) {
  GPTQ_CHECK_ARG(dead_in && (!actorder || perm_in), "gptq_fasterquant_rows_factored: dead_in / perm_in missing");
  GPTQ_CHECK_ARG(use_rform(C, blocksize, groupsize, static_groups), "gptq_fasterquant_rows_factored: factor form only");
  return fasterquant_rows_impl(W, ldw, H, ldh, R, C, bits, sym, blocksize, 0.f, groupsize, actorder, static_groups,
                               scale_io, zero_io, preset, group_scale, group_zero, nullptr, codes, error_out, row_loss, info,
                               workspace, workspace_bytes, stream, dead_in, perm_in);
}

static int fasterquant_rows_impl(float* W, int ldw, float* H, int ldh, int R, int C, int bits, int sym,
                                 int blocksize, float percdamp, int groupsize, int actorder,
                                 int static_groups, float* scale_io, float* zero_io, int preset,
                                 float* group_scale, float* group_zero, int32_t* perm_out,
                                 uint8_t* codes, float* error_out, float* row_loss, int32_t* info,
                                 void* workspace, size_t workspace_bytes, gptq_stream_t stream,
                                 const int32_t* dead_in, const int32_t* perm_in) {
  const bool factored = dead_in != nullptr;
  GPTQ_CHECK_ARG(W && H && scale_io && zero_io && error_out && workspace, "gptq_fasterquant: null pointer");
  GPTQ_CHECK_ARG(R > 0 && C > 0 && ldw >= C && ldh >= C, "gptq_fasterquant: bad sizes");
  GPTQ_CHECK_ARG(R <= 4000000, "gptq_fasterquant: R too large (4,000,000 rows per call)");
  GPTQ_CHECK_ARG(bits >= 1 && bits <= 8, "gptq_fasterquant: bits must be in 1..8 (trits are out of scope)");
  GPTQ_CHECK_ARG(groupsize == -1 || groupsize > 0, "gptq_fasterquant: groupsize must be -1 or positive");
  if (blocksize < 1 || blocksize > 256) {
    set_error("gptq_fasterquant: blocksize %d unsupported (1 ... 256; the reference default is 128)", blocksize);
    return GPTQ_ERR_UNSUPPORTED;
  }
  GPTQ_CHECK_ARG(reinterpret_cast<uintptr_t>(workspace) % 256 == 0, "gptq_fasterquant: workspace must be 256-byte aligned");
  GPTQ_CHECK_ARG(workspace_bytes >= gptq_fasterquant_workspace_bytes(R, C, blocksize, groupsize, actorder, static_groups),
                 "gptq_fasterquant: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const SolveWs ws = carve_solve(workspace, R, C, blocksize, groupsize, actorder, static_groups);
  const bool rform = ws.W0 != nullptr;
  const bool grouped = groupsize > 0;
  const bool use_static = static_groups && grouped;   // range(0, C, -1) is empty (gptq.py:159)
  const int G = grouped ? cdiv(C, groupsize) : 1;
  const float maxq = (float)((1 << bits) - 1);
  const int TB = 256;

  // dead columns (gptq.py:143-145)
  if (factored) {                                                // (done by gptq_solve_prepare before the factorization)
    GPTQ_CHECK_HIP(hipMemcpyAsync(ws.dead, dead_in, sizeof(int32_t) * C, hipMemcpyDeviceToDevice, s));
    if (actorder) GPTQ_CHECK_HIP(hipMemcpyAsync(ws.perm, perm_in, sizeof(int32_t) * C, hipMemcpyDeviceToDevice, s));
    if (info) GPTQ_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int32_t), s));
  } else {
    dead_fix_kernel<<<cdiv(C, TB), TB, 0, s>>>(H, ldh, C, ws.dead, ws.diag);
    if (actorder) argsort_desc_kernel<<<cdiv(C, 64), 256, 0, s>>>(ws.diag, C, ws.perm);   // gptq.py:166
  }
  // Everything that prepares W (dead columns zeroed, static-group grids, act-order gather, full-row grid) depends on
  // diag(H) only, not on the factorization: it runs on the helper stream beside the chain below, which is serial and
  // latency-bound and leaves the chip idle.  (No helper stream: same kernels, caller's stream.)
  SideCtx* sc = (lookahead_mask() & 2) ? side_ctx(s) : nullptr;
  hipStream_t ps = s;
  if (sc) {
    GPTQ_CHECK_HIP(hipEventRecord(sc->main_done, s));
    GPTQ_CHECK_HIP(hipStreamWaitEvent(sc->stream, sc->main_done, 0));
    ps = sc->stream;
  }
  zero_dead_kernel<<<dim3(cdiv(C, TB), std::min(R, 256)), TB, 0, ps>>>(W, ldw, R, C, ws.dead);   // (usually nothing is dead: few workgroups)
  // static groups: grids of the ORIGINAL, uncompensated columns (gptq.py:157-163)
  if (use_static)
    find_params_kernel<<<dim3(cdiv(R, 4), G), 256, 0, ps>>>(W, ldw, R, 0, C, groupsize, maxq, sym,
                                                            ws.stab, ws.ztab, G, 0);
  // act-order (gptq.py:165-169)
  float* Wk = W;
  int ldk = ldw;
  const int32_t* perm = nullptr;
  if (actorder) {
    if (int rc = permute_cols<false>(W, ldw, ws.Wp, C, R, C, ws.perm, ps)) return rc;
    Wk = ws.Wp;
    ldk = C;
    perm = ws.perm;
    if (perm_out) GPTQ_CHECK_HIP(hipMemcpyAsync(perm_out, ws.perm, sizeof(int32_t) * C, hipMemcpyDeviceToDevice, ps));
  }
  // full-row grid unless the quantizer is ready (gptq.py:181-185; per-row min / max do not depend on the column order
  // nor on the factorization, so this may precede it)
  if (!grouped) {
    if (preset) {
      GPTQ_CHECK_HIP(hipMemcpyAsync(ws.stab, scale_io, sizeof(float) * R, hipMemcpyDeviceToDevice, ps));
      GPTQ_CHECK_HIP(hipMemcpyAsync(ws.ztab, zero_io, sizeof(float) * R, hipMemcpyDeviceToDevice, ps));
    } else {
      find_params_kernel<<<dim3(cdiv(R, 4), 1), 256, 0, ps>>>(Wk, ldk, R, 0, C, C, maxq, sym, ws.stab, ws.ztab, 1, 0);
    }
  } else {
    col_group_kernel<<<cdiv(C, TB), TB, 0, ps>>>(ws.cgroup, C, use_static ? perm : nullptr, groupsize);
  }
  if (rform)                                                      // factor form: the original (permuted, dead columns zeroed) weights
    GPTQ_CHECK_HIP(hipMemcpy2DAsync(ws.W0, sizeof(float) * C, Wk, sizeof(float) * ldk, sizeof(float) * C, R,
                                    hipMemcpyDeviceToDevice, ps));
  if (sc) GPTQ_CHECK_HIP(hipEventRecord(sc->prep_done, sc->stream));
  // damped inverse factor (gptq.py:174-180): H <- U, or (factor form) what gptq_rfactor_upper leaves
  if (!factored) {
    const int rc = rform ? gptq_rfactor_upper(H, ldh, C, percdamp, perm, info, ws.hinv, ws.hinv_bytes, stream)
                         : gptq_hinv_upper(H, ldh, C, percdamp, perm, info, ws.hinv, ws.hinv_bytes, stream);
    if (rc != GPTQ_OK) return rc;
  }
  if (sc) GPTQ_CHECK_HIP(hipStreamWaitEvent(s, sc->prep_done, 0));
  GPTQ_CHECK_HIP(hipMemsetAsync(ws.loss, 0, sizeof(float) * R, s));

  const bool bvec_base = (ldh % 4 == 0) && (reinterpret_cast<uintptr_t>(H) % 16 == 0);
  // Super-blocks: dynamic groups read the CURRENT global W of their columns (gptq.py:253-255), so every group must lie
  // inside one super-block (whose columns are kept up to date block by block); otherwise one block = one super-block.
  const bool hier = !(grouped && !use_static) || ((SUPER * blocksize) % groupsize == 0);
  const int SB = hier ? SUPER * blocksize : blocksize;
  // Look-ahead: the far update of a super-block is split into "the next super-block's columns" (caller's stream: the
  // column loop needs them next) and "everything beyond" (helper stream, underneath the next super-block's loop).
  bool side_busy = false;
  int sblk = 0;
  // One launch per super-block (quant_super.hip) whenever the factor form runs with static or no groups and every row
  // allows 16-byte accesses; the per-block launches below remain for everything else.
  auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  const int qs_lanes = (rform && hier && !(grouped && !use_static) && bvec_base && ldk % 4 == 0 && al16(Wk) && al16(ws.W0) &&
                        al16(ws.Err)) ? quant_super_lanes(R) : 0;
  for (int s0 = 0; s0 < C; s0 += SB, ++sblk) {
    const int s1 = std::min(s0 + SB, C);
    float* ErrS = ws.Err + (size_t)(sblk & 1) * R * SUPER * blocksize;   // [R, SB]: block b of the super-block at column b * blocksize
    if (qs_lanes) {
      QuantSuperArgs qa{Wk, ldk, R, s0, (s1 - s0) / blocksize, H, ldh, ws.stab, ws.ztab, G, grouped ? ws.cgroup : nullptr, maxq,
                        ErrS, SB, codes, C, perm, ws.loss, ws.W0, C,
                        (codes && !perm && C % 16 == 0 && al16(codes)) ? 1 : 0};
      const int rc = launch_quant_super(qa, grouped, qs_lanes, s);
      if (rc != GPTQ_OK) return rc;
    }
    for (int i1 = s0; i1 < s1 && !qs_lanes; i1 += blocksize) {       // gptq.py:191
      const int i2 = std::min(i1 + blocksize, s1);
      const int count = i2 - i1;
      float* Err = ErrS + (i1 - s0);
      if (grouped && !use_static) {
        // dynamic groups starting inside this block read the CURRENT global W (gptq.py:253-255)
        const int first = cdiv(i1, groupsize) * groupsize;
        if (first < i2) {
          const int ngb = cdiv(i2 - first, groupsize);
          const int c1 = std::min(C, first + ngb * groupsize);
          if (c1 > s1 && side_busy) {                                // (only without super-blocks) group reaches past them
            GPTQ_CHECK_HIP(hipStreamWaitEvent(s, sc->side_done, 0));
            side_busy = false;
          }
          find_params_kernel<<<dim3(cdiv(R, 4), ngb), 256, 0, s>>>(Wk, ldk, R, first, c1, groupsize, maxq, sym,
                                                                   ws.stab, ws.ztab, G, first / groupsize);
        }
      }
      QuantBlockArgs a{Wk, ldk, R, i1, count, H, ldh, ws.stab, ws.ztab, G,
                       grouped ? ws.cgroup : nullptr, maxq, Err, codes, C, perm, ws.loss, SB, blocksize, rform ? ws.W0 : nullptr, C, 0};
      a.wide = quant_block_wide(a, blocksize);
      const int rc = launch_quant_block(a, blocksize, grouped, s);
      if (rc != GPTQ_OK) return rc;
      if (i2 < s1)                                                   // the rest of this super-block: rank-blocksize
        trailing_kernel<<<dim3(cdiv(s1 - i2, SBN), cdiv(R, SBM)), GEMM_THREADS, 0, s>>>(
            Wk, ldk, R, i2, s1, Err, SB, count, H, ldh, i1, bvec_base && (i2 % 4 == 0));
    }
    if (s1 < C) {                                                    // everything beyond: rank-(s1 - s0)
      if (side_busy) {                                               // the previous far update wrote these columns too
        GPTQ_CHECK_HIP(hipStreamWaitEvent(s, sc->side_done, 0));
        side_busy = false;
      }
      const bool bvec = bvec_base && (s1 % 4 == 0);
      const int next_end = std::min(C, s1 + SB);
      trailing_kernel<<<dim3(cdiv(next_end - s1, SBN), cdiv(R, SBM)), GEMM_THREADS, 0, s>>>(
          Wk, ldk, R, s1, next_end, ErrS, SB, s1 - s0, H, ldh, s0, bvec);
      if (next_end < C) {
        hipStream_t ts = s;
        if (sc) {
          GPTQ_CHECK_HIP(hipEventRecord(sc->main_done, s));
          GPTQ_CHECK_HIP(hipStreamWaitEvent(sc->stream, sc->main_done, 0));
          ts = sc->stream;
        }
        trailing128_kernel<<<dim3(cdiv(C - next_end, GBN), cdiv(R, GBM)), GEMM_THREADS, 0, ts>>>(
            Wk, ldk, R, next_end, C, ErrS, SB, s1 - s0, H, ldh, s0, bvec_base && (next_end % 4 == 0));
        if (sc) {
          GPTQ_CHECK_HIP(hipEventRecord(sc->side_done, sc->stream));
          side_busy = true;
        }
      }
    }
  }
  if (side_busy) GPTQ_CHECK_HIP(hipStreamWaitEvent(s, sc->side_done, 0));
  if (actorder)                                                  // gptq.py:300-301
    if (int rc = permute_cols<true>(ws.Wp, C, W, ldw, R, C, ws.perm, s)) return rc;

  // grid left in the quantizer + optional tables + sum(Losses) (gptq.py:294)
  gather_tab_col_kernel<<<cdiv(R, TB), TB, 0, s>>>(ws.stab, ws.ztab, G, grouped ? ws.cgroup : nullptr,
                                                   C - 1, R, scale_io, zero_io);
  if (grouped && group_scale && group_zero) {
    GPTQ_CHECK_HIP(hipMemcpyAsync(group_scale, ws.stab, sizeof(float) * (size_t)R * G, hipMemcpyDeviceToDevice, s));
    GPTQ_CHECK_HIP(hipMemcpyAsync(group_zero, ws.ztab, sizeof(float) * (size_t)R * G, hipMemcpyDeviceToDevice, s));
  }
  sum_kernel<<<1, 256, 0, s>>>(ws.loss, R, error_out);
  if (row_loss) GPTQ_CHECK_HIP(hipMemcpyAsync(row_loss, ws.loss, sizeof(float) * (size_t)R, hipMemcpyDeviceToDevice, s));
  GPTQ_CHECK_LAUNCH("gptq_fasterquant");
  return GPTQ_OK;
}

extern "C" int gptq_fasterquant(float* W, int ldw, float* H, int ldh, int R, int C, int bits, int sym,
                                int blocksize, float percdamp, int groupsize, int actorder,
                                int static_groups, float* scale_io, float* zero_io, int preset,
                                float* group_scale, float* group_zero, int32_t* perm_out,
                                uint8_t* codes, float* error_out, int32_t* info, void* workspace,
                                size_t workspace_bytes, gptq_stream_t stream) {
  return gptq_fasterquant_rows(W, ldw, H, ldh, R, C, bits, sym, blocksize, percdamp, groupsize, actorder, static_groups,
                               scale_io, zero_io, preset, group_scale, group_zero, perm_out, codes, error_out, nullptr,
                               info, workspace, workspace_bytes, stream);
}
