// Damped inverse factor  U (upper, U^T U = (H + damp I)^-1)  -- replaces gptq.py:174-180.
//
// The reference runs three LAPACK factorizations (cholesky, cholesky_inverse,
// cholesky(upper)), ~4/3 C^3 flop.  Here U is obtained in 2/3 C^3:
//   Abar = J (H + damp I) J        (J = index reversal; padded to a multiple of 128 with I)
//   Abar = L L^T                   (blocked right-looking Cholesky, fp32 MFMA trailing updates)
//   => H + damp I = R R^T with R = J L J upper triangular, hence (H + damp I)^-1 = R^-T R^-1
//   U = R^-1 = J L^-1 J            (L^-1 by recursive doubling: log2(C/128) levels of batched GEMMs)
// gptq_rfactor_upper stops after the Cholesky (C^3 / 3): the column loop's cross-block compensation can run on the
// FACTOR's rows (e = (w0 - q) R, see fasterquant.hip), so it leaves Rt = R blockdiag(U_kk) above the diagonal 128-blocks
// and U_kk = R_kk^-1 inside them -- no triangular inverse.
// All arithmetic is IEEE fp32 (no TF32/bf16), like gptq.py:18-19.
#include <stdlib.h>

#include "gemm2_f32.h"

namespace gptq {

constexpr int NB = 128;   // Cholesky block size == GEMM tile size

// ---------------------------------------------------------------------------------------------
// damp = percdamp * mean(diag(H))   (gptq.py:174), one workgroup.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void diag_mean_kernel(const float* __restrict__ H, int ldh, int C,
                                                        float percdamp, float* __restrict__ damp,
                                                        int32_t* __restrict__ info) {
  __shared__ double part[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < C; i += 256) s += (double)H[(long)i * ldh + i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *damp = percdamp * (float)(part[0] / (double)C);
    if (info) *info = 0;
  }
}

// Abar[i][j] = Hd[p(C-1-i)][p(C-1-j)] + (i == j) * damp for i, j < C; identity in the padding.
// Only the upper triangle of H is read (H is symmetric; add_batch maintains the upper half).
__global__ __launch_bounds__(256) void build_abar_kernel(const float* __restrict__ H, int ldh, int C,
                                                         int Cp, const int32_t* __restrict__ perm,
                                                         const float* __restrict__ damp,
                                                         float* __restrict__ A) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= Cp || j > i) return;   // lower triangle of Abar only
  float v;
  if (i < C) {
    int pi = C - 1 - i, pj = C - 1 - j;
    if (perm) { pi = perm[pi]; pj = perm[pj]; }
    const int lo = min(pi, pj), hi = max(pi, pj);
    v = H[(long)lo * ldh + hi];
    if (i == j) v += *damp;
  } else {
    v = (i == j) ? 1.f : 0.f;
  }
  A[(long)i * Cp + j] = v;
}

// The same with an act-order permutation, from a SYMMETRIC H (both triangles valid): row i of Abar is row
// perm[C-1-i] of H gathered through perm -- the row goes through LDS with 16-byte reads, the gather happens there
// (the direct kernel above reads 4 bytes per 64-byte sector: 0.76 ms at C = 11008 against 0.2 ms with the mirror pass).
__global__ __launch_bounds__(512) void build_abar_perm_kernel(const float* __restrict__ H, int ldh, int C, int Cp,
                                                              const int32_t* __restrict__ perm,
                                                              const float* __restrict__ damp, float* __restrict__ A) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];
  for (int i = blockIdx.x; i < Cp; i += gridDim.x) {
    float* a = A + (long)i * Cp;
    if (i >= C) {                                                  // padding: identity
      for (int j = threadIdx.x; j <= i; j += 512) a[j] = (j == i) ? 1.f : 0.f;
      continue;
    }
    const float* h = H + (long)perm[C - 1 - i] * ldh;
    for (int p = threadIdx.x * 4; p < C; p += 2048)
      *reinterpret_cast<float4*>(rowbuf + p) = *reinterpret_cast<const float4*>(h + p);
    __syncthreads();
    const float dmp = *damp;
    for (int j = threadIdx.x; j <= i; j += 512) a[j] = rowbuf[perm[C - 1 - j]] + (j == i ? dmp : 0.f);
    __syncthreads();
  }
}

// acc(32x32, MFMA C layout) += sign * A * B  over K = 32, operands addressed as A(i,k) = Ap[i*lda + k],
// B(k,j) = Bp[j*ldb + k]  (i.e. B given as its transpose, row-major), all in LDS.
__device__ __forceinline__ void lds_mfma32(f32x16& acc, const float* Ap, int lda, const float* Bp, int ldb,
                                           float sign, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const float a = sign * Ap[r * lda + 2 * kk + h];
    const float b = Bp[r * ldb + 2 * kk + h];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
}
// same with B(k,j) = Bp[k*ldb + j] (B row-major)
__device__ __forceinline__ void lds_mfma32_bn(f32x16& acc, const float* Ap, int lda, const float* Bp, int ldb,
                                              float sign, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const float a = sign * Ap[r * lda + 2 * kk + h];
    const float b = Bp[(2 * kk + h) * ldb + r];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
}
__device__ __forceinline__ void acc_load(f32x16& acc, const float* T, int ld, int lane) {
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = T[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * ld + (lane & 31)];
}
__device__ __forceinline__ void acc_store(const f32x16& acc, float* T, int ld, int lane) {
#pragma unroll
  for (int e = 0; e < 16; ++e) T[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * ld + (lane & 31)] = acc[e];
}

// ---------------------------------------------------------------------------------------------
// Diagonal block: L_kk = chol(A_kk), X = L_kk^-1 written to the diagonal block of Linv.  Hierarchical on 32-wide
// sub-blocks; the 32 x 32 diagonal sub-blocks are factorized AND inverted on the matrix
// cores, one rank-1 update per column.  The sub-block D sits in ONE wave as a 32x32 MFMA accumulator (C layout:
// lane = column, register = row group).  D is symmetric, so its row j -- register e_j of the 32 lanes of half h_j -- is
// also its column j, already spread one element per lane exactly where `v_mfma_f32_32x32x2_f32` takes its A and B
// operands of k-slot h_j: no broadcast, no transposition, the rank-1 update D -= l l^T is ONE MFMA with the scaled row
// in both operand registers (the other half of the wave supplies zeros).  The inverse rides along: M starts as I, per
// column x_j = M[j, :] / l_jj is row j of D^-1 and M -= l_(i>j) x_j is a second MFMA (forward substitution of all 32
// unit vectors at once).  Per column: readlane (pivot) -> rsq + Newton -> two selects -> two MFMAs; the 32-column
// chain takes ~2.5 us instead of ~7 us with v_readlane broadcasts of every multiplier (+ ~1.5 us for the inverse).
// The sub-panel below D is then a plain product with D^-1 (MFMA) instead of a per-row substitution.
// ---------------------------------------------------------------------------------------------
// Round 3: the two recurrences of a column step run on TWO waves.  Wave 0 keeps the factorization (D -= l l^T) and
// publishes every column's l and 1 / l_jj through LDS; wave 1, a step behind on another SIMD, carries the inverse
// (M -= l x_j^T).  In one wave the two dependent MFMAs of a step cost 350 cycles, the factorization alone 250 (round-2
// stamps): wave 1 needs no more than that per step, so the chain of 32 columns ends ~100 cycles after wave 0's.
// Hand-off: wave 0's three LDS stores of a step (l, 1 / l_jj, step counter) are volatile, hence issued in this order,
// and the LDS executes one wave's instructions in order: a reader that has seen the counter reads the step's data.
template <int J>
__device__ __forceinline__ void factor32_stepD(f32x16& S, float* Ld, volatile float* invs, volatile int* step, int base,
                                               int& bad, int c, int lane) {
  constexpr int e = (J & 3) + 4 * (J >> 3);       // register holding row J
  constexpr int h = (J >> 2) & 1;                 // half of the wave holding row J
  const float srow = S[e];                        // (a copy: __builtin_bit_cast of a vector ELEMENT reads element 0)
  const float ajj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, srow), J + 32 * h));
  // a non-positive pivot is only REMEMBERED here (first one wins; reported once, after the 32 steps): no branch and
  // no atomic on the chain
  bad = (bad == 0 && !(ajj > 0.f)) ? J + 1 : bad;
  // 1/sqrt by v_rsq_f32 + one Newton step, l_jj = a_jj * that (tolerance-level, like the rest of the chain)
  float inv = __builtin_amdgcn_rsqf(ajj);
  inv = inv * __builtin_fmaf(-0.5f * ajj * inv, inv, 1.5f);
  // `c` is laundered by the caller once per step, so that the 64 lane masks (c >= J) are compared here, in the shadow of
  // the MFMA, instead of being hoisted out of the 32 steps into scalar registers that spill
  const bool mine = (lane >> 5) == h;
  const float l = (mine && c >= J) ? srow * inv : 0.f;          // column J of L_D, l_J = sqrt(a_JJ)
  // both halves store (the idle half into the scratch column 32 of the 33-wide buffer): no exec-mask branch per step
  // plain stores kept in program order by compiler barriers (volatile ones are each followed by s_waitcnt lgkmcnt(0): an
  // LDS round trip per store on the chain -- measured 500 instead of 250 cycles per step); the hardware keeps the order
  Ld[c * 33 + (mine ? J : 32)] = l;
  const_cast<float*>(invs)[J] = inv;
  asm volatile("" ::: "memory");
  *const_cast<int*>(step) = base + J + 1;
  asm volatile("" ::: "memory");
  S = __builtin_amdgcn_mfma_f32_32x32x2f32(-l, l, S, 0, 0, 0);
}
// Wave 1's view of a column: the step counter FIRST, then the column's data -- issued back to back, executed by the
// LDS in that order, so data that travels with a counter value >= its step is the published data (one LDS round trip
// per step instead of two; the read for column J + 1 is issued before column J's MFMA).
struct ColView { int sv; float l, inv; };
template <int J>
__device__ __forceinline__ void col_fetch(ColView& v, float* Ld, volatile float* invs, volatile int* step, int c) {
  asm volatile("" ::: "memory");
  v.sv = *const_cast<int*>(step);
  asm volatile("" ::: "memory");                                // (the counter's read is issued before the data's)
  v.l = Ld[c * 33 + J];
  v.inv = const_cast<float*>(invs)[J];
  asm volatile("" ::: "memory");
}
template <int J>
__device__ __forceinline__ void factor32_stepM(f32x16& M, float* Xd, int ldx, float* Ld, volatile float* invs,
                                               volatile int* step, int base, int c, int lane, ColView& v) {
  constexpr int e = (J & 3) + 4 * (J >> 3);
  constexpr int h = (J >> 2) & 1;
  while (v.sv < base + J + 1) {                                 // (wave 0 of this workgroup always gets there)
    __builtin_amdgcn_s_sleep(0);
    col_fetch<J>(v, Ld, invs, step, c);
  }
  const float l = v.l, inv = v.inv;                             // column J of L_D (rows c), 1 / l_JJ
  if constexpr (J < 31) col_fetch<J + 1>(v, Ld, invs, step, c); // speculative: checked by the next step
  const bool mine = (lane >> 5) == h;
  const float x = mine ? M[e] * inv : 0.f;                      // row J of L_D^-1 (exact zeros right of the diagonal)
  *(mine ? Xd + J * ldx + c : Ld + 32 * 33 + c) = x;            // (the idle half: 32 scratch floats behind Ld)
  const float ls = (mine && c > J) ? l : 0.f;
  M = __builtin_amdgcn_mfma_f32_32x32x2f32(-ls, x, M, 0, 0, 0);
}

#ifdef GPTQ_DIAG   // diagnostic library only: s_memtime at the phase boundaries of the LAST launch (100 MHz... see tools)
__device__ unsigned long long potrf_stamps[32];
__device__ int potrf_ablate;
#define POTRF_STAMP(i) do { if (threadIdx.x == 0) potrf_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define POTRF_STAMP(i) do { } while (0)
#endif
// (WT: the block's inverse leaves with write-through `sc1` stores -- chol_panel_kernel hands it to other workgroups of the
//  same launch behind a flag)
// The block's lower triangle from A into the 128 x 129 image S, mirrored (diagonal tiles are kept full): coalesced, ALL 32
// loads of a thread in flight before the first one is used (the plain loop took 21 k of the kernel's 106 k cycles: one L2
// round trip per iteration); the transposed store has stride 129: conflict-free.  The caller passes a barrier next.
__device__ __forceinline__ void potrf_load_block(float* S, const float* __restrict__ Ak, int Cp, const int tid) {
  constexpr int LD = NB + 1;
  float v[NB * NB / 512];
#pragma unroll
  for (int j = 0; j < NB * NB / 512; ++j) {
    const int idx = tid + 512 * j, i = idx >> 7, k = idx & 127;
    v[j] = (k <= i) ? Ak[(long)i * Cp + k] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < NB * NB / 512; ++j) {
    const int idx = tid + 512 * j, i = idx >> 7, k = idx & 127;
    if (k <= i) {
      S[i * LD + k] = v[j];
      S[k * LD + i] = v[j];
    }
  }
}
// (WT: the block's inverse leaves with write-through `sc1` stores -- chol_panel_kernel hands it to other workgroups of the
//  same launch behind a flag.)  S = dsm holds the mirrored block; no barrier needed in between.
template <bool WT>
__device__ __forceinline__ void potrf_inv_diag_block(float* dsm, float* __restrict__ A, float* __restrict__ Linv, int Cp,
                                                     int kb, int32_t* __restrict__ info, const int tid) {
  // ONE 128 x 129 image (66 KB) serves the block, its factor and its inverse: a sub-block's inverse replaces it as
  // soon as it is final.  The footprint matters: the kernel is a single workgroup on the critical path and must find
  // a compute unit with that much free LDS beside the big low-priority updates of the helper stream.
  constexpr int LD = NB + 1;            // 129: row-strided accesses (lane = row) are conflict-free
  float* S = dsm;                       // [128][129]  A_kk (full, mirrored) -> L off-diagonal blocks -> L_kk^-1
  float* Tt = S + 64;                   // [64][LD]    level-2 intermediate, in the dead quadrant S[0:64, 64:128]
  float* Ld = S + NB * LD;              // [32][33]    factor of the current diagonal 32 x 32 sub-block (+ 32 scratch floats)
  volatile float* invs = Ld + 32 * 33 + 32;                       // [32] 1 / l_jj of the current sub-block (wave 0 -> wave 1)
  volatile int* step = reinterpret_cast<volatile int*>(Ld + 32 * 33 + 64);   // columns published so far (monotonic)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* Ak = A + (long)kb * NB * Cp + (long)kb * NB;

  if (tid == 0) *step = 0;
  __syncthreads();
  POTRF_STAMP(1);
  POTRF_STAMP(2);

  // ------------------------------- factorization -------------------------------
#pragma unroll 1
  for (int s = 0; s < 4; ++s) {
    const int o = 32 * s;
    if (wave == 0) {                                            // (A1) D = L_D L_D^T on the matrix cores ...
      f32x16 D;
      acc_load(D, S + o * LD + o, LD, lane);
      const int c0 = kb * NB + o;
      int bad = 0, cl = lane & 31;
#define FSTEP(J) asm volatile("" : "+v"(cl)); factor32_stepD<J>(D, Ld, invs, step, 32 * s, bad, cl, lane)
      FSTEP(0); FSTEP(1); FSTEP(2); FSTEP(3); FSTEP(4); FSTEP(5); FSTEP(6); FSTEP(7);
      FSTEP(8); FSTEP(9); FSTEP(10); FSTEP(11); FSTEP(12); FSTEP(13); FSTEP(14); FSTEP(15);
      FSTEP(16); FSTEP(17); FSTEP(18); FSTEP(19); FSTEP(20); FSTEP(21); FSTEP(22); FSTEP(23);
      FSTEP(24); FSTEP(25); FSTEP(26); FSTEP(27); FSTEP(28); FSTEP(29); FSTEP(30); FSTEP(31);
#undef FSTEP
      if (bad && lane == 0 && info) atomicCAS(info, 0, c0 + bad);
    } else if (wave == 1) {                                     // ... and X_D = L_D^-1 one step behind, on another SIMD
      f32x16 M;
#pragma unroll
      for (int e = 0; e < 16; ++e) M[e] = ((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) == (lane & 31)) ? 1.f : 0.f;
      float* Xd = S + o * LD + o;                               // X_D takes D's place (wave 0 holds D in registers by now)
      int cl = lane & 31;
      // (wave 0 reads D out of the image before its first step publishes anything: D's image is dead once step 1 is seen)
      ColView cv;
      col_fetch<0>(cv, Ld, invs, step, cl);
#define MSTEP(J) asm volatile("" : "+v"(cl)); factor32_stepM<J>(M, Xd, LD, Ld, invs, step, 32 * s, cl, lane, cv)
      MSTEP(0); MSTEP(1); MSTEP(2); MSTEP(3); MSTEP(4); MSTEP(5); MSTEP(6); MSTEP(7);
      MSTEP(8); MSTEP(9); MSTEP(10); MSTEP(11); MSTEP(12); MSTEP(13); MSTEP(14); MSTEP(15);
      MSTEP(16); MSTEP(17); MSTEP(18); MSTEP(19); MSTEP(20); MSTEP(21); MSTEP(22); MSTEP(23);
      MSTEP(24); MSTEP(25); MSTEP(26); MSTEP(27); MSTEP(28); MSTEP(29); MSTEP(30); MSTEP(31);
#undef MSTEP
    }
    __syncthreads();
    POTRF_STAMP(3 + 3 * s);
    const int nb_rem = 3 - s;                                   // 32-row blocks under the diagonal sub-block
    // L_kk itself goes back to A (the column loop's far updates multiply with the FACTOR, see gptq_rfactor_upper):
    // the diagonal sub-block's factor now, by a wave that has nothing to do in (A2); the blocks under it in (A3)
    if (wave >= 4) {                                            // (four waves: one took 2.7 k cycles, the whole phase)
      for (int idx = (wave - 4) * 64 + lane; idx < 32 * 32; idx += 256) {
        const int i = idx >> 5, k = idx & 31;
        if (k <= i) Ak[(long)(o + i) * Cp + o + k] = Ld[i * 33 + k];
      }
    }
    if (wave < nb_rem) {                                        // (A2) P_I = A[I, s] * X_D^T, one block per wave
      const int ri = 32 * (s + 1 + wave);
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      lds_mfma32(acc, S + ri * LD + o, LD, S + o * LD + o, LD, 1.f, lane);
      acc_store(acc, S + ri * LD + o, LD, lane);
    }
    __syncthreads();
    POTRF_STAMP(4 + 3 * s);
    {                                                           // (A3) S[I,K] -= P_I P_K^T, s < K <= I
      const int ntile = nb_rem * (nb_rem + 1) / 2;
      if (wave < ntile) {
        int t = wave, K = 0;
        while (t >= nb_rem - K) { t -= nb_rem - K; ++K; }
        const int I = K + t;                                    // relative indices, I >= K
        const int ri = 32 * (s + 1 + I), rk = 32 * (s + 1 + K);
        f32x16 acc;
        acc_load(acc, S + ri * LD + rk, LD, lane);
        lds_mfma32(acc, S + ri * LD + o, LD, S + rk * LD + o, LD, -1.f, lane);
        acc_store(acc, S + ri * LD + rk, LD, lane);
      }
      if (wave >= 6) {                                          // L[I, s] of this column of sub-blocks -> A (final since A2)
        for (int idx = (wave - 6) * 64 + lane; idx < nb_rem * 32 * 32; idx += 128) {
          const int i = 32 * (s + 1) + (idx >> 5), k = o + (idx & 31);
          Ak[(long)i * Cp + k] = S[i * LD + k];
        }
      }
    }
    __syncthreads();
    POTRF_STAMP(5 + 3 * s);
  }

  // ---------------------------------- inverse ----------------------------------
  // S now holds: X_D on the four diagonal 32-blocks (exact zeros above their diagonals), L below them, junk above.
  if (wave < 2) {                                               // X[2p+1, 2p] = -X_C * (L_CA * X_A), 32x32 blocks
    const int a0 = 64 * wave, c0 = a0 + 32;
    f32x16 T;
#pragma unroll
    for (int e = 0; e < 16; ++e) T[e] = 0.f;
    lds_mfma32_bn(T, S + c0 * LD + a0, LD, S + a0 * LD + a0, LD, 1.f, lane);   // T = L_CA * X_A
    // X = -X_C * T with T straight from the accumulator registers: MFMA step t takes, in lane half h,
    // the k index  r(t, h) = (t & 3) + 8 * (t >> 2) + 4 * h  -- the row of T that register t holds
    f32x16 X;
#pragma unroll
    for (int e = 0; e < 16; ++e) X[e] = 0.f;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int k = (t & 3) + 8 * (t >> 2) + 4 * h;
      const float av = -S[(c0 + r) * LD + c0 + k];
      X = __builtin_amdgcn_mfma_f32_32x32x2f32(av, T[t], X, 0, 0, 0);
    }
    acc_store(X, S + c0 * LD + a0, LD, lane);                   // over L_CA (this wave is its only reader)
  } else if (wave < 4) {                                        // the blocks above: zero (X_A, X_C are read as 64x64 below)
    const int a0 = 64 * (wave - 2);
    for (int idx = lane; idx < 32 * 32; idx += 64) S[(a0 + (idx >> 5)) * LD + a0 + 32 + (idx & 31)] = 0.f;
  }
  __syncthreads();
  if (wave < 4) {                                               // T[64x64] = L[64:128, 0:64] * X[0:64, 0:64]
    const int I = wave >> 1, J = wave & 1;
    f32x16 T;
#pragma unroll
    for (int e = 0; e < 16; ++e) T[e] = 0.f;
#pragma unroll
    for (int K = 0; K < 2; ++K)
      lds_mfma32_bn(T, S + (64 + 32 * I) * LD + 32 * K, LD, S + (32 * K) * LD + 32 * J, LD, 1.f, lane);
    acc_store(T, Tt + (32 * I) * LD + 32 * J, LD, lane);
  }
  __syncthreads();
  if (wave < 4) {                                               // X[64:128, 0:64] = -X[64:128, 64:128] * T
    const int I = wave >> 1, J = wave & 1;
    f32x16 X;
#pragma unroll
    for (int e = 0; e < 16; ++e) X[e] = 0.f;
#pragma unroll
    for (int K = 0; K < 2; ++K)
      lds_mfma32_bn(X, S + (64 + 32 * I) * LD + 64 + 32 * K, LD, Tt + (32 * K) * LD + 32 * J, LD, -1.f, lane);
    acc_store(X, S + (64 + 32 * I) * LD + 32 * J, LD, lane);    // over L[64:128, 0:64] (dead since the barrier)
  }
  __syncthreads();
  POTRF_STAMP(15);
  float* Xk = Linv + (long)kb * NB * Cp + (long)kb * NB;
#pragma unroll 8
  for (int idx = tid; idx < NB * NB; idx += 512) {
    const int i = idx >> 7, k = idx & 127;
    const float x = (k <= i) ? S[i * LD + k] : 0.f;
    if (WT) __hip_atomic_store(Xk + (long)i * Cp + k, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else Xk[(long)i * Cp + k] = x;
  }
  POTRF_STAMP(16);
}
constexpr int POTRF_LDS_FLOATS = NB * (NB + 1) + 32 * 33 + 32 + 32 + 4;
constexpr size_t POTRF_LDS = sizeof(float) * POTRF_LDS_FLOATS;

__global__ __launch_bounds__(512) void potrf_inv_diag_kernel(float* __restrict__ A, float* __restrict__ Linv,
                                                             int Cp, int kb, int32_t* __restrict__ info) {
  critical_path_priority();
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  POTRF_STAMP(0);
  potrf_load_block(dsm, A + (long)kb * NB * Cp + (long)kb * NB, Cp, threadIdx.x);
  potrf_inv_diag_block<false>(dsm, A, Linv, Cp, kb, info, threadIdx.x);
}


// Panel:  P <- P * inv(L_kk)^T  for the block column kb below the diagonal (rows (kb+1)*128 ...), IN PLACE.
// One workgroup per 64 rows of a block: it forms both 64 x 64 halves of its 64 x 128 result before it stores either
// (its rows are read by nobody else), i.e. half the serial MFMA time of one workgroup per 128 x 128 block.
__global__ __launch_bounds__(GEMM_THREADS) void panel_kernel(float* __restrict__ A,
                                                             const float* __restrict__ Linv, int Cp, int kb) {
  critical_path_priority();
  __shared__ __attribute__((aligned(16))) float smem[GEMM64X2_LDS_FLOATS];
  const int tm = kb + 1 + (blockIdx.x >> 1), sm = blockIdx.x & 1;
  float* P = A + ((long)tm * NB + 64 * sm) * Cp + (long)kb * NB;
  const float* D = Linv + (long)kb * NB * Cp + (long)kb * NB;
  Operand<float> a{P, Cp, 1, 64, true};
  Operand<float> b0{D, Cp, 1, 64, true};
  Operand<float> b1{D + 64L * Cp, Cp, 1, 64, true};
  f32x16 acc0, acc1;
  // both halves in ONE pass over P; inv(L_kk) is lower triangular: its first 64 rows end at k = 64
  gemm_acc64x2<float, float, true, true>(a, b0, b1, 0, 64, NB, smem, acc0, acc1);   // (ends with a barrier: all reads of P are done)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  tile_epilogue64(acc0, Epilogue{P, Cp, 1, EPI_STORE, TRI_ALL, 0.f, 0.f}, 64, 64, wave >> 1, wave & 1, lane);
  tile_epilogue64(acc1, Epilogue{P + 64, Cp, 1, EPI_STORE, TRI_ALL, 0.f, 0.f}, 64, 64, wave >> 1, wave & 1, lane);
}

#ifdef GPTQ_DIAG   // (measured slower as a stand-alone step, GPTQ_CHOL_SUPER: diagnostic library only)
// Panel of a whole OUTER panel [p0, p0 + nb) for the rows below it (row blocks >= p0 + nb), one workgroup per 64 rows:
// blocked forward substitution  P_c = (A_c - sum_{j < c} P_j L[p0 + c, p0 + j]^T) inv(L_cc)^T,  c = 0 .. nb - 1,  IN PLACE.
// The row slabs are independent (they only read the diagonal super-block's factor and the inverses of its diagonal
// blocks), so the four panel solves and the three rank-128 updates that a right-looking outer panel spends on these
// rows -- seven latency-bound launches -- become one, with K up to 384 per product instead of 128.
__global__ __launch_bounds__(GEMM_THREADS) void panel_super_kernel(float* __restrict__ A, const float* __restrict__ Linv,
                                                                   int Cp, int p0, int nb, int row_blk0) {
  critical_path_priority();
  __shared__ __attribute__((aligned(16))) float smem[GEMM64X2_LDS_FLOATS];
  const long r0 = (long)row_blk0 * NB + 64L * blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = 0; c < nb; ++c) {
    float* P = A + r0 * Cp + (long)(p0 + c) * NB;
    f32x16 acc0, acc1;
    if (c > 0) {                                               // A_c -= [P_0 .. P_{c-1}] * L[p0 + c, p0 .. p0 + c)^T
      const float* Arow = A + r0 * Cp + (long)p0 * NB;
      const float* Lrow = A + (long)(p0 + c) * NB * Cp + (long)p0 * NB;
      Operand<float> a{Arow, Cp, 1, 64, true};
      Operand<float> b0{Lrow, Cp, 1, 64, true};
      Operand<float> b1{Lrow + 64L * Cp, Cp, 1, 64, true};
      float old0[16], old1[16];
      const Epilogue e0{P, Cp, 1, EPI_SUB, TRI_ALL, 0.f, 0.f}, e1{P + 64, Cp, 1, EPI_SUB, TRI_ALL, 0.f, 0.f};
      tile_load_old64(old0, e0, 64, 64, wave >> 1, wave & 1, lane);
      tile_load_old64(old1, e1, 64, 64, wave >> 1, wave & 1, lane);
      gemm_acc64x2<float, float, true, true>(a, b0, b1, 0, c * NB, c * NB, smem, acc0, acc1);
      tile_finish64(acc0, old0, e0, 64, 64, wave >> 1, wave & 1, lane);
      tile_finish64(acc1, old1, e1, 64, 64, wave >> 1, wave & 1, lane);
      __syncthreads();                                         // the slab's column c is up to date for every wave
    }
    const float* D = Linv + (long)(p0 + c) * NB * Cp + (long)(p0 + c) * NB;
    Operand<float> a{P, Cp, 1, 64, true};
    Operand<float> b0{D, Cp, 1, 64, true};
    Operand<float> b1{D + 64L * Cp, Cp, 1, 64, true};
    gemm_acc64x2<float, float, true, true>(a, b0, b1, 0, 64, NB, smem, acc0, acc1);   // (ends with a barrier: all reads of P are done)
    tile_epilogue64(acc0, Epilogue{P, Cp, 1, EPI_STORE, TRI_ALL, 0.f, 0.f}, 64, 64, wave >> 1, wave & 1, lane);
    tile_epilogue64(acc1, Epilogue{P + 64, Cp, 1, EPI_STORE, TRI_ALL, 0.f, 0.f}, 64, 64, wave >> 1, wave & 1, lane);
    __syncthreads();                                           // P_c is visible to the waves that read it as an operand next
  }
}
#endif

// Trailing update:  A[m][n] -= sum_{k in [k0, k0 + K)} L[m][k] L[n][k]  on the lower tiles (tm >= tn) of the block
// columns tn in [tn0, tn1), all block rows down to nblk.  L's columns k0 .. k0 + K are final (panel solved).
// Two levels, like the column loop's trailing updates: after every 128-column step only the rest of the current
// OUTER panel of CSUPER block columns is updated (K = 128, critical path); once per outer panel everything beyond
// it gets ONE rank-(CSUPER * 128) update.
constexpr int CSUPER = 4;
__global__ __launch_bounds__(GEMM_THREADS) void syrk_kernel(float* __restrict__ A, int Cp, int nblk, int k0, int K,
                                                            int tn0, int tn1) {
  // 64 x 64 output tiles (gemm2_f32.h): blockIdx.y selects the quarter (sm, sn) of a 128 x 128 block.  Measured on
  // the launches of one step (31 ... 255 blocks): up to 2x faster than one workgroup per block, bit-identical.
  critical_path_priority();
  __shared__ __attribute__((aligned(16))) float smem[GEMM64_LDS_FLOATS];
  int rest = blockIdx.x, tn = tn0;             // column block tn, row block tm >= tn
  while (rest >= nblk - tn) { rest -= nblk - tn; ++tn; }
  if (tn >= tn1) return;
  const int tm = tn + rest;
  const int sm = blockIdx.y >> 1, sn = blockIdx.y & 1;
  const bool diag_blk = tm == tn;
  if (diag_blk && sm < sn) return;             // strictly upper quarter of a diagonal block
  const long r0 = (long)tm * NB + 64 * sm, c0 = (long)tn * NB + 64 * sn;
  Operand<float> a{A + r0 * Cp + k0, Cp, 1, 64, true};
  Operand<float> b{A + c0 * Cp + k0, Cp, 1, 64, true};
  float* Ct = A + r0 * Cp + c0;
  const bool diag = diag_blk && sm == sn;
  gemm_tile64<float, float, true, true>(a, b, 0, K, smem,
                                        Epilogue{Ct, Cp, 1, EPI_SUB, diag ? TRI_LOWER : TRI_ALL, 0.f, 0.f});
}
// The same update with one workgroup per 128 x 128 block (67 KB of LDS, two per compute unit): for the FAR updates on the
// helper stream.  Same rate at K = 512 (95 TFLOP/s), but one finishing workgroup frees a slot any kernel of the
// caller's stream fits into; with four 35 KB workgroups per compute unit the 66 KB diagonal factorization waited for
// two of them to end together (measured 116 us per diagonal block instead of 40 at C = 16384).
__global__ __launch_bounds__(GEMM_THREADS) void syrk128_kernel(float* __restrict__ A, int Cp, int nblk, int k0, int K,
                                                               int tn0, int tn1) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM_LDS_FLOATS];
  int rest = blockIdx.x, tn = tn0;
  while (rest >= nblk - tn) { rest -= nblk - tn; ++tn; }
  if (tn >= tn1) return;
  const int tm = tn + rest;
  const long r0 = (long)tm * NB, c0 = (long)tn * NB;
  Operand<float> a{A + r0 * Cp + k0, Cp, 1, NB, true};
  Operand<float> b{A + c0 * Cp + k0, Cp, 1, NB, true};
  gemm_tile<float, float, true, true, true>(a, b, 0, K, smem,
                                            Epilogue{A + r0 * Cp + c0, Cp, 1, EPI_SUB, tm == tn ? TRI_LOWER : TRI_ALL, 0.f, 0.f});
}
static inline int syrk_tiles(int nblk, int tn0, int tn1) {       // sum_{tn in [tn0, tn1)} (nblk - tn)
  const int n = tn1 - tn0;
  return n <= 0 ? 0 : n * nblk - (tn0 + tn1 - 1) * n / 2;
}

// ---------------------------------------------------------------------------------------------
// ONE launch per outer panel [p0, p1) of the factorization (round 3) instead of three per 128-column step: the diagonal
// chain, the panel solves of all rows below and the rank-128 updates inside the outer panel are ROLES of one resident
// grid, handed from role to role through flags in device memory.
//   workgroup 0 ("chain"): for kb = p0 .. p1 - 1: factorize + invert the diagonal block (potrf_inv_diag_block), publish
//     D = step kb; in front of every block but the first, its OWN look-ahead from the inverse it still holds in LDS: the
//     panel block P = A[kb, kb-1] inv(L)^T of its next row block and the update of its next diagonal tile
//     D = A[kb, kb] - P P^T (it waits only until the two slabs of row block kb have applied step kb - 2, which they did
//     one diagonal block ago), and it publishes P as those two slabs' panel pieces.
//   workgroup 1 + j ("slabs" j, j + nwg, ...; slab s = 64 rows: row block p0 + 1 + s / 2, half s & 1): for every step kb
//     at least two row blocks above its rows (row block kb + 1 is the chain's at step kb): wait for D, solve its 64 x 128
//     piece of the panel in place (panel_kernel's product), publish it if its rows lie inside the outer panel (their
//     pieces are the B operands of everybody's updates), then apply the rank-128 update to its rows of the block columns
//     kb + 1 .. p1 - 1 (syrk_kernel's 64 x 64 tiles) as soon as the pieces of those block columns are published, and
//     publish "done" (the chain's go-ahead two steps later).
// What this buys is look-ahead without any hand-off on the critical path: between two diagonal blocks the chain does
// two small products on its own compute unit, everything else happens underneath the next diagonal block.
// Same device functions, same ascending k order as the separate launches: the factor is bit-identical.
// Hand-offs (MI355X_MICROARCH.md, inter-workgroup visibility): handed-off bytes leave with write-through `sc1` stores,
// every storing wave drains vmcnt, workgroup barrier, one relaxed agent-scope flag store; the consumer polls relaxed
// from one lane, then ONE agent-scope acquire + vmcnt(0) + workgroup barrier, then plain loads.  Flags are monotonic
// (8 * launch index + step + 1; zeroed once per factorization), nobody ever waits for a slab below the outer panel, and
// every wait is bounded: a workgroup that gives up raises `abort` (every poller sees it) and reports info = -9.
// Residency: a waited-for workgroup (chain, slabs of the outer panel: the 7 lowest block indices) never waits for a
// workgroup above itself in dispatch order, and the grid is capped well under what the idle chip holds.
// ---------------------------------------------------------------------------------------------
struct PanelArgs {
  float* A;
  float* Linv;
  int32_t* info;
  int* flags;        // [0] abort, [1] D, [4 + 2 s] panel piece of slab s stored, [5 + 2 s] slab s done with the step
  int Cp, nblk, p0, p1, base, nwg;
};
constexpr int PANEL_FLAG_SLAB0 = 4;
#ifdef GPTQ_DIAG   // s_memtime stamps of the chain (0..15) and of slabs 0 / 1 (16..47 / 48..79) in the launch p0 == panel_stamp_p0
__device__ unsigned long long panel_stamps[96];
__device__ int panel_stamp_p0;
__device__ int panel_fault;   // fault injection (tests/test_gpu_parity.py): 1 = the chain never publishes its second block
#define PANEL_STAMP(i) do { if (threadIdx.x == 0 && a.p0 == panel_stamp_p0) panel_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PANEL_STAMP(i) do { } while (0)
#endif
constexpr int PANEL_SPIN_LIMIT = 1 << 21;                // polls of ~1 us each

// all live threads of the workgroup; true = go on, false = somebody gave up
__device__ __forceinline__ bool panel_wait(const PanelArgs& a, const int* f0, const int* f1, int val, int* lds_word) {
  if (threadIdx.x == 0) {
    int it = 0, ok = 1;
    while (__hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < val ||
           __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < val) {
      __builtin_amdgcn_s_sleep(2);
      if ((++it & 31) == 0) {
        if (__hip_atomic_load(a.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
        if (it > PANEL_SPIN_LIMIT) {
          __hip_atomic_store(a.flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (a.info) atomicCAS(a.info, 0, -9);
          ok = 0;
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *lds_word = ok;
  }
  __syncthreads();
  const int ok = *lds_word;
  __syncthreads();                                        // (the word may be rewritten by the next wait)
  return ok != 0;
}
// all live threads of the workgroup, after their `sc1` stores
__device__ __forceinline__ void panel_publish(int* flag, int val) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(512) void chol_panel_kernel(PanelArgs a) {
  critical_path_priority();
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  int* lds_word = reinterpret_cast<int*>(dsm + POTRF_LDS_FLOATS);
  const int Cp = a.Cp;
  if (blockIdx.x == 0) {                                    // ---- the chain ----
    constexpr int LD = NB + 1;
    float* S = dsm;
#pragma unroll 1
    for (int kb = a.p0; kb < a.p1; ++kb) {
      // (the thread index is laundered once per step: left alone hipcc hoists the diagonal block's per-lane LDS addresses
      //  out of this loop and spills them -- 270 registers, reloaded inside the 32-step chains)
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63, r = lane & 31, h = lane >> 5;
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      PANEL_STAMP(4 * (kb - a.p0));
      if (kb == a.p0) {
        PANEL_STAMP(1);
        potrf_load_block(dsm, a.A + (long)kb * NB * Cp + (long)kb * NB, Cp, tid);
      } else {
        // Look-ahead inside the chain: S still holds X = inv(L_(kb-1)), so this workgroup solves ITS next panel block
        // P = A[kb, kb-1] X^T and updates ITS next diagonal tile D = A[kb, kb] - P P^T itself -- no hand-off on the
        // critical path.  The slabs of row block kb brought both blocks up to step kb - 2 (their "done" flags).
        if (kb > a.p0 + 1) {
          const int s0 = 2 * (kb - a.p0 - 1);
          if (!panel_wait(a, a.flags + PANEL_FLAG_SLAB0 + 2 * s0 + 1, a.flags + PANEL_FLAG_SLAB0 + 2 * s0 + 3,
                          a.base + (kb - 2 - a.p0) + 1, lds_word))
            return;
        }
        PANEL_STAMP(4 * (kb - a.p0) + 1);
        float* Pg = a.A + (long)kb * NB * Cp + (long)(kb - 1) * NB;     // block (kb, kb-1), in place
        const float* Dg = a.A + (long)kb * NB * Cp + (long)kb * NB;     // tile (kb, kb), lower triangle
        // (a) P: wave w forms the 32 x 32 tiles (I, N) of rows I = w >> 1 with N in {0, 3} or {1, 2} (X is lower
        //     triangular: tile (I, N) runs over the k tiles 0 .. N -- 80 MFMAs per wave either way).  A's fragments come
        //     straight from L2: lane (r, h) needs A[32 I + r][2 kk + h], i.e. two of every four consecutive floats.
        const int I = wave >> 1;
        const int N0 = (wave & 1) ? 1 : 0, N1 = (wave & 1) ? 2 : 3;
        float4 af[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) af[j] = *reinterpret_cast<const float4*>(Pg + (long)(32 * I + r) * Cp + 4 * j);
        // the old values of the diagonal tile's lower 32 x 32 tiles: tile t of 10 -> (TI, TK), wave w takes t = w and, for
        // w < 2, t = 8 + w
        auto tile_ik = [](int t, int& ti, int& tk) { ti = 0; while (t > ti) { t -= ti + 1; ++ti; } tk = t; };
        int TI0, TK0, TI1 = 0, TK1 = 0;
        tile_ik(wave, TI0, TK0);
        const bool two = wave < 2;
        if (two) tile_ik(8 + wave, TI1, TK1);
        float old0[16], old1[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          old0[e] = Dg[(long)(32 * TI0 + row) * Cp + 32 * TK0 + r];
          old1[e] = two ? Dg[(long)(32 * TI1 + row) * Cp + 32 * TK1 + r] : 0.f;
        }
        f32x16 pa0, pa1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { pa0[e] = 0.f; pa1[e] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          if (kt <= N1) {
            const float* X1 = S + (32 * N1 + r) * LD + 32 * kt + h;
            const float* X0 = S + (32 * N0 + r) * LD + 32 * kt + h;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
              const float4 v = af[8 * kt + (kk >> 1)];
              const float av = (kk & 1) ? (h ? v.w : v.z) : (h ? v.y : v.x);
              if (kt <= N0) pa0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, X0[2 * kk], pa0, 0, 0, 0);
              pa1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, X1[2 * kk], pa1, 0, 0, 0);
            }
          }
        }
        __syncthreads();                                       // every wave is done with X
        // (b) P into S (the operand image of the update) and, write-through, in place into A (the slabs' B operand)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = 32 * I + (e & 3) + 8 * (e >> 2) + 4 * h;
          const float v0 = pa0[e], v1 = pa1[e];                  // (scalars first: a vector ELEMENT handed to a builtin that
                                                               //  bit-casts it reads element 0 -- hipcc, ROCm 7.2)
          S[row * LD + 32 * N0 + r] = v0;
          S[row * LD + 32 * N1 + r] = v1;
          __hip_atomic_store(Pg + (long)row * Cp + 32 * N0 + r, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(Pg + (long)row * Cp + 32 * N1 + r, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        {                                                      // = the panel pieces of the two slabs of row block kb
          const int sk = 2 * (kb - a.p0 - 1), pv = a.base + (kb - 1 - a.p0) + 1;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();                                     // (also: P is complete in S)
          if (tid == 0) {
            __hip_atomic_store(a.flags + PANEL_FLAG_SLAB0 + 2 * sk, pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.flags + PANEL_FLAG_SLAB0 + 2 * sk + 2, pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        // (c) D = old - P P^T on the lower tiles: from zero over k = 0 .. 127 ascending, then old - acc (syrk_kernel's EPI_SUB)
        f32x16 d0, d1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { d0[e] = 0.f; d1[e] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          lds_mfma32(d0, S + 32 * TI0 * LD + 32 * kt, LD, S + 32 * TK0 * LD + 32 * kt, LD, 1.f, lane);
          if (two) lds_mfma32(d1, S + 32 * TI1 * LD + 32 * kt, LD, S + 32 * TK1 * LD + 32 * kt, LD, 1.f, lane);
        }
        __syncthreads();                                       // every wave is done with P
#pragma unroll
        for (int e = 0; e < 16; ++e) {                         // the mirrored block, as the load phase leaves it
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          const float v0 = old0[e] - d0[e];
          if (TI0 > TK0 || row >= r) {
            S[(32 * TI0 + row) * LD + 32 * TK0 + r] = v0;
            S[(32 * TK0 + r) * LD + 32 * TI0 + row] = v0;
          }
          if (two) {
            const float v1 = old1[e] - d1[e];
            if (TI1 > TK1 || row >= r) {
              S[(32 * TI1 + row) * LD + 32 * TK1 + r] = v1;
              S[(32 * TK1 + r) * LD + 32 * TI1 + row] = v1;
            }
          }
        }
      }
      potrf_inv_diag_block<true>(dsm, a.A, a.Linv, Cp, kb, a.info, tid);   // (starts with a barrier)
      PANEL_STAMP(4 * (kb - a.p0) + 2);
#ifdef GPTQ_DIAG
      if (panel_fault == 1 && kb == a.p0 + 1) return;            // (the chain is gone: the slabs' bounded waits must end the launch)
#endif
      panel_publish(a.flags + 1, a.base + (kb - a.p0) + 1);
      PANEL_STAMP(4 * (kb - a.p0) + 3);
    }
    return;
  }
  if (threadIdx.x >= GEMM_THREADS) return;                  // ---- slabs: four waves (the tile code's geometry) ----
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nslab = 2 * (a.nblk - a.p0 - 1);
  for (int kb = a.p0; kb < a.p1; ++kb) {
    const int val = a.base + (kb - a.p0) + 1;
    bool have_d = false;
    for (int s = blockIdx.x - 1; s < nslab; s += a.nwg) {
      const int tm = a.p0 + 1 + (s >> 1), sm = s & 1;
      if (tm <= kb || (tm == kb + 1 && tm < a.p1)) continue;   // the chain's: factorized, or its look-ahead at this step
#ifdef GPTQ_DIAG
      const int sb = (s < 2) ? 16 + 32 * s + 8 * (kb - a.p0) : 88;   // (slabs 0 and 1 are stamped)
#define SLAB_STAMP(i) PANEL_STAMP(sb + (s < 2 ? (i) : 0))
#else
#define SLAB_STAMP(i) do { } while (0)
#endif
      SLAB_STAMP(0);
      if (!have_d) {
        if (!panel_wait(a, a.flags + 1, a.flags + 1, val, lds_word)) return;
        have_d = true;
      }
      SLAB_STAMP(1);
      const bool inside = tm < a.p1;                        // rows of the outer panel: their results are handed on
      float* P = a.A + ((long)tm * NB + 64 * sm) * Cp + (long)kb * NB;
      {                                                     // panel piece: P <- P inv(L_kk)^T (panel_kernel)
        const float* D = a.Linv + (long)kb * NB * Cp + (long)kb * NB;
        Operand<float> pa{P, Cp, 1, 64, true};
        Operand<float> b0{D, Cp, 1, 64, true};
        Operand<float> b1{D + 64L * Cp, Cp, 1, 64, true};
        f32x16 acc0, acc1;
        gemm_acc64x2<float, float, true, true>(pa, b0, b1, 0, 64, NB, dsm, acc0, acc1);
        Epilogue e0{P, Cp, 1, EPI_STORE, TRI_ALL, 0.f, 0.f}, e1{P + 64, Cp, 1, EPI_STORE, TRI_ALL, 0.f, 0.f};
        e0.wt = e1.wt = inside;
        tile_epilogue64(acc0, e0, 64, 64, wave >> 1, wave & 1, lane);
        tile_epilogue64(acc1, e1, 64, 64, wave >> 1, wave & 1, lane);
      }
      SLAB_STAMP(2);
      if (inside) panel_publish(a.flags + PANEL_FLAG_SLAB0 + 2 * s, val);
      else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // (own P is an operand next)
      SLAB_STAMP(3);
      const int c_last = min(a.p1 - 1, tm);
      for (int c = kb + 1; c <= c_last; ++c) {              // A[slab, c] -= P_slab P_c^T (syrk_kernel's tiles)
        const int sc = 2 * (c - a.p0 - 1);                  // the slabs of row block c
        if (c < tm || sm == 1) {                            // (a diagonal block's upper slab needs only its own piece)
          const int* f0 = a.flags + PANEL_FLAG_SLAB0 + 2 * sc;
          const int* f1 = (c == tm) ? f0 : f0 + 2;          // (own piece: published above)
          if (!panel_wait(a, f0, f1, val, lds_word)) return;
        }
        if (c == kb + 1) SLAB_STAMP(4);
        for (int sn = 0; sn < 2; ++sn) {
          if (c == tm && sn > sm) continue;                 // strictly upper quarter of a diagonal block
          Operand<float> ua{P, Cp, 1, 64, true};
          Operand<float> ub{a.A + ((long)c * NB + 64 * sn) * Cp + (long)kb * NB, Cp, 1, 64, true};
          Epilogue ep{a.A + ((long)tm * NB + 64 * sm) * Cp + (long)c * NB + 64 * sn, Cp, 1, EPI_SUB,
                      (c == tm && sn == sm) ? TRI_LOWER : TRI_ALL, 0.f, 0.f};
          ep.wt = inside;
          gemm_tile64<float, float, true, true>(ua, ub, 0, NB, dsm, ep);
        }
      }
      SLAB_STAMP(5);
      if (inside) panel_publish(a.flags + PANEL_FLAG_SLAB0 + 2 * s + 1, val);
      SLAB_STAMP(6);
    }
  }
}
constexpr size_t PANEL_LDS = POTRF_LDS + 16;
static_assert(PANEL_LDS >= sizeof(float) * GEMM64X2_LDS_FLOATS, "the slab role's tile code needs its LDS too");
static inline int panel_slab_wgs(int nblk, int p0) {
  static const int cap = [] { const char* e = getenv("GPTQ_CHOL_WGS"); return e ? atoi(e) : 160; }();
  const int nslab = 2 * (nblk - p0 - 1);
  return std::max(std::min(nslab, std::max(cap, 8)), 0);
}
static inline int chol_persist() {
  static const int v = [] { const char* e = getenv("GPTQ_CHOL_PERSIST"); return e ? atoi(e) : 1; }();
  return v;
}
static int launch_panel(float* A, float* Linv, int Cp, int nblk, int p0, int p1, int32_t* info, int* flags,
                        hipStream_t s) {
  GPTQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&chol_panel_kernel),      // (per device: every call)
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)PANEL_LDS));
  PanelArgs pa{A, Linv, info, flags, Cp, nblk, p0, p1, 8 * (p0 / CSUPER), panel_slab_wgs(nblk, p0)};
  chol_panel_kernel<<<1 + pa.nwg, 512, PANEL_LDS, s>>>(pa);
  return GPTQ_OK;
}
static inline size_t panel_flag_ints(int nblk) { return PANEL_FLAG_SLAB0 + 4 * (size_t)nblk + 64; }

// Recursive-doubling inverse, level with segment size s (in 128-blocks).  For every pair
// (A = blocks [2ps, 2ps+s), Cc = blocks [2ps+s, 2ps+2s) clipped) the off-diagonal block of the
// inverse is X = -Cc^-1 * B * A^-1 with B = L[Cc, A].
//   step 1:  T = B * A^-1, stored TRANSPOSED in the (unused) upper triangle of Linv;
//   step 2:  X = -Cc^-1 * T  into the lower triangle of Linv.
__global__ __launch_bounds__(GEMM_THREADS) void trtri_step1_kernel(const float* __restrict__ L,
                                                                   float* __restrict__ Linv, int Cp,
                                                                   int nblk, int s, int p0) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM64_LDS_FLOATS];
  // tiles with the longest K range (small tj) come first in dispatch order: the level ends without a long tail;
  // blockIdx.z = quarter (sm, sn) of the 128 x 128 block (64 x 64 output tiles, gemm2_f32.h)
  const int p = blockIdx.y + p0, tj = blockIdx.x / s, ti = blockIdx.x % s;
  const int a0 = 2 * p * s, cb0 = a0 + s;
  if (cb0 + ti >= nblk) return;
  const int sm = blockIdx.z >> 1, sn = blockIdx.z & 1;
  const long rm = (long)(cb0 + ti) * NB + 64 * sm, cn = (long)(a0 + tj) * NB + 64 * sn, ka = (long)a0 * NB;
  Operand<float> a{L + rm * Cp + ka, Cp, 1, 64, true};            // B[m][k]
  Operand<float> b{Linv + ka * Cp + cn, 1, Cp, 64, true};         // Ainv[k][n], lower: k >= n
  float* Tt = Linv + cn * Cp + rm;                                // T^T lives at [n][m]
  // T^T[n][m] = sum_k Ainv[k][n] * B[m][k]: the operands swap roles so that the tile comes out already transposed
  // and its rows (m contiguous) are stored coalesced; products commute, so the bits are those of B * Ainv
  gemm_tile64<float, float, false, true>(b, a, tj * NB + 64 * sn, s * NB, smem,
                                         Epilogue{Tt, Cp, 1, EPI_STORE, TRI_ALL, 0.f, 0.f});
}

__global__ __launch_bounds__(GEMM_THREADS) void trtri_step2_kernel(float* __restrict__ Linv, int Cp,
                                                                   int nblk, int s, int p0) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM64_LDS_FLOATS];
  // K = (ti + 1) * 128: the bottom rows first; blockIdx.z = quarter (sm, sn) of the 128 x 128 block
  const int p = blockIdx.y + p0, ti = s - 1 - blockIdx.x / s, tj = blockIdx.x % s;
  const int a0 = 2 * p * s, cb0 = a0 + s;
  if (cb0 + ti >= nblk) return;
  const int sm = blockIdx.z >> 1, sn = blockIdx.z & 1;
  const long rm = (long)(cb0 + ti) * NB + 64 * sm, cn = (long)(a0 + tj) * NB + 64 * sn, kc = (long)cb0 * NB;
  Operand<float> a{Linv + rm * Cp + kc, Cp, 1, 64, true};         // Cinv[m][k], lower: k <= m
  Operand<float> b{Linv + cn * Cp + kc, Cp, 1, 64, true};         // T[k][n] read from T^T[n][k]
  float* X = Linv + rm * Cp + cn;
  gemm_tile64<float, float, true, true>(a, b, 0, ti * NB + 64 * (sm + 1), smem,
                                        Epilogue{X, Cp, 1, EPI_STORE_NEG, TRI_ALL, 0.f, 0.f});
}

// U[i][j] = Linv[C-1-i][C-1-j] for j >= i, zero below the diagonal.
__global__ __launch_bounds__(256) void flip_to_upper_kernel(const float* __restrict__ Linv, int Cp, int C,
                                                            float* __restrict__ U, int ldu) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= C) return;
  U[(long)i * ldu + j] = (j >= i) ? Linv[(long)(C - 1 - i) * Cp + (C - 1 - j)] : 0.f;
}

// gptq_rfactor_upper's output (C % 128 == 0; L-space block (tm, tk), tm > tk, maps to U-space block
// (nblk-1-tm, nblk-1-tk), element (r, c) to (C-1-r, C-1-c)):
//   * off-diagonal blocks: Rt[B, blk] = R[B, blk] * U_blk,blk  with R = J L J and U_kk = R_kk^-1 = J L_kk^-1 J, i.e.
//     the flipped product L[tm, tk] * L_kk^-1 -- one 64 x 64 tile per workgroup, stored through negative strides;
//   * diagonal blocks: U_kk (flipped L_kk^-1), zero under their diagonals.
__global__ __launch_bounds__(GEMM_THREADS) void rtilde_kernel(const float* __restrict__ L,
                                                              const float* __restrict__ Linv, int Cp, int nblk, int C,
                                                              float* __restrict__ H, int ldh) {
  __shared__ __attribute__((aligned(16))) float smem[GEMM64_LDS_FLOATS];
  int rest = blockIdx.x, tk = 0;                 // strictly lower block (tm, tk), column by column
  while (rest >= nblk - 1 - tk) { rest -= nblk - 1 - tk; ++tk; }
  const int tm = tk + 1 + rest;
  const int sm = blockIdx.y >> 1, sn = blockIdx.y & 1;
  const long r0 = (long)tm * NB + 64 * sm, kb = (long)tk * NB, c0 = kb + 64 * sn;
  Operand<float> a{L + r0 * Cp + kb, Cp, 1, 64, true};            // L[r0 + m][kb + k]
  Operand<float> b{Linv + kb * Cp + c0, 1, Cp, 64, true};         // Linv_kk[k][n], lower: k >= n
  float* Ht = H + (long)(C - 1 - r0) * ldh + (C - 1 - c0);
  gemm_tile64<float, float, true, false>(a, b, 64 * sn, NB, smem, Epilogue{Ht, -ldh, -1, EPI_STORE, TRI_ALL, 0.f, 0.f});
}
__global__ __launch_bounds__(256) void flip_diag_kernel(const float* __restrict__ Linv, int Cp, int C,
                                                        float* __restrict__ H, int ldh) {
  const int i = blockIdx.x * 2 + (threadIdx.x >> 7);             // U-space row
  const int j = (i / NB) * NB + (threadIdx.x & 127);             // the columns of its diagonal block
  if (i >= C) return;
  H[(long)i * ldh + j] = (j < i) ? 0.f : Linv[(long)(C - 1 - i) * Cp + (C - 1 - j)];
}

}  // namespace gptq

using namespace gptq;

static inline int padded(int C) { return cdiv(C, NB) * NB; }

extern "C" size_t gptq_hinv_workspace_bytes(int C) {
  if (C <= 0) return 0;
  const size_t Cp = padded(C);
  Carver cv(nullptr);
  cv.take<float>(Cp * Cp);   // Abar / L
  cv.take<float>(Cp * Cp);   // Linv (+ T^T in its upper triangle)
  cv.take<float>(64);        // damp
  cv.take<int>(panel_flag_ints((int)(Cp / NB)));   // hand-off flags of chol_panel_kernel
  return cv.used();
}

static int factor_chain(float* H, int ldh, int C, float percdamp, const int32_t* perm, int32_t* info, void* workspace,
                        size_t workspace_bytes, gptq_stream_t stream, bool rfactor) {
  GPTQ_CHECK_ARG(H && workspace, "gptq_hinv_upper: null pointer");
  GPTQ_CHECK_ARG(C > 0 && ldh >= C, "gptq_hinv_upper: bad sizes");
  GPTQ_CHECK_ARG(workspace_bytes >= gptq_hinv_workspace_bytes(C), "gptq_hinv_upper: workspace too small");
  GPTQ_CHECK_ARG(reinterpret_cast<uintptr_t>(workspace) % 256 == 0, "gptq_hinv_upper: workspace must be 256-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int Cp = padded(C), nblk = Cp / NB;
  Carver cv(workspace);
  float* A = cv.take<float>((size_t)Cp * Cp);
  float* Linv = cv.take<float>((size_t)Cp * Cp);
  float* damp = cv.take<float>(64);
  int* flags = cv.take<int>(panel_flag_ints(nblk));

  diag_mean_kernel<<<1, 256, 0, s>>>(H, ldh, C, percdamp, damp, info);
  const bool persist = chol_persist() != 0;
  if (persist) GPTQ_CHECK_HIP(hipMemsetAsync(flags, 0, sizeof(int) * panel_flag_ints(nblk), s));
  const bool lds_gather = perm && C % 4 == 0 && ldh % 4 == 0 && C <= 36864 && reinterpret_cast<uintptr_t>(H) % 16 == 0;
  if (lds_gather) {
    if (int rc = gptq_symmetrize(H, ldh, C, stream)) return rc;   // (H is consumed: its lower triangle is free)
    GPTQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&build_abar_perm_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * C)));
    build_abar_perm_kernel<<<std::min(Cp, 2048), 512, sizeof(float) * C, s>>>(H, ldh, C, Cp, perm, damp, A);
  } else {
    build_abar_kernel<<<dim3(cdiv(Cp, 256), Cp), 256, 0, s>>>(H, ldh, C, Cp, perm, damp, A);
  }
  // Look-ahead: the far update of an outer panel is split into "the next outer panel's block columns" (caller's stream:
  // the factorization needs them next) and "everything beyond" (helper stream, underneath the next outer panel's steps,
  // which are serial and latency-bound).
  SideCtx* sc = (lookahead_mask() & 1) ? side_ctx(s) : nullptr;
  bool side_busy = false;
  GPTQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_inv_diag_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTRF_LDS));
  for (int p0 = 0; p0 < nblk; p0 += CSUPER) {
    const int p1 = std::min(p0 + CSUPER, nblk);
#ifdef GPTQ_DIAG
    static const int super_env = tune_knob("GPTQ_CHOL_SUPER", 0);   // (measured: 4096 x 4096 3.82 -> 4.23 ms -- the same small launches plus one long one)
    if (super_env) {
      // the diagonal super-block [p0, p1) alone: the right-looking steps restricted to ITS rows ...
      for (int kb = p0; kb < p1; ++kb) {
        potrf_inv_diag_kernel<<<1, 512, POTRF_LDS, s>>>(A, Linv, Cp, kb, info);
        if (kb + 1 < p1) {
          panel_kernel<<<2 * (p1 - kb - 1), GEMM_THREADS, 0, s>>>(A, Linv, Cp, kb);
          syrk_kernel<<<dim3(syrk_tiles(p1, kb + 1, p1), 4), GEMM_THREADS, 0, s>>>(A, Cp, p1, kb * NB, NB, kb + 1, p1);
        }
      }
      // ... and ONE launch for the rows below it
      if (p1 < nblk) panel_super_kernel<<<2 * (nblk - p1), GEMM_THREADS, 0, s>>>(A, Linv, Cp, p0, p1 - p0, p1);
    } else
#endif
    if (persist) {
      if (int rc = launch_panel(A, Linv, Cp, nblk, p0, p1, info, flags, s)) return rc;
    } else {
    for (int kb = p0; kb < p1; ++kb) {
      potrf_inv_diag_kernel<<<1, 512, POTRF_LDS, s>>>(A, Linv, Cp, kb, info);
      const int nrem = nblk - kb - 1;
      if (nrem <= 0) break;
      panel_kernel<<<2 * nrem, GEMM_THREADS, 0, s>>>(A, Linv, Cp, kb);
      if (kb + 1 < p1)                                             // the rest of this outer panel: rank-128
        syrk_kernel<<<dim3(syrk_tiles(nblk, kb + 1, p1), 4), GEMM_THREADS, 0, s>>>(A, Cp, nblk, kb * NB, NB, kb + 1, p1);
    }
    }
    if (p1 < nblk) {                                               // everything beyond: rank-((p1 - p0) * 128)
      if (side_busy) {                                             // the previous far update wrote these tiles too
        GPTQ_CHECK_HIP(hipStreamWaitEvent(s, sc->side_done, 0));
        side_busy = false;
      }
      const int q1 = std::min(p1 + CSUPER, nblk);
      syrk_kernel<<<dim3(syrk_tiles(nblk, p1, q1), 4), GEMM_THREADS, 0, s>>>(A, Cp, nblk, p0 * NB, (p1 - p0) * NB, p1, q1);
      if (q1 < nblk) {
        hipStream_t ts = s;
        if (sc) {
          GPTQ_CHECK_HIP(hipEventRecord(sc->main_done, s));
          GPTQ_CHECK_HIP(hipStreamWaitEvent(sc->stream, sc->main_done, 0));
          ts = sc->stream;
        }
        syrk128_kernel<<<syrk_tiles(nblk, q1, nblk), GEMM_THREADS, 0, ts>>>(A, Cp, nblk, p0 * NB, (p1 - p0) * NB, q1, nblk);
        if (sc) {
          GPTQ_CHECK_HIP(hipEventRecord(sc->side_done, sc->stream));
          side_busy = true;
        }
      }
    }
  }
  if (side_busy) GPTQ_CHECK_HIP(hipStreamWaitEvent(s, sc->side_done, 0));
  if (rfactor) {                                                 // no triangular inverse: half the chain's flops
    if (nblk > 1)
      rtilde_kernel<<<dim3(nblk * (nblk - 1) / 2, 4), GEMM_THREADS, 0, s>>>(A, Linv, Cp, nblk, C, H, ldh);
    flip_diag_kernel<<<cdiv(C, 2), 256, 0, s>>>(Linv, Cp, C, H, ldh);
    GPTQ_CHECK_LAUNCH("gptq_rfactor_upper");
    return GPTQ_OK;
  }
  for (int sz = 1; sz < nblk; sz *= 2) {
    const int pairs = cdiv(nblk, 2 * sz);
    trtri_step1_kernel<<<dim3(sz * sz, pairs, 4), GEMM_THREADS, 0, s>>>(A, Linv, Cp, nblk, sz, 0);
    trtri_step2_kernel<<<dim3(sz * sz, pairs, 4), GEMM_THREADS, 0, s>>>(Linv, Cp, nblk, sz, 0);
  }
  flip_to_upper_kernel<<<dim3(cdiv(C, 256), C), 256, 0, s>>>(Linv, Cp, C, H, ldh);
  GPTQ_CHECK_LAUNCH("gptq_hinv_upper");
  return GPTQ_OK;
}

#ifdef GPTQ_DIAG
extern "C" int gptq_diag_chain64_stamps(unsigned long long* out4) {   // the last 64-tile launch of THIS file
  GPTQ_CHECK_HIP(hipMemcpyFromSymbol(out4, HIP_SYMBOL(gemm64_stamps), sizeof(unsigned long long) * 4));
  return GPTQ_OK;
}
extern "C" int gptq_diag_potrf_ablate(int v) {
  GPTQ_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(potrf_ablate), &v, sizeof(int)));
  return GPTQ_OK;
}
extern "C" int gptq_diag_panel_fault(int v) {
  GPTQ_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(panel_fault), &v, sizeof(int)));
  return GPTQ_OK;
}
extern "C" int gptq_diag_panel_stamps(unsigned long long* out96, int p0) {   // read the last stamps, select the next launch
  GPTQ_CHECK_HIP(hipMemcpyFromSymbol(out96, HIP_SYMBOL(panel_stamps), sizeof(unsigned long long) * 96));
  GPTQ_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(panel_stamp_p0), &p0, sizeof(int)));
  return GPTQ_OK;
}
extern "C" int gptq_diag_potrf_stamps(unsigned long long* out17) {
  GPTQ_CHECK_HIP(hipMemcpyFromSymbol(out17, HIP_SYMBOL(potrf_stamps), sizeof(unsigned long long) * 17));
  return GPTQ_OK;
}
#endif

// ---------------------------------------------------------------------------------------------
// The factor-form chain in pieces, for a factorization whose OUTER PANELS are spread over several GPUs (the reference has
// no counterpart: SURVEY 8e; gptq_amd/parallel.py::rfactor_sharded drives them).  All ranks hold the same H and call
// `begin`; outer panel j (block columns 4j .. 4j + 3) belongs to one rank, which factorizes it with `panel` once its
// columns carry the updates of all earlier panels, and broadcasts it (the host's collective library); every rank applies
// `update` to the block columns IT owns; `end` forms Rt / U_kk in H like gptq_rfactor_upper.  With one rank the sequence
// begin, {panel(j), update(j, everything beyond)}..., end IS gptq_rfactor_upper (same kernels, no helper stream).
// Workspace: gptq_hinv_workspace_bytes(C); A = [Cp, Cp] floats at offset 0, the diagonal blocks' inverses in Linv =
// [Cp, Cp] floats at offset Cp * Cp * 4 (Cp = C rounded up to 128: both offsets are 256-byte aligned).
// ---------------------------------------------------------------------------------------------
static int chol_ws(void* workspace, int C, float** A, float** Linv, float** damp, int** flags = nullptr) {
  GPTQ_CHECK_ARG(workspace && C > 0 && C % NB == 0, "gptq_chol_*: C must be a positive multiple of 128");
  GPTQ_CHECK_ARG(reinterpret_cast<uintptr_t>(workspace) % 256 == 0, "gptq_chol_*: workspace must be 256-byte aligned");
  Carver cv(workspace);
  *A = cv.take<float>((size_t)C * C);
  *Linv = cv.take<float>((size_t)C * C);
  *damp = cv.take<float>(64);
  if (flags) *flags = cv.take<int>(panel_flag_ints(C / NB));
  return GPTQ_OK;
}

extern "C" int gptq_chol_begin(float* H, int ldh, int C, float percdamp, const int32_t* perm, int32_t* info,
                               void* workspace, size_t workspace_bytes, gptq_stream_t stream) {
  GPTQ_CHECK_ARG(H && ldh >= C, "gptq_chol_begin: bad arguments");
  GPTQ_CHECK_ARG(workspace_bytes >= gptq_hinv_workspace_bytes(C), "gptq_chol_begin: workspace too small");
  float *A, *Linv, *damp;
  int* flags;
  if (int rc = chol_ws(workspace, C, &A, &Linv, &damp, &flags)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  GPTQ_CHECK_HIP(hipMemsetAsync(flags, 0, sizeof(int) * panel_flag_ints(C / NB), s));
  diag_mean_kernel<<<1, 256, 0, s>>>(H, ldh, C, percdamp, damp, info);
  build_abar_kernel<<<dim3(cdiv(C, 256), C), 256, 0, s>>>(H, ldh, C, C, perm, damp, A);
  GPTQ_CHECK_LAUNCH("gptq_chol_begin");
  return GPTQ_OK;
}

extern "C" int gptq_chol_panel(void* workspace, int C, int p0, int32_t* info, gptq_stream_t stream) {
  float *A, *Linv, *damp;
  int* flags;
  if (int rc = chol_ws(workspace, C, &A, &Linv, &damp, &flags)) return rc;
  const int nblk = C / NB;
  GPTQ_CHECK_ARG(p0 >= 0 && p0 < nblk && p0 % CSUPER == 0, "gptq_chol_panel: p0 must be the first block of an outer panel");
  hipStream_t s = static_cast<hipStream_t>(stream);
  GPTQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_inv_diag_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)POTRF_LDS));
  const int p1 = std::min(p0 + CSUPER, nblk);
  if (chol_persist()) {
    if (int rc = launch_panel(A, Linv, C, nblk, p0, p1, info, flags, s)) return rc;
    GPTQ_CHECK_LAUNCH("gptq_chol_panel");
    return GPTQ_OK;
  }
  for (int kb = p0; kb < p1; ++kb) {
    potrf_inv_diag_kernel<<<1, 512, POTRF_LDS, s>>>(A, Linv, C, kb, info);
    const int nrem = nblk - kb - 1;
    if (nrem <= 0) break;
    panel_kernel<<<2 * nrem, GEMM_THREADS, 0, s>>>(A, Linv, C, kb);
    if (kb + 1 < p1)
      syrk_kernel<<<dim3(syrk_tiles(nblk, kb + 1, p1), 4), GEMM_THREADS, 0, s>>>(A, C, nblk, kb * NB, NB, kb + 1, p1);
  }
  GPTQ_CHECK_LAUNCH("gptq_chol_panel");
  return GPTQ_OK;
}

extern "C" int gptq_chol_update(void* workspace, int C, int p0, int tn0, int tn1, gptq_stream_t stream) {
  float *A, *Linv, *damp;
  if (int rc = chol_ws(workspace, C, &A, &Linv, &damp)) return rc;
  const int nblk = C / NB;
  const int p1 = std::min(p0 + CSUPER, nblk);
  GPTQ_CHECK_ARG(p0 >= 0 && p0 < nblk && p0 % CSUPER == 0 && tn0 >= p1 && tn1 <= nblk, "gptq_chol_update: bad block range");
  if (tn0 >= tn1) return GPTQ_OK;
  syrk128_kernel<<<syrk_tiles(nblk, tn0, tn1), GEMM_THREADS, 0, static_cast<hipStream_t>(stream)>>>(
      A, C, nblk, p0 * NB, (p1 - p0) * NB, tn0, tn1);
  GPTQ_CHECK_LAUNCH("gptq_chol_update");
  return GPTQ_OK;
}

extern "C" int gptq_chol_end(float* H, int ldh, int C, void* workspace, gptq_stream_t stream) {
  GPTQ_CHECK_ARG(H && ldh >= C, "gptq_chol_end: bad arguments");
  float *A, *Linv, *damp;
  if (int rc = chol_ws(workspace, C, &A, &Linv, &damp)) return rc;
  const int nblk = C / NB;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (nblk > 1) rtilde_kernel<<<dim3(nblk * (nblk - 1) / 2, 4), GEMM_THREADS, 0, s>>>(A, Linv, C, nblk, C, H, ldh);
  flip_diag_kernel<<<cdiv(C, 2), 256, 0, s>>>(Linv, C, C, H, ldh);
  GPTQ_CHECK_LAUNCH("gptq_chol_end");
  return GPTQ_OK;
}

extern "C" int gptq_hinv_upper(float* H, int ldh, int C, float percdamp, const int32_t* perm,
                               int32_t* info, void* workspace, size_t workspace_bytes,
                               gptq_stream_t stream) {
  return factor_chain(H, ldh, C, percdamp, perm, info, workspace, workspace_bytes, stream, false);
}

extern "C" int gptq_rfactor_upper(float* H, int ldh, int C, float percdamp, const int32_t* perm,
                                  int32_t* info, void* workspace, size_t workspace_bytes,
                                  gptq_stream_t stream) {
  GPTQ_CHECK_ARG(C > 0 && C % NB == 0, "gptq_rfactor_upper: C must be a positive multiple of 128");
  return factor_chain(H, ldh, C, percdamp, perm, info, workspace, workspace_bytes, stream, true);
}
