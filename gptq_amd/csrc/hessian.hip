// Hessian running mean  H <- H*n/(n+b) + 2/(n+b) * X^T X   (replaces gptq.py:38-65).
//
// X [tokens, C] arrives in the model dtype (fp16/bf16/fp32) and is widened to fp32
// on load (gptq.py:62).  The product is an exact-fp32 MFMA SYRK: only the upper
// triangle tiles (ti <= tj) are formed -- H is symmetric -- which halves the
// dominant FLOP term of the whole pipeline; gptq_symmetrize mirrors it on demand.
#include <stdlib.h>

#include <algorithm>

#include "gemm_f32.h"

namespace gptq {

template <typename T>
__global__ __launch_bounds__(GEMM_THREADS) void hessian_kernel(float* __restrict__ H, int ldh,
                                                               const T* __restrict__ X, int ldx, int C,
                                                               int tokens, float alpha, float beta,
                                                               bool vec) {
  // upper-triangle tile pair from a linear id (row-major over ti <= tj)
  const int nt = (C + GBM - 1) / GBM;
  int rest = blockIdx.x, ti = 0;
  while (rest >= nt - ti) { rest -= nt - ti; ++ti; }
  const int tj = ti + rest;

  __shared__ __attribute__((aligned(16))) float smem[GEMM_LDS_FLOATS];
  Operand<T> a{X + (long)ti * GBM, 1, ldx, min(GBM, C - ti * GBM), vec};
  Operand<T> b{X + (long)tj * GBN, 1, ldx, min(GBN, C - tj * GBN), vec};
  float* Ht = H + (long)ti * GBM * ldh + (long)tj * GBN;
  const bool diag = ti == tj;
  gemm_tile<T, T, false, false>(a, b, 0, tokens, smem,
                                Epilogue{Ht, ldh, 1, EPI_AXPBY, diag ? TRI_UPPER : TRI_ALL, alpha, beta});
}

// ---------------------------------------------------------------------------------------------
// fp16 / bf16 activations: the products x_si * x_sj of two 11-bit (8-bit) significands are EXACT in
// fp32, so the 16x-faster f16/bf16 MFMA with fp32 accumulation loses nothing against the reference's
// fp32 matmul of the widened inputs (gptq.py:62-65) -- it only rounds the running sum, like any fp32
// GEMM does.  The 2/n scaling is applied once per element in the epilogue.
//
// 128x128 tile, BK = 64 tokens per stage, v_mfma_f32_32x32x16_{f16,bf16}.  X tiles sit in LDS exactly
// as in HBM ([token][channel], channel contiguous); the MFMA wants 8 consecutive k (tokens) per lane,
// which is what ds_read_b64_tr_b16 (gfx950's transposing LDS read) delivers: per 16-lane group a
// 4-token x 16-channel block, column-major.  Row stride 320 B puts the 4 rows x 2 groups of a half-wave
// on 8 disjoint 32-byte bank slots (conflict-free).
// ---------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HBK = 64;                 // tokens per stage
constexpr int HROW = 160;               // LDS row stride in 16-bit elements (128 + 32 pad = 320 B)
constexpr int HTILE = HBK * HROW;       // elements per operand stage

template <bool BF16>
__device__ __forceinline__ f32x16 mfma16(s16x8 a, s16x8 b, f32x16 c) {
  if (BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// this thread's 4 x 16-byte pieces of a [HBK tokens][128 channels] slab starting at token k0
__device__ __forceinline__ void h16_load(const unsigned short* __restrict__ X, long ldx, int rem, bool vec,
                                         int k0, int tokens, uint4 r[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int v = tid + 256 * e;
    const int row = v >> 4, c8 = (v & 15) * 8;
    const int k = k0 + row;
    const unsigned short* p = X + (long)k * ldx + c8;
    if (k < tokens && vec && c8 + 8 <= rem) {
      r[e] = *reinterpret_cast<const uint4*>(p);
    } else {
      unsigned short t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = (k < tokens && c8 + j < rem) ? p[j] : (unsigned short)0;
      r[e] = make_uint4(t[0] | (t[1] << 16), t[2] | (t[3] << 16), t[4] | (t[5] << 16), t[6] | (t[7] << 16));
    }
  }
}

__device__ __forceinline__ void h16_store(unsigned short* S, const uint4 r[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int v = tid + 256 * e;
    *reinterpret_cast<uint4*>(S + (v >> 4) * HROW + (v & 15) * 8) = r[e];
  }
}

// Upper-triangle tile (ti <= tj) for workgroup `bid`, chosen so that the tiles running together on
// one XCD share operand panels in that XCD's L2.  Workgroups are dealt round-robin to the 8 XCDs
// (bid % 8 labels the XCD group; speed only, never correctness); each group gets a contiguous run of
// a supertile-ordered enumeration (8 x 8 tiles per supertile, diagonal supertiles triangular).
__device__ __forceinline__ void hessian_tile_of(int bid, int nt, int& ti, int& tj) {
  constexpr int SUP = 8;
  const int T = nt * (nt + 1) / 2;
  const int xcd = bid & 7, q = T >> 3, r = T & 7;
  int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  int si = 0, rows = min(SUP, nt);
  for (;; ++si) {
    rows = min(SUP, nt - si * SUP);
    const int cnt = rows * (rows + 1) / 2 + rows * (nt - si * SUP - rows);
    if (v < cnt) break;
    v -= cnt;
  }
  const int dcnt = rows * (rows + 1) / 2;
  if (v < dcnt) {
    int rr = 0;
    while (v >= rows - rr) { v -= rows - rr; ++rr; }
    ti = si * SUP + rr;
    tj = ti + v;
  } else {
    v -= dcnt;
    const int sjo = v / (rows * SUP);
    const int col0 = (si + 1 + sjo) * SUP;
    const int cols = min(SUP, nt - col0);
    const int rem = v - sjo * rows * SUP;
    ti = si * SUP + rem / cols;
    tj = col0 + rem % cols;
  }
}

template <bool BF16>
__global__ __launch_bounds__(256) void hessian16_kernel(float* __restrict__ H, int ldh,
                                                        const unsigned short* __restrict__ X, int ldx, int C,
                                                        int tokens, float alpha, float beta, bool vec) {
  extern __shared__ __attribute__((aligned(16))) unsigned short hsm[];   // [2 buffers][A, B][HTILE]
  const int nt = (C + GBM - 1) / GBM;
  int ti, tj;
  hessian_tile_of(blockIdx.x, nt, ti, tj);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int rem_a = min(GBM, C - ti * GBM), rem_b = min(GBN, C - tj * GBN);
  const unsigned short* Xa = X + (long)ti * GBM;
  const unsigned short* Xb = X + (long)tj * GBN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // transposed-read address of this lane inside a 16-token k-step: row 8h + q, channel 16g + 4p
  const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  const int frag_off = (8 * h + q) * HROW + 16 * g + 4 * p;

  uint4 ra[4], rb[4];
  const int nk = (tokens + HBK - 1) / HBK;
  h16_load(Xa, ldx, rem_a, vec, 0, tokens, ra);
  h16_load(Xb, ldx, rem_b, vec, 0, tokens, rb);
  h16_store(hsm, ra);
  h16_store(hsm + HTILE, rb);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      h16_load(Xa, ldx, rem_a, vec, (kt + 1) * HBK, tokens, ra);
      h16_load(Xb, ldx, rem_b, vec, (kt + 1) * HBK, tokens, rb);
    }
    const unsigned short* As = hsm + cur * 2 * HTILE + wm * 64 + frag_off;
    const unsigned short* Bs = hsm + cur * 2 * HTILE + HTILE + wn * 64 + frag_off;
#pragma unroll
    for (int kk = 0; kk < HBK / 16; ++kk) {
      s16x8 fa[2], fb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const unsigned short* pa = As + kk * 16 * HROW + t * 32;
        const unsigned short* pb = Bs + kk * 16 * HROW + t * 32;
        const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
        const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 4 * HROW));
        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb));
        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + 4 * HROW));
        fa[t] = s16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        fb[t] = s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      }
      acc[0][0] = mfma16<BF16>(fa[0], fb[0], acc[0][0]);
      acc[0][1] = mfma16<BF16>(fa[0], fb[1], acc[0][1]);
      acc[1][0] = mfma16<BF16>(fa[1], fb[0], acc[1][0]);
      acc[1][1] = mfma16<BF16>(fa[1], fb[1], acc[1][1]);
    }
    if (more) {
      h16_store(hsm + (cur ^ 1) * 2 * HTILE, ra);
      h16_store(hsm + (cur ^ 1) * 2 * HTILE + HTILE, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  float* Ht = H + (long)ti * GBM * ldh + (long)tj * GBN;
  tile_epilogue(acc, Epilogue{Ht, ldh, 1, EPI_AXPBY, ti == tj ? TRI_UPPER : TRI_ALL, alpha, beta}, rem_a, rem_b, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------
// Deep-prefetch variant for aligned shapes (tokens % 64 == 0, C % 128 == 0, 16-byte aligned rows).
// The register-staged kernel above keeps ONE stage in flight, which leaves every 64-token stage
// exposed to HBM / Infinity-Cache latency (measured: ~3.9K cycles per stage against 512 cycles of
// MFMA).  Here X streams global -> LDS with LDS-DMA (global_load_lds_dwordx4, no VGPRs) into a ring of
// RING stages, three stages ahead of the matrix cores; counted `s_waitcnt vmcnt(N)` + one raw
// s_barrier per stage.
//
// LDS image per operand stage: 64 token rows x 256 B, unpadded (an LDS-DMA wave-instruction writes 1 KiB
// contiguously = 4 rows), XOR-swizzled in 16-byte chunks: physical chunk = logical ^ ((row & 3) << 2).
// The swizzle lives on the per-lane SOURCE address of the DMA and on the transposed read, and makes
// ds_read_b64_tr_b16 conflict-free: the 4 rows x 4 chunks a half-wave touches land on 16 distinct chunks.
// ---------------------------------------------------------------------------------------------
constexpr int RING_DEFAULT = 4;
constexpr int MAX_XLIST = 16;
struct XList { const unsigned short* p[MAX_XLIST]; };   // up to 16 equally shaped activation slabs per launch
constexpr int MAX_PROB = 8;
// several independent (H, slabs) problems of identical shape in ONE launch: the Linears of a block that
// share in_features (q,k,v,out,fc1 ...) each bring only C/128*(C/128+1)/2 tiles, too few to fill 256 CUs alone
struct ProbGroup {
  float* H[MAX_PROB]; XList x[MAX_PROB]; float alpha[MAX_PROB]; float beta[MAX_PROB];
  // the 256 x 256 kernels also take problems of DIFFERENT width in one launch (fc2 together with q/k/v/out/fc1 of
  // the block): per-problem C, leading dimensions and the first tile id of each problem
  int C[MAX_PROB]; int ldx[MAX_PROB]; int ldh[MAX_PROB]; int tile_start[MAX_PROB + 1]; int n_prob;
};
__device__ __forceinline__ int prob_of_tile(const ProbGroup& pg, int tile) {
  int p = 0;
  while (p + 1 < pg.n_prob && tile >= pg.tile_start[p + 1]) ++p;
  return p;
}
constexpr int DSTAGE = 2 * HBK * 256;        // bytes per ring slot: A + B, 64 rows x 256 B each

__device__ __forceinline__ void dma_stage(const unsigned short* __restrict__ Xa, const unsigned short* __restrict__ Xb,
                                          long ldx, int k0, char* slot, int wave, int lane) {
  // loader wave w moves token rows [16w, 16w+16) of both operands: 4 + 4 one-KiB pieces
  const int rl = lane >> 4;                                   // row inside the 4-row piece
  const int chunk = (lane & 15) ^ (rl << 2);                  // logical 16-byte chunk this lane fetches
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = 16 * wave + 4 * u;
    const long goff = (long)(k0 + r + rl) * ldx + 8 * chunk;
    char* dst = slot + r * 256;                               // wave-uniform; the DMA adds lane * 16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xa + goff),
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Xb + goff),
                                     (__attribute__((address_space(3))) void*)(dst + HBK * 256), 16, 0, 0);
  }
}

typedef s16x4 frag_t;
__device__ __forceinline__ s16x8 join8(frag_t lo, frag_t hi) {
  return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// one transposed 4-token x 16-channel read; OFF is an immediate byte offset
#define TR_READ(dst, addr, OFF) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
// the 8 reads of k-step KK: A tile 0 (rows +0, +4), A tile 1, B tile 0, B tile 1
#define TR_READ_STEP(F, KK)                         \
  do {                                              \
    TR_READ(F[0], aa0, (KK) * 4096);                \
    TR_READ(F[1], aa0, (KK) * 4096 + 1024);         \
    TR_READ(F[2], aa1, (KK) * 4096);                \
    TR_READ(F[3], aa1, (KK) * 4096 + 1024);         \
    TR_READ(F[4], ab0, (KK) * 4096);                \
    TR_READ(F[5], ab0, (KK) * 4096 + 1024);         \
    TR_READ(F[6], ab1, (KK) * 4096);                \
    TR_READ(F[7], ab1, (KK) * 4096 + 1024);         \
  } while (0)

// Workgroup = 8 waves: waves 0-3 own the MFMA quadrants, waves 4-7 only issue LDS-DMA (an LDS-DMA
// piece costs its wave ~100-185 cycles of issue; interleaved with the MFMAs it serialised them).
template <bool BF16, int ABLATE = 0, int RING = RING_DEFAULT>   // ABLATE (diagnostic builds only): 1 = no MFMA side, 2 = no DMA
__global__ __launch_bounds__(512) void hessian16_dma_kernel(ProbGroup pg, int tiles_per_prob, int ldh, int nx,
                                                            int ldx, int C, int tokens) {
  // `tokens` rows per slab, `nx` slabs: the K loop walks all of them (one H update for the whole batch)
  const int prob = blockIdx.x / tiles_per_prob;               // workgroup-uniform problem index
  float* __restrict__ H = pg.H[prob];
  const XList& xl = pg.x[prob];
  const float alpha = pg.alpha[prob], beta = pg.beta[prob];
  extern __shared__ __attribute__((aligned(1024))) char ring[];          // RING x DSTAGE, the ONLY LDS object
  const int nt = C / GBM;
  int ti, tj;
  hessian_tile_of(blockIdx.x - prob * tiles_per_prob, nt, ti, tj);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;                              // wave-uniform role
  const int wm = (wave >> 1) & 1, wn = wave & 1;
  const int spk = tokens / HBK;                               // stages per slab
  const int nk = spk * nx;
  const long goffa = (long)ti * GBM, goffb = (long)tj * GBN;
  auto issue = [&](int st, int lw) {
    const int sl = st / spk;                                  // slab of this stage (wave-uniform)
    const unsigned short* X = xl.p[sl];
    dma_stage(X + goffa, X + goffb, ldx, (st - sl * spk) * HBK, ring + (st % RING) * DSTAGE, lw, lane);
  };

  if (loader) {
    const int lw = wave - 4;
#pragma unroll
    for (int st = 0; st < RING - 1; ++st)
      if (st < nk) issue(st, lw);
    for (int kt = 0; kt < nk; ++kt) {
      const int ahead = min(RING - 2, nk - 1 - kt);           // stages issued after stage kt (uniform)
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                           // stage kt landed; slot (kt-1) % RING is free
      if (ABLATE != 2 && kt + RING - 1 < nk) issue(kt + RING - 1, lw);
    }
    return;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // per-lane byte offsets of the transposed reads (row 8h + q of a 16-token k-step, m/n-tile t)
  const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  int offa[2], offb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ca = 8 * wm + 4 * t + 2 * g + (p >> 1), cb = 8 * wn + 4 * t + 2 * g + (p >> 1);
    offa[t] = (8 * h + q) * 256 + 16 * (ca ^ (q << 2)) + 8 * (p & 1);
    offb[t] = (8 * h + q) * 256 + 16 * (cb ^ (q << 2)) + 8 * (p & 1) + HBK * 256;
  }

  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)ring);
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_barrier();                             // pairs with the loaders' barrier of stage kt
    if (ABLATE == 1) continue;
    // Fragment reads go through inline asm: hipcc treats every LDS read as aliasing the in-flight
    // LDS-DMA and would drain it with vmcnt(0).  We order them ourselves: the counted vmcnt + barrier
    // above cover the DMA, lgkmcnt(0) + sched_barrier cover the reads (the MFMAs are register-only and
    // would otherwise be hoisted above the wait).  Reads of k-step kk+1 fly under the MFMAs of kk.
    const unsigned sbase = lds0 + (kt % RING) * DSTAGE;
    const unsigned aa0 = sbase + offa[0], aa1 = sbase + offa[1], ab0 = sbase + offb[0], ab1 = sbase + offb[1];
    frag_t f[3][8];                                            // fragment reads run TWO k-steps ahead
    TR_READ_STEP(f[0], 0);
    TR_READ_STEP(f[1], 1);
#pragma unroll
    for (int kk = 0; kk < HBK / 16; ++kk) {
      // LDS reads return in order: allow the 8 reads of k-step kk+1 to stay in flight
      if (kk < HBK / 16 - 1) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 0) { TR_READ_STEP(f[2], 2); }
      else if (kk == 1) { TR_READ_STEP(f[0], 3); }
      const frag_t* c = f[kk % 3];
      const s16x8 fa0 = join8(c[0], c[1]), fa1 = join8(c[2], c[3]);
      const s16x8 fb0 = join8(c[4], c[5]), fb1 = join8(c[6], c[7]);
      acc[0][0] = mfma16<BF16>(fa0, fb0, acc[0][0]);
      acc[0][1] = mfma16<BF16>(fa0, fb1, acc[0][1]);
      acc[1][0] = mfma16<BF16>(fa1, fb0, acc[1][0]);
      acc[1][1] = mfma16<BF16>(fa1, fb1, acc[1][1]);
    }
  }

  float* Ht = H + (long)ti * GBM * ldh + (long)tj * GBN;
  tile_epilogue(acc, Epilogue{Ht, ldh, 1, EPI_AXPBY, ti == tj ? TRI_UPPER : TRI_ALL, alpha, beta}, GBM, GBN, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------
// 256 x 256 tiles for C % 256 == 0 (every hidden size of the OPT / LLaMA families).  Why: a CU pulls
// operands from L2 into LDS at ~40 B/clk (fill_rate microbench; 64 B/clk is the vector-L1 ceiling), and a
// 128 x 128 tile needs 64 B/clk to keep the f16 MFMAs busy (8 KB per 16-token k-step = 4 MFMAs x 32 clk per
// SIMD), so the kernel above tops out near 30 % of the matrix peak however deep its ring is.  A 256 x 256
// tile needs 32 B/clk.  8 waves as 2 (M) x 4 (N), 128 x 64 outputs each (128 accumulator registers, two waves
// per SIMD so one wave's LDS-DMA issue / fragment waits sit under the other's MFMAs); every wave issues its
// own 4 LDS-DMA pieces per 32-token stage; ring of 4 stages x 32 KB.  A stage is four [32 tokens][128
// channels] images (A lo/hi, B lo/hi) in the same XOR-swizzled layout as above, so the fragment addressing
// is unchanged: 12 transposed reads feed 8 MFMAs per 16-token k-step.
// The K order per output element is the same as in the 128 x 128 kernel, so both produce identical bits.
// ---------------------------------------------------------------------------------------------
constexpr int BT = 256;                       // tile edge
constexpr int BBK = 32;                       // tokens per stage
constexpr int BHALF = BBK * 256;              // bytes of one [32 tokens][128 channels] image
constexpr int BSTAGE = 4 * BHALF;             // A lo, A hi, B lo, B hi
constexpr int BRING_DEFAULT = 4;

#define BSB __builtin_amdgcn_sched_barrier(0)
#ifdef GPTQ_DIAG   // the 32x32x16 twin of the 256 x 256 kernel (hessian16_big_kernel): diagnostic library only
#define BTR_A(F, KK)                                         \
  do {                                                       \
    TR_READ(F[0], a0, (KK) * 4096);  TR_READ(F[1], a0, (KK) * 4096 + 1024);   \
    TR_READ(F[2], a1, (KK) * 4096);  TR_READ(F[3], a1, (KK) * 4096 + 1024);   \
    TR_READ(F[4], a2, (KK) * 4096);  TR_READ(F[5], a2, (KK) * 4096 + 1024);   \
    TR_READ(F[6], a3, (KK) * 4096);  TR_READ(F[7], a3, (KK) * 4096 + 1024);   \
  } while (0)
#define BTR_B(F, KK)                                         \
  do {                                                       \
    TR_READ(F[8], b0, (KK) * 4096);  TR_READ(F[9], b0, (KK) * 4096 + 1024);   \
    TR_READ(F[10], b1, (KK) * 4096); TR_READ(F[11], b1, (KK) * 4096 + 1024);  \
  } while (0)
// The twelve reads of a k-step in four groups of three (one group per MFMA gap: three transposed reads beside
// an MFMA are free, MI355X_MICROARCH.md "Issued between MFMAs")
#define BTR_G0(F, KK) do { TR_READ(F[0], a0, (KK) * 4096); TR_READ(F[1], a0, (KK) * 4096 + 1024); TR_READ(F[2], a1, (KK) * 4096); } while (0)
#define BTR_G1(F, KK) do { TR_READ(F[3], a1, (KK) * 4096 + 1024); TR_READ(F[4], a2, (KK) * 4096); TR_READ(F[5], a2, (KK) * 4096 + 1024); } while (0)
#define BTR_G2(F, KK) do { TR_READ(F[6], a3, (KK) * 4096); TR_READ(F[7], a3, (KK) * 4096 + 1024); TR_READ(F[8], b0, (KK) * 4096); } while (0)
#define BTR_G3(F, KK) do { TR_READ(F[9], b0, (KK) * 4096 + 1024); TR_READ(F[10], b1, (KK) * 4096); TR_READ(F[11], b1, (KK) * 4096 + 1024); } while (0)
// Eight MFMAs on the complete fragment set FC; the reads of the NEXT k-step (set FN, k-step KKN of the stage that
// a0..b1 address) ride in the first four MFMA gaps, this wave's four LDS-DMA pieces (PIECES) in the last four.
#define BSTEP(FC, FN, KKN, PIECES)                                                  \
  do {                                                                              \
    const s16x8 fb0 = join8(FC[8], FC[9]), fb1 = join8(FC[10], FC[11]);             \
    s16x8 fa = join8(FC[0], FC[1]);                                                 \
    BSB;                                                                            \
    if (ABL != 1) acc[0][0] = mfma16<BF16>(fa, fb0, acc[0][0]);                     \
    BSB; if (ABL != 1) BTR_G0(FN, KKN); BSB;                                        \
    if (ABL != 1) acc[0][1] = mfma16<BF16>(fa, fb1, acc[0][1]);                     \
    BSB; if (ABL != 1) BTR_G1(FN, KKN); BSB;                                        \
    fa = join8(FC[2], FC[3]);                                                       \
    if (ABL != 1) acc[1][0] = mfma16<BF16>(fa, fb0, acc[1][0]);                     \
    BSB; if (ABL != 1) BTR_G2(FN, KKN); BSB;                                        \
    if (ABL != 1) acc[1][1] = mfma16<BF16>(fa, fb1, acc[1][1]);                     \
    BSB; if (ABL != 1) BTR_G3(FN, KKN); BSB;                                        \
    fa = join8(FC[4], FC[5]);                                                       \
    if (ABL != 1) acc[2][0] = mfma16<BF16>(fa, fb0, acc[2][0]);                     \
    BSB; if (PIECES) piece(0); BSB;                                                 \
    if (ABL != 1) acc[2][1] = mfma16<BF16>(fa, fb1, acc[2][1]);                     \
    BSB; if (PIECES) piece(1); BSB;                                                 \
    fa = join8(FC[6], FC[7]);                                                       \
    if (ABL != 1) acc[3][0] = mfma16<BF16>(fa, fb0, acc[3][0]);                     \
    BSB; if (PIECES) piece(2); BSB;                                                 \
    if (ABL != 1) acc[3][1] = mfma16<BF16>(fa, fb1, acc[3][1]);                     \
    BSB; if (PIECES) piece(3); BSB;                                                 \
  } while (0)
#endif

// Work decomposition (data-parallel rounds + a split last round).  T tiles over P CUs leave the last round
// partly empty (fc2 of OPT-1.3b: 528 tiles = 2.06 rounds on 256 CUs, paid as 3).  The first floor(T/P)*P tiles
// are whole-K workgroups with the direct epilogue; the `left_tiles` remaining ones are cut along K: `workers`
// workgroups take equal contiguous runs of the (tile, stage) sequence -- at most two segments each, since a run is
// never longer than one tile's K -- write their fp32 partial tiles to a workspace, and hessian16_big_fixup
// adds the partials of a tile in run order (fixed order: results do not depend on timing) and applies the epilogue.
struct BigPlan {
  int dp_tiles;         // tiles done whole by workgroups [0, dp_tiles)
  int left_tiles;       // tiles cut along K
  int workers;          // workgroups [dp_tiles, dp_tiles + workers)
  int chunk;            // stages per worker run
  float* ws;            // workers x 2 partial tiles of 256 x 256 fp32
  int item0;            // first work item of THIS launch (launches by rounds, see hessian_launch_big)
  int item1;            // one past its last work item
  int head;             // > 0: heads + tails (below) instead of equal runs of the (tile, stage) sequence
};
constexpr int BTILE_FLOATS = BT * BT;

// Heads + tails.  Equal contiguous runs leave every worker of the split round at its own K offset, so nothing an XCD's
// 32 workgroups fetch is shared: at C = 11008 that round moved 5.6 GB in 0.82 ms (6.9 TB/s, HBM-bound) while a whole
// round of 256 tiles moves 2.1-2.4 GB in 0.88 ms.  With `head` = chunk the first `left_tiles` workers take stages
// [0, head) of ONE tile each -- they start together and walk K in step like a whole round -- and only the remaining
// workers take equal runs of the concatenated tails [head, nk_all); a run then touches up to BIG_MAXSEG tiles.
// Partial slots: worker w < left_tiles owns slot w, tail worker v owns slots left_tiles + BIG_MAXSEG v + seg.
constexpr int BIG_MAXSEG = 4;
// segment `seg` of split worker `wi`: left tile l, stages [s0, s1), partial slot; false = no such segment
__device__ __forceinline__ bool big_segment(const BigPlan& plan, int wi, int seg, int nk_all, int& l, int& s0, int& s1,
                                            int& slot) {
  if (plan.head == 0) {
    if (seg >= 2) return false;
    const int run_begin = wi * plan.chunk, run_end = min(run_begin + plan.chunk, plan.left_tiles * nk_all);
    l = run_begin / nk_all + seg;
    s0 = seg == 0 ? run_begin - l * nk_all : 0;
    s1 = min(run_end - l * nk_all, nk_all);
    slot = wi * 2 + seg;
    return s1 > s0;
  }
  if (wi < plan.left_tiles) {
    l = wi; s0 = 0; s1 = plan.head; slot = wi;
    return seg == 0;
  }
  const int tl = nk_all - plan.head, v = wi - plan.left_tiles;
  const int tb = v * plan.chunk, te = min(tb + plan.chunk, plan.left_tiles * tl);
  l = tb / tl + seg;
  if (l >= plan.left_tiles) return false;
  s0 = seg == 0 ? tb - l * tl : 0;
  s1 = min(te - l * tl, tl);
  slot = plan.left_tiles + v * BIG_MAXSEG + seg;
  if (s1 <= s0) return false;
  s0 += plan.head;
  s1 += plan.head;
  return true;
}
// the partial slots of left tile l, in the fixed order they are summed: calls f(slot)
template <typename F>
__device__ __forceinline__ void big_for_each_partial(const BigPlan& plan, int l, int nk_all, F f) {
  if (plan.head == 0) {
    const int first = l * nk_all, last = first + nk_all - 1;
    const int w0 = first / plan.chunk, w1 = min(last / plan.chunk, plan.workers - 1);
    for (int w = w0; w <= w1; ++w) f(w * 2 + ((w * plan.chunk < first) ? 1 : 0));   // 1: the run began in the previous tile
    return;
  }
  f(l);
  const int tl = nk_all - plan.head, ntail = plan.workers - plan.left_tiles;
  const int tb = l * tl, te = tb + tl - 1;
  const int v0 = tb / plan.chunk, v1 = min(te / plan.chunk, ntail - 1);
  for (int v = v0; v <= v1; ++v) f(plan.left_tiles + v * BIG_MAXSEG + (l - (v * plan.chunk) / tl));
}

#ifdef GPTQ_DIAG   // (the twin: measured 2-3 % behind the 16x16x32 kernel below, kept for re-measuring only)
// register (t, u, e) of wave `wave`, lane `lane`  <->  element of the 256 x 256 tile (C/D map of the 32x32 MFMA:
// col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)); partial tiles are stored in register
// order ([wave][t][u][e][lane]: every store instruction writes 256 contiguous bytes)
__device__ __forceinline__ long big_part_index(int wave, int t, int u, int e, int lane) {
  return ((((long)wave * 4 + t) * 2 + u) * 16 + e) * 64 + lane;
}

// H = alpha H + beta v for the 32-row block pair t of this wave: all old values are loaded before any store
// (a per-element read-modify-write compiles into serialized round trips)
__device__ __forceinline__ void big_epilogue_rows(float* __restrict__ H, int ldh, int ti, int tj, int wm, int wn,
                                                  int lane, int t, const float (&v)[2][16], float alpha, float beta) {
  const bool diag = ti == tj;
  const int row0 = ti * BT + wm * 128 + 4 * (lane >> 5) + t * 32;
  const int col0 = tj * BT + wn * 64 + (lane & 31);
  float old[2][16];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = row0 + (e & 3) + 8 * (e >> 2), col = col0 + u * 32;
      old[u][e] = (!diag || row <= col) ? H[(long)row * ldh + col] : 0.f;
    }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = row0 + (e & 3) + 8 * (e >> 2), col = col0 + u * 32;
      const float out = alpha * old[u][e] + beta * v[u][e];              // contraction off: fl(fl(a*h) + fl(b*v))
      if (!diag || row <= col) H[(long)row * ldh + col] = out;
    }
}

// ABL (diagnostic builds only): 1 = no fragment reads / MFMAs, 2 = no LDS-DMA.
template <bool BF16, int BRING = BRING_DEFAULT, int ABL = 0>
__global__ __launch_bounds__(512) void hessian16_big_kernel(ProbGroup pg, BigPlan plan, int nx, int tokens) {
  extern __shared__ __attribute__((aligned(1024))) char ring[];          // BRING x BSTAGE, the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int spk = tokens / BBK;                               // stages per slab
  const int nk_all = spk * nx;                                // stages of a whole tile

  // transposed-read byte offsets inside a stage (row 8h + q of a 16-token k-step); the XOR swizzle acts on
  // bits 2-3 of the chunk index, i.e. on the 32-channel block number: block t of this lane sits at (t ^ q)
  const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  const int lane_off = (8 * h + q) * 256 + 32 * g + 16 * (p >> 1) + 8 * (p & 1);
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)ring);
  const unsigned oa = lds0 + wm * BHALF + lane_off;
  const unsigned ob = lds0 + (2 + (wn >> 1)) * BHALF + lane_off;
  const unsigned oa0 = oa + 64 * (0 ^ q), oa1 = oa + 64 * (1 ^ q), oa2 = oa + 64 * (2 ^ q), oa3 = oa + 64 * (3 ^ q);
  const unsigned ob0 = ob + 64 * ((2 * (wn & 1)) ^ q), ob1 = ob + 64 * ((2 * (wn & 1) + 1) ^ q);
  // LDS-DMA: wave w fills image (w >> 1), token rows 16 (w & 1) + 4 u + (lane >> 4), u = 0..3
  const int img = wave >> 1;
  const int rl = lane >> 4;
  const int chunk16 = (lane & 15) ^ (rl << 2);                // logical 16-byte chunk this lane fetches
  const int dma_dst = img * BHALF + (16 * (wave & 1)) * 256;  // wave-uniform; the DMA adds lane * 16

  // this workgroup's segments: (tile, stages [s0, s1), partial slot or -1 for the direct epilogue)
  // work items: [0, dp_tiles) whole tiles, then `workers` K-split runs; a launch sized for a CU budget has fewer
  // workgroups than items and every workgroup strides over them
  for (int bid = plan.item0 + blockIdx.x; bid < plan.item1; bid += gridDim.x) {
  int run_begin = 0, run_end = 0;
  const bool whole = bid < plan.dp_tiles;
  if (!whole) {
    run_begin = (bid - plan.dp_tiles) * plan.chunk;
    run_end = min(run_begin + plan.chunk, plan.left_tiles * nk_all);
  }
  for (int seg = 0; seg < 2; ++seg) {
    int tile, s0, s1;
    if (whole) {
      if (seg == 1) break;
      tile = bid; s0 = 0; s1 = nk_all;
    } else {
      const int l = run_begin / nk_all + seg;
      s0 = seg == 0 ? run_begin - l * nk_all : 0;
      s1 = min(run_end - l * nk_all, nk_all);
      if (s1 <= s0) break;
      tile = plan.dp_tiles + l;
    }
    const int prob = prob_of_tile(pg, tile);
    const XList& xl = pg.x[prob];
    const int ldx = pg.ldx[prob], ldh = pg.ldh[prob];
    int ti, tj;
    hessian_tile_of(tile - pg.tile_start[prob], pg.C[prob] / BT, ti, tj);
    const int nk = s1 - s0;

    const long gcol = (img < 2 ? (long)ti * BT : (long)tj * BT) + (img & 1) * 128;
    const long lane_goff = (long)(16 * (wave & 1) + rl) * ldx + gcol + 8 * chunk16;
    // issue cursor: stages are issued strictly in order, so the source pointer of the NEXT stage is advanced
    // right after each issue (the slab-pointer load then has a whole stage to return)
    int is_slab = s0 / spk, is_in = s0 - is_slab * spk, is_st = 0;
    const unsigned short* src = xl.p[is_slab] + lane_goff + (long)is_in * BBK * ldx;
    auto piece = [&](int u) {
      if (ABL == 2) return;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)(4 * u) * ldx),
                                       (__attribute__((address_space(3))) void*)(ring + (is_st % BRING) * BSTAGE + dma_dst + u * 1024),
                                       16, 0, 0);
    };
    // Stages past the end of the segment re-load its last stage (3 redundant stage loads per segment), so that
    // the loop below is branch-free: always BRING - 1 stages in flight, one constant `vmcnt`.
    auto advance = [&]() {
      ++is_st;
      if (is_st < nk) {
        if (++is_in == spk) {
          is_in = 0;
          ++is_slab;
          src = xl.p[is_slab] + lane_goff;
        } else {
          src += (long)BBK * ldx;
        }
      }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;

#pragma unroll
    for (int st = 0; st < BRING - 1; ++st) {
#pragma unroll
      for (int u = 0; u < 4; ++u) piece(u);
      advance();
    }
    if (BRING == 5) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // stage 0 landed for every wave
#pragma unroll
    for (int u = 0; u < 4; ++u) piece(u);
    advance();

    // NOTE for maintainers: the compiler does not know these reads are asynchronous.  Nothing may touch their
    // destination registers before the covering s_waitcnt -- tools/check_async_lds.py (run by the CPU tests)
    // verifies that on the generated ISA.
    frag_t f0[12], f1[12];
    {
      const unsigned a0 = oa0, a1 = oa1, a2 = oa2, a3 = oa3, b0 = ob0, b1 = ob1;
      if (ABL != 1) { BTR_A(f0, 0); BTR_B(f0, 0); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int kt = 0; kt < nk; ++kt) {
      {
        // k-step 0 of stage kt (f0, complete) under the reads of its k-step 1
        const unsigned sb = (kt % BRING) * BSTAGE;
        const unsigned a0 = oa0 + sb, a1 = oa1 + sb, a2 = oa2 + sb, a3 = oa3 + sb, b0 = ob0 + sb, b1 = ob1 + sb;
        BSTEP(f0, f1, 1, false);
      }
      // own pieces of stage kt+1 landed (stages kt+2 .. kt+BRING-1 stay in flight); all reads of stage kt done
      if (BRING == 5) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                             // stage kt+1 visible; slot of stage kt is free
      {
        // k-step 1 of stage kt (f1, complete) under the first reads of stage kt+1 (after the last stage: harmless
        // dummy reads) and this wave's four LDS-DMA pieces of stage kt + BRING, into the slot stage kt just left
        const unsigned sn = ((kt + 1) % BRING) * BSTAGE;
        const unsigned a0 = oa0 + sn, a1 = oa1 + sn, a2 = oa2 + sn, a3 = oa3 + sn, b0 = ob0 + sn, b1 = ob1 + sn;
        BSTEP(f1, f0, 0, true);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      advance();
    }
    // the dummy reads and the redundant tail stages must be gone before registers / ring slots are reused
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();

    if (whole) {
      float* __restrict__ H = pg.H[prob];
      const float alpha = pg.alpha[prob], beta = pg.beta[prob];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float v[2][16];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 16; ++e) v[u][e] = acc[t][u][e];
        big_epilogue_rows(H, ldh, ti, tj, wm, wn, lane, t, v, alpha, beta);
      }
    } else {
      float* __restrict__ part = plan.ws + ((long)(bid - plan.dp_tiles) * 2 + seg) * BTILE_FLOATS;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 16; ++e) part[big_part_index(wave, t, u, e, lane)] = acc[t][u][e];
    }
  }
  }
}

// 32 single-wave workgroups per K-split tile (block pair row t = blockIdx.y, wave slice = blockIdx.z): sum the
// partial tiles in run order -- loads of four partials in flight, the adds stay in order -- then the usual epilogue.
__global__ __launch_bounds__(64) void hessian16_big_fixup(ProbGroup pg, BigPlan plan, int nk_all) {
  const int l = blockIdx.x;
  const int tile = plan.dp_tiles + l;
  const int prob = prob_of_tile(pg, tile);
  const int ldh = pg.ldh[prob];
  int ti, tj;
  hessian_tile_of(tile - pg.tile_start[prob], pg.C[prob] / BT, ti, tj);
  const int lane = threadIdx.x & 63, wave = blockIdx.z;
  const int wm = wave >> 2, wn = wave & 3;
  const int first = l * nk_all, last = first + nk_all - 1;
  const int w0 = first / plan.chunk, w1 = min(last / plan.chunk, plan.workers - 1);
  float* __restrict__ H = pg.H[prob];
  const float alpha = pg.alpha[prob], beta = pg.beta[prob];
  const int t = blockIdx.y;
  float v[2][16];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 16; ++e) v[u][e] = 0.f;
  auto part_of = [&](int w) {
    const int seg = (w * plan.chunk < first) ? 1 : 0;         // the run began in the previous tile
    return plan.ws + ((long)w * 2 + seg) * BTILE_FLOATS;
  };
  int w = w0;
  for (; w + 3 <= w1; w += 4) {
    float x[4][2][16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* __restrict__ part = part_of(w + j);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) x[j][u][e] = part[big_part_index(wave, t, u, e, lane)];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) v[u][e] += x[j][u][e];
  }
  for (; w <= w1; ++w) {
    const float* __restrict__ part = part_of(w);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) v[u][e] += part[big_part_index(wave, t, u, e, lane)];
  }
  big_epilogue_rows(H, ldh, ti, tj, wm, wn, lane, t, v, alpha, beta);
}
#endif   // GPTQ_DIAG

// ---------------------------------------------------------------------------------------------
// The 256 x 256 tile on v_mfma_f32_16x16x32_{f16,bf16} (it walks stages in pairs: tokens % 64 == 0; the diagnostic library
// also holds a 32x32x16 twin, GPTQ_HESS_SHAPE=32).  Cycles per flop are those of the 32x32x16 form;
// which of the two the chip clocks higher under load is an empirical matter (MI355X_MICROARCH.md "DVFS give-back"
// item 7), so both are built on the same tile and work split: this one measures 1.5-3 % faster in the bench.
// One k-step is a whole 32-token stage: A fragment of block t = channels 16t..16t+15 x tokens 8g..8g+7 for lane
// group g = lane >> 4, i.e. two transposed reads (token rows 8g + q and 8g + 4 + q).  The two groups of a
// half-wave read rows 8 apart in the same columns, so the image also XORs chunk bit 1 with token-row bit 3:
// physical chunk = logical ^ ((row & 3) << 2) ^ (((row >> 3) & 1) << 1).  In block terms the lane reads block
// (t ^ m), m = (g & 1) | (q << 1): address = lane_base ^ (32 t).
// Stage = P1 (A blocks 0-3 x B, under the reads of A blocks 4-7) | barrier | P2 (A blocks 4-7 x B, under the
// reads of the NEXT stage's A blocks 0-3 and B -- into the other B register set -- and the LDS-DMA pieces).
// ---------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool BF16>
__device__ __forceinline__ f32x4v mfma16s(s16x8 a, s16x8 b, f32x4v c) {
  if (BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ long big16_part_index(int wave, int t, int u, int e, int lane) {
  return ((((long)wave * 8 + t) * 4 + u) * 4 + e) * 64 + lane;
}

// H = alpha H + beta v for the 16-row blocks 2*t2 and 2*t2+1 of this wave (C/D map of the 16x16 MFMA:
// col = lane & 15, row = 4 * (lane >> 4) + reg); loads batched before the stores
__device__ __forceinline__ void big16_epilogue_rows(float* __restrict__ H, int ldh, int ti, int tj, int wm, int wn,
                                                    int lane, int t2, const float (&v)[2][4][4], float alpha, float beta) {
  const bool diag = ti == tj;
  const int row0 = ti * BT + wm * 128 + 32 * t2 + 4 * (lane >> 4);
  const int col0 = tj * BT + wn * 64 + (lane & 15);
  float old[2][4][4];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = row0 + 16 * tt + e, col = col0 + 16 * u;
        old[tt][u][e] = (!diag || row <= col) ? H[(long)row * ldh + col] : 0.f;
      }
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = row0 + 16 * tt + e, col = col0 + 16 * u;
        const float out = alpha * old[tt][u][e] + beta * v[tt][u][e];
        if (!diag || row <= col) H[(long)row * ldh + col] = out;
      }
}

// fragment reads: A block T (two reads) into FA[2T], FA[2T+1]; B block U into FB[2U], FB[2U+1]
#define B16_RA(FA, T, TBLK) do { const unsigned ad_ = abase ^ (32u * (TBLK)); TR_READ(FA[2 * (T)], ad_, 0); TR_READ(FA[2 * (T) + 1], ad_, 1024); } while (0)
#define B16_RB(FB, U) do { const unsigned ad_ = bbase ^ (32u * (bblk0 + (U))); TR_READ(FB[2 * (U)], ad_, 0); TR_READ(FB[2 * (U) + 1], ad_, 1024); } while (0)
template <bool BF16>
__global__ __launch_bounds__(512) void hessian16_big16_kernel(ProbGroup pg, BigPlan plan, int nx, int tokens) {
  constexpr int BRING = BRING_DEFAULT;
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int spk = tokens / BBK;
  const int nk_all = spk * nx;                                // even (host-checked), and so is every segment

  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int m = (g & 1) | (q << 1);
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)ring);
  const unsigned lane_base = (8 * g + q) * 256 + 32 * m + 16 * (p >> 1) + 8 * (p & 1);
  const unsigned oa = lds0 + wm * BHALF + lane_base;                 // A half image wm, block t at oa ^ (32 t)
  const unsigned ob = lds0 + (2 + (wn >> 1)) * BHALF + lane_base;    // B half image, blocks 4 (wn & 1) + u
  const unsigned bblk0 = 4 * (wn & 1);
  // LDS-DMA: wave w fills image (w >> 1), token rows 16 (w & 1) + 4 u + (lane >> 4), u = 0..3
  const int img = wave >> 1;
  const int rl = lane >> 4;
  const int chunk_lo = (lane & 15) ^ (rl << 2);               // pieces 0, 1 (token-row bit 3 clear)
  const int chunk_hi = chunk_lo ^ 2;                          // pieces 2, 3
  const int dma_dst = img * BHALF + (16 * (wave & 1)) * 256;

  // work items: [0, dp_tiles) whole tiles, then `workers` K-split runs; a launch sized for a CU budget has fewer
  // workgroups than items and every workgroup strides over them
  for (int bid = plan.item0 + blockIdx.x; bid < plan.item1; bid += gridDim.x) {
  const bool whole = bid < plan.dp_tiles;
  for (int seg = 0; seg < BIG_MAXSEG; ++seg) {
    int tile, s0, s1, slot = 0;
    if (whole) {
      if (seg == 1) break;
      tile = bid; s0 = 0; s1 = nk_all;
    } else {
      int l;
      if (!big_segment(plan, bid - plan.dp_tiles, seg, nk_all, l, s0, s1, slot)) break;
      tile = plan.dp_tiles + l;
    }
    const int prob = prob_of_tile(pg, tile);
    const XList& xl = pg.x[prob];
    const int ldx = pg.ldx[prob], ldh = pg.ldh[prob];
    int ti, tj;
    hessian_tile_of(tile - pg.tile_start[prob], pg.C[prob] / BT, ti, tj);
    const int nk = s1 - s0;

    const long gcol = (img < 2 ? (long)ti * BT : (long)tj * BT) + (img & 1) * 128;
    const long lane_goff = (long)(16 * (wave & 1) + rl) * ldx + gcol + 8 * chunk_lo;
    const long hi_delta = 8 * (chunk_hi - chunk_lo);
    int is_slab = s0 / spk, is_in = s0 - is_slab * spk, is_st = 0;
    const unsigned short* src = xl.p[is_slab] + lane_goff + (long)is_in * BBK * ldx;
    auto piece = [&](int u) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)(4 * u) * ldx + (u >= 2 ? hi_delta : 0)),
                                       (__attribute__((address_space(3))) void*)(ring + (is_st % BRING) * BSTAGE + dma_dst + u * 1024),
                                       16, 0, 0);
    };
    auto advance = [&]() {
      ++is_st;
      if (is_st < nk) {
        if (++is_in == spk) {
          is_in = 0;
          ++is_slab;
          src = xl.p[is_slab] + lane_goff;
        } else {
          src += (long)BBK * ldx;
        }
      }
    };

    f32x4v acc[8][4];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][u][e] = 0.f;

#pragma unroll
    for (int st = 0; st < BRING - 1; ++st) {
#pragma unroll
      for (int u = 0; u < 4; ++u) piece(u);
      advance();
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int u = 0; u < 4; ++u) piece(u);
    advance();

    frag_t fl[8], fh[8], fbA[8], fbB[8];                        // A blocks 0-3 / 4-7, B (two sets, alternating by stage)
    {
      const unsigned abase = oa, bbase = ob;
      B16_RA(fl, 0, 0); B16_RA(fl, 1, 1); B16_RA(fl, 2, 2); B16_RA(fl, 3, 3);
      B16_RB(fbA, 0); B16_RB(fbA, 1); B16_RB(fbA, 2); B16_RB(fbA, 3);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BSB;
#define B16_STAGE(KT, FB_CUR, FB_NEXT)                                                                      \
    {                                                                                                       \
      {                                                                                                     \
        const unsigned abase = oa + ((KT) % BRING) * BSTAGE;                                                \
        /* P1: gap k < 4 reads A block 4 + k of this stage */                                               \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                                  \
          const s16x8 fa_ = join8(fl[2 * t_], fl[2 * t_ + 1]);                                              \
          _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                                \
            const s16x8 fb_ = join8(FB_CUR[2 * u_], FB_CUR[2 * u_ + 1]);                                    \
            acc[t_][u_] = mfma16s<BF16>(fa_, fb_, acc[t_][u_]);                                             \
            BSB;                                                                                            \
            if (t_ == 0) B16_RA(fh, u_, 4 + u_);                                                            \
            BSB;                                                                                            \
          }                                                                                                 \
        }                                                                                                   \
      }                                                                                                     \
      BSB;                                                                                                  \
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                           \
      __builtin_amdgcn_s_barrier();                                                                         \
      BSB;                                                                                                  \
      {                                                                                                     \
        const unsigned abase = oa + (((KT) + 1) % BRING) * BSTAGE, bbase = ob + (((KT) + 1) % BRING) * BSTAGE; \
        /* P2: gaps 0-3 read the next stage's A blocks 0-3, gaps 4-7 its B blocks, gaps 8-11 carry the DMA */ \
        _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                                  \
          const s16x8 fa_ = join8(fh[2 * t_], fh[2 * t_ + 1]);                                              \
          _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                                \
            const s16x8 fb_ = join8(FB_CUR[2 * u_], FB_CUR[2 * u_ + 1]);                                    \
            acc[4 + t_][u_] = mfma16s<BF16>(fa_, fb_, acc[4 + t_][u_]);                                     \
            BSB;                                                                                            \
            if (t_ == 0) B16_RA(fl, u_, u_);                                                                \
            if (t_ == 1) B16_RB(FB_NEXT, u_);                                                               \
            if (t_ == 2) piece(u_);                                                                         \
            BSB;                                                                                            \
          }                                                                                                 \
        }                                                                                                   \
      }                                                                                                     \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
      BSB;                                                                                                  \
      advance();                                                                                            \
    }
    for (int kt = 0; kt < nk; kt += 2) {
      B16_STAGE(kt, fbA, fbB);
      B16_STAGE(kt + 1, fbB, fbA);
    }
#undef B16_STAGE
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    BSB;
    __builtin_amdgcn_s_barrier();

    if (whole) {
      float* __restrict__ H = pg.H[prob];
      const float alpha = pg.alpha[prob], beta = pg.beta[prob];
#pragma unroll
      for (int t2 = 0; t2 < 4; ++t2) {
        float v[2][4][4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[tt][u][e] = acc[2 * t2 + tt][u][e];
        big16_epilogue_rows(H, ldh, ti, tj, wm, wn, lane, t2, v, alpha, beta);
      }
    } else {
      float* __restrict__ part = plan.ws + (long)slot * BTILE_FLOATS;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) part[big16_part_index(wave, t, u, e, lane)] = acc[t][u][e];
    }
  }
  }
}

__global__ __launch_bounds__(512) void hessian16_big16_fixup(ProbGroup pg, BigPlan plan, int nk_all) {
  const int l = blockIdx.x;
  const int tile = plan.dp_tiles + l;
  const int prob = prob_of_tile(pg, tile);
  const int ldh = pg.ldh[prob];
  int ti, tj;
  hessian_tile_of(tile - pg.tile_start[prob], pg.C[prob] / BT, ti, tj);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  float* __restrict__ H = pg.H[prob];
  const float alpha = pg.alpha[prob], beta = pg.beta[prob];
  const int t2 = blockIdx.y;
  float v[2][4][4];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[tt][u][e] = 0.f;
  big_for_each_partial(plan, l, nk_all, [&](int slot) {
    const float* __restrict__ part = plan.ws + (long)slot * BTILE_FLOATS;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[tt][u][e] += part[big16_part_index(wave, 2 * t2 + tt, u, e, lane)];
  });
  big16_epilogue_rows(H, ldh, ti, tj, wm, wn, lane, t2, v, alpha, beta);
}

// A[r][c] = A[c][r] for r > c, through a 32x33 LDS tile so both sides stay coalesced.
__global__ __launch_bounds__(256) void symmetrize_kernel(float* __restrict__ A, int lda, int n) {
  __shared__ float t[32][33];
  const int bi = blockIdx.y, bj = blockIdx.x;   // source tile rows bi*32, cols bj*32 (upper: bi <= bj)
  if (bi > bj) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = bi * 32 + ty + 8 * k, c = bj * 32 + tx;
    t[ty + 8 * k][tx] = (r < n && c < n) ? A[(long)r * lda + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = bj * 32 + ty + 8 * k, c = bi * 32 + tx;   // destination (lower) element
    if (r < n && c < n && r > c) A[(long)r * lda + c] = t[tx][ty + 8 * k];
  }
}

}  // namespace gptq

using namespace gptq;

// One H update per problem; every problem brings `n_x` equally shaped slabs [tokens, C] (row-major, ldx).
struct HostProb { float* H; const void* const* xs; float alpha, beta; int C, ldx, ldh; };

// GPTQ_HESS_BIG: 0 = 128 x 128 kernel only, 1 = default, 2 = 256 x 256 kernel whenever the shapes allow,
//                3 = like 2 but without the K-split last round
static int hess_big_env() {
  static const int v = [] { const char* e = getenv("GPTQ_HESS_BIG"); return e ? atoi(e) : 1; }();
  return v;
}

static bool big_eligible(const HostProb& pr, int n_x, int x_dtype, int tokens) {
  if (x_dtype != GPTQ_F16 && x_dtype != GPTQ_BF16) return false;
  if (pr.C % BT != 0 || pr.ldx % 8 != 0 || tokens % (2 * BBK) != 0) return false;   // (stages in pairs: 64 tokens)
  for (int i = 0; i < n_x; ++i)
    if (reinterpret_cast<uintptr_t>(pr.xs[i]) % 16 != 0) return false;
  return true;
}

// 256 x 256 kernels: up to MAX_PROB problems per launch, of possibly different widths (all big_eligible).
// cu_limit > 0: size the launch for at most that many compute units (the workgroups stride over the items).
static int hessian_launch_big(const HostProb* probs, int n_prob, int n_x, int x_dtype, int tokens, int cu_limit,
                              hipStream_t s) {
  const int big_env = hess_big_env();
  int dev = 0, n_cu = 256;
  GPTQ_CHECK_HIP(hipGetDevice(&dev));
  GPTQ_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  if (cu_limit > 0) n_cu = std::max(8, std::min(n_cu, cu_limit));   // leave the other CUs to concurrent streams
  for (int p0 = 0; p0 < n_prob; p0 += MAX_PROB) {
    const int np = std::min(MAX_PROB, n_prob - p0);
    for (int i0 = 0; i0 < n_x; i0 += MAX_XLIST) {
      const int nx = std::min(MAX_XLIST, n_x - i0);
      ProbGroup pg{};
      pg.n_prob = np;
      int total = 0;
      for (int p = 0; p < np; ++p) {
        const HostProb& pr = probs[p0 + p];
        pg.H[p] = pr.H;
        pg.alpha[p] = i0 == 0 ? pr.alpha : 1.f;
        pg.beta[p] = pr.beta;
        pg.C[p] = pr.C; pg.ldx[p] = pr.ldx; pg.ldh[p] = pr.ldh;
        pg.tile_start[p] = total;
        total += (pr.C / BT) * (pr.C / BT + 1) / 2;
        for (int i = 0; i < nx; ++i) pg.x[p].p[i] = static_cast<const unsigned short*>(pr.xs[i0 + i]);
      }
      for (int p = np; p <= MAX_PROB; ++p) pg.tile_start[p] = total;
      const int nk_all = tokens / BBK * nx;
#ifdef GPTQ_DIAG
      static const int shape_env = [] { const char* e = getenv("GPTQ_HESS_SHAPE"); return e ? atoi(e) : 16; }();
      const bool shape16 = shape_env == 16;
#else
      constexpr bool shape16 = true;
#endif
      BigPlan plan{total, 0, 0, 1, nullptr, 0, 0, 0};   // dp_tiles, left_tiles, workers, chunk, ws, item0, item1, head
      const int full = total / n_cu * n_cu, left = total - full;
      // cut the last round along K when it would run under 90 % full and a run still has >= 8 stages
      if (big_env != 3 && left > 0 && left * 10 < n_cu * 9 && (long)left * nk_all >= 8L * n_cu) {
        plan.dp_tiles = full;
        plan.left_tiles = left;
        plan.chunk = cdiv((long)left * nk_all, n_cu);
        if (shape16) plan.chunk += plan.chunk & 1;                  // the 16x16x32 kernel walks stages in pairs
        plan.workers = cdiv((long)left * nk_all, plan.chunk);
        size_t slots = 2 * (size_t)plan.workers;
        // heads + tails (16x16x32 kernel; see BigPlan): while a tail run touches at most BIG_MAXSEG tiles
        static const int heads_env = tune_knob("GPTQ_HESS_HEADS", 1);
        const int tail_len = nk_all - plan.chunk;
        // (measured: the split round of C = 11008, 178 tiles, 0.83 -> 0.60 ms and 5.6 -> 3.0 GB; but 4-7 % SLOWER at
        //  136 tiles (C = 4096) and 16 tiles (C = 8192), where equal runs happen to line up and more segments cost more
        //  than the traffic they save: taken from 60 % of a round on)
        if (heads_env && shape16 && tail_len > 0 && cdiv(plan.chunk, tail_len) + 1 <= BIG_MAXSEG &&
            (heads_env == 2 || left * 10 >= n_cu * 6)) {
          plan.head = plan.chunk;
          const int ntail = cdiv((long)left * tail_len, plan.chunk);
          plan.workers = left + ntail;
          slots = (size_t)left + (size_t)BIG_MAXSEG * ntail;
        }
        plan.ws = static_cast<float*>(scratch_buffer(s, sizeof(float) * BTILE_FLOATS * slots));
        GPTQ_CHECK_ARG(plan.ws != nullptr, "gptq_hessian_accum: cannot allocate the %zu-byte split-K workspace",
                       sizeof(float) * BTILE_FLOATS * slots);
      }
      // One launch per ROUND of n_cu work items instead of one launch for all of them: the workgroups of a launch start
      // together, so the tiles an XCD runs side by side walk K in step and share their operand panels in its L2.  In a
      // single launch the later rounds start one by one as compute units free up; a tile that starts 5 % of a tile
      // time after its neighbours is 50 stages behind them, far outside what 4 MB of L2 keeps.
      // (A launch sized for a CU budget keeps the single strided launch.)
      static const int rounds_env = tune_knob("GPTQ_HESS_ROUNDS", 1);
      const int items = plan.dp_tiles + plan.workers;
      const bool by_rounds = rounds_env != 0 && cu_limit <= 0;
      const int per_launch = by_rounds ? n_cu : items;
#ifdef GPTQ_DIAG   // timing-only ablation builds exist in the diagnostic library alone (python -m gptq_amd.build --diag)
      static const int bring_env = tune_knob("GPTQ_HESS_RING", BRING_DEFAULT);
      static const int babl_env = [] { const char* e = getenv("GPTQ_HESS_ABLATE"); return e ? atoi(e) : 0; }();
#endif
#ifdef GPTQ_DIAG
#define HBIG(BF, RG, AB)                                                                                      \
  do {                                                                                                        \
    const size_t lds_b = (size_t)(RG) * BSTAGE;                                                               \
    GPTQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hessian16_big_kernel<BF, RG, AB>),      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));              \
    hessian16_big_kernel<BF, RG, AB><<<grid, 512, lds_b, s>>>(pg, plan, nx, tokens);                          \
  } while (0)
#endif
      for (int it0 = 0; it0 < items; it0 += per_launch) {
        plan.item0 = it0;
        plan.item1 = std::min(items, it0 + per_launch);
        int grid = plan.item1 - plan.item0;
        if (cu_limit > 0) grid = std::min(grid, n_cu);            // a real budget: the workgroups stride over the items
        if (shape16) {
          const size_t lds_b = (size_t)BRING_DEFAULT * BSTAGE;
          if (x_dtype == GPTQ_BF16) {
            GPTQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hessian16_big16_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
            hessian16_big16_kernel<true><<<grid, 512, lds_b, s>>>(pg, plan, nx, tokens);
          } else {
            GPTQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hessian16_big16_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
            hessian16_big16_kernel<false><<<grid, 512, lds_b, s>>>(pg, plan, nx, tokens);
          }
          continue;
        }
#ifdef GPTQ_DIAG
        if (x_dtype == GPTQ_BF16) HBIG(true, BRING_DEFAULT, 0);
        else if (babl_env == 1) HBIG(false, BRING_DEFAULT, 1);
        else if (babl_env == 2) HBIG(false, BRING_DEFAULT, 2);
        else if (bring_env == 5) HBIG(false, 5, 0);
        else HBIG(false, BRING_DEFAULT, 0);
#endif
      }
#undef HBIG
      if (plan.left_tiles > 0) {
        if (shape16) hessian16_big16_fixup<<<dim3(plan.left_tiles, 4), 512, 0, s>>>(pg, plan, nk_all);
#ifdef GPTQ_DIAG
        else hessian16_big_fixup<<<dim3(plan.left_tiles, 4, 8), 64, 0, s>>>(pg, plan, nk_all);
#endif
      }
    }
  }
  GPTQ_CHECK_LAUNCH("hessian16_big_kernel");
  return GPTQ_OK;
}

// Problems of one shape (C, ldx, ldh as in probs[0]).
static int hessian_launch(const HostProb* probs, int n_prob, int n_x, int x_dtype, int tokens, int cu_limit,
                          hipStream_t s) {
  const int C = probs[0].C, ldx = probs[0].ldx, ldh = probs[0].ldh;
  const int nt = cdiv(C, GBM);
  const int blocks = nt * (nt + 1) / 2;
  if (x_dtype == GPTQ_F16 || x_dtype == GPTQ_BF16) {
    bool aligned = (ldx % 8 == 0) && (tokens % HBK == 0) && (C % GBM == 0);
    for (int p = 0; p < n_prob; ++p)
      for (int i = 0; i < n_x; ++i) aligned = aligned && (reinterpret_cast<uintptr_t>(probs[p].xs[i]) % 16 == 0);
    const int big_env = hess_big_env();
    bool big_ok = big_env != 0;
    for (int p = 0; p < n_prob; ++p) big_ok = big_ok && big_eligible(probs[p], n_x, x_dtype, tokens);
    const long big_tiles = (long)(C / BT) * (C / BT + 1) / 2 * n_prob;
    if (big_ok && (big_env >= 2 || big_tiles >= 100))
      return hessian_launch_big(probs, n_prob, n_x, x_dtype, tokens, cu_limit, s);
    if (aligned) {
      static const int ring_env = tune_knob("GPTQ_HESS_RING", RING_DEFAULT);
      const size_t lds = (size_t)ring_env * DSTAGE;
#ifdef GPTQ_DIAG
      static const int ablate = [] { const char* e = getenv("GPTQ_HESS_ABLATE"); return e ? atoi(e) : 0; }();
#endif
      for (int p0 = 0; p0 < n_prob; p0 += MAX_PROB) {
        const int np = std::min(MAX_PROB, n_prob - p0);
        for (int i0 = 0; i0 < n_x; i0 += MAX_XLIST) {
          const int nx = std::min(MAX_XLIST, n_x - i0);
          ProbGroup pg{};
          for (int p = 0; p < np; ++p) {
            pg.H[p] = probs[p0 + p].H;
            pg.alpha[p] = i0 == 0 ? probs[p0 + p].alpha : 1.f;
            pg.beta[p] = probs[p0 + p].beta;
            for (int i = 0; i < nx; ++i) pg.x[p].p[i] = static_cast<const unsigned short*>(probs[p0 + p].xs[i0 + i]);
          }
#define HDMA(BF, AB, RG)                                                                                       \
  do {                                                                                                         \
    GPTQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hessian16_dma_kernel<BF, AB, RG>),       \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
    hessian16_dma_kernel<BF, AB, RG><<<np * blocks, 512, lds, s>>>(pg, blocks, ldh, nx, ldx, C, tokens);       \
  } while (0)
          if (x_dtype == GPTQ_BF16) HDMA(true, 0, RING_DEFAULT);
          else if (ring_env == 3) HDMA(false, 0, 3);
#ifdef GPTQ_DIAG
          else if (ablate == 1) HDMA(false, 1, RING_DEFAULT);
          else if (ablate == 2) HDMA(false, 2, RING_DEFAULT);
#endif
          else HDMA(false, 0, RING_DEFAULT);
#undef HDMA
        }
      }
      GPTQ_CHECK_LAUNCH("hessian16_dma_kernel");
      return GPTQ_OK;
    }
  }
  for (int p = 0; p < n_prob; ++p) {
    float* H = probs[p].H;
    for (int i = 0; i < n_x; ++i) {
      const float a = i == 0 ? probs[p].alpha : 1.f;
      const float beta = probs[p].beta;
      switch (x_dtype) {
        case GPTQ_F32: {
          const float* x = static_cast<const float*>(probs[p].xs[i]);
          hessian_kernel<float><<<blocks, GEMM_THREADS, 0, s>>>(H, ldh, x, ldx, C, tokens, a, beta, vec_ok(x, ldx));
          break;
        }
        case GPTQ_F16:
        case GPTQ_BF16: {
          const unsigned short* x = static_cast<const unsigned short*>(probs[p].xs[i]);
          const bool vec = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (ldx % 8 == 0);
          const size_t lds = sizeof(unsigned short) * 4 * HTILE;
          if (x_dtype == GPTQ_F16) {
            GPTQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hessian16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hessian16_kernel<false><<<blocks, 256, lds, s>>>(H, ldh, x, ldx, C, tokens, a, beta, vec);
          } else {
            GPTQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&hessian16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hessian16_kernel<true><<<blocks, 256, lds, s>>>(H, ldh, x, ldx, C, tokens, a, beta, vec);
          }
          break;
        }
        default:
          GPTQ_CHECK_ARG(false, "gptq_hessian_accum: unknown dtype %d", x_dtype);
      }
    }
  }
  GPTQ_CHECK_LAUNCH("hessian_kernel");
  return GPTQ_OK;
}

extern "C" int gptq_hessian_accum_group(int n_prob, float* const* H, int ldh, const void* const* X, int n_x,
                                        int x_dtype, int ldx, int C, int tokens_each,
                                        const int* nsamples_before, int batch_total, gptq_stream_t stream) {
  GPTQ_CHECK_ARG(n_prob > 0 && n_prob <= 64 && H && X && nsamples_before && n_x > 0, "gptq_hessian_accum_group: bad arguments");
  GPTQ_CHECK_ARG(C > 0 && tokens_each > 0 && batch_total > 0, "gptq_hessian_accum_group: bad sizes");
  GPTQ_CHECK_ARG(ldh >= C && ldx >= C, "gptq_hessian_accum_group: leading dimension smaller than C");
  HostProb probs[64];
  for (int p = 0; p < n_prob; ++p) {
    GPTQ_CHECK_ARG(H[p] != nullptr && nsamples_before[p] >= 0, "gptq_hessian_accum_group: bad problem %d", p);
    for (int i = 0; i < n_x; ++i) GPTQ_CHECK_ARG(X[(long)p * n_x + i] != nullptr, "gptq_hessian_accum_group: null slab");
    const int n_after = nsamples_before[p] + batch_total;
    probs[p] = HostProb{H[p], X + (long)p * n_x,
                        (float)((double)nsamples_before[p] / (double)n_after),   // gptq.py:59
                        (float)(2.0 / (double)n_after),                          // gptq.py:62 squared
                        C, ldx, ldh};
  }
  return hessian_launch(probs, n_prob, n_x, x_dtype, tokens_each, 0, static_cast<hipStream_t>(stream));
}

// Problems of different widths in one call (the Linears of a block hooked in the same forward passes): those the
// 256 x 256 kernel takes (C % 256 == 0, 16-bit activations, aligned) share its launches -- their tiles fill the
// chip together and share the K-split last round -- the others go out per shape.
extern "C" int gptq_hessian_accum_mixed(int n_prob, float* const* H, const int* ldh, const void* const* X, int n_x,
                                        int x_dtype, const int* ldx, const int* C, int tokens_each,
                                        const int* nsamples_before, int batch_total, int n_cu,
                                        gptq_stream_t stream) {
  GPTQ_CHECK_ARG(n_prob > 0 && n_prob <= 64 && H && ldh && X && ldx && C && nsamples_before && n_x > 0,
                 "gptq_hessian_accum_mixed: bad arguments");
  GPTQ_CHECK_ARG(tokens_each > 0 && batch_total > 0 && n_cu >= 0, "gptq_hessian_accum_mixed: bad sizes");
  hipStream_t s = static_cast<hipStream_t>(stream);
  HostProb big[64], rest[64];
  int n_big = 0, n_rest = 0;
  long big_tiles = 0;
  for (int p = 0; p < n_prob; ++p) {
    GPTQ_CHECK_ARG(H[p] != nullptr && nsamples_before[p] >= 0 && C[p] > 0 && ldh[p] >= C[p] && ldx[p] >= C[p],
                   "gptq_hessian_accum_mixed: bad problem %d", p);
    for (int i = 0; i < n_x; ++i) GPTQ_CHECK_ARG(X[(long)p * n_x + i] != nullptr, "gptq_hessian_accum_mixed: null slab");
    const int n_after = nsamples_before[p] + batch_total;
    const HostProb pr{H[p], X + (long)p * n_x, (float)((double)nsamples_before[p] / (double)n_after),
                      (float)(2.0 / (double)n_after), C[p], ldx[p], ldh[p]};
    if (hess_big_env() != 0 && big_eligible(pr, n_x, x_dtype, tokens_each)) {
      big[n_big++] = pr;
      big_tiles += (long)(C[p] / BT) * (C[p] / BT + 1) / 2;
    } else {
      rest[n_rest++] = pr;
    }
  }
  if (n_big > 0 && !(hess_big_env() >= 2 || big_tiles >= 100)) {      // too few tiles for the big kernel
    for (int p = 0; p < n_big; ++p) rest[n_rest++] = big[p];
    n_big = 0;
  }
  if (n_big > 0)
    if (int rc = hessian_launch_big(big, n_big, n_x, x_dtype, tokens_each, n_cu, s)) return rc;
  // the others: one call per shape
  bool done[64] = {};
  for (int p = 0; p < n_rest; ++p) {
    if (done[p]) continue;
    HostProb same[64];
    int n = 0;
    for (int q = p; q < n_rest; ++q)
      if (!done[q] && rest[q].C == rest[p].C && rest[q].ldx == rest[p].ldx && rest[q].ldh == rest[p].ldh) {
        same[n++] = rest[q];
        done[q] = true;
      }
    if (int rc = hessian_launch(same, n, n_x, x_dtype, tokens_each, n_cu, s)) return rc;
  }
  return GPTQ_OK;
}

extern "C" int gptq_hessian_accum_multi(float* H, int ldh, const void* const* X, int n_x, int x_dtype, int ldx,
                                        int C, int tokens_each, int nsamples_before, int batch_total,
                                        gptq_stream_t stream) {
  float* Hs[1] = {H};
  const int nb[1] = {nsamples_before};
  return gptq_hessian_accum_group(1, Hs, ldh, X, n_x, x_dtype, ldx, C, tokens_each, nb, batch_total, stream);
}

extern "C" int gptq_hessian_accum(float* H, int ldh, const void* X, int x_dtype, int ldx, int C,
                                  int tokens, int nsamples_before, int batch, gptq_stream_t stream) {
  GPTQ_CHECK_ARG(X != nullptr, "gptq_hessian_accum: null pointer");
  const void* xs[1] = {X};
  return gptq_hessian_accum_multi(H, ldh, xs, 1, x_dtype, ldx, C, tokens, nsamples_before, batch, stream);
}

extern "C" int gptq_symmetrize(float* A, int lda, int n, gptq_stream_t stream) {
  GPTQ_CHECK_ARG(A && n > 0 && lda >= n, "gptq_symmetrize: bad arguments");
  const int nt = cdiv(n, 32);
  symmetrize_kernel<<<dim3(nt, nt), 256, 0, static_cast<hipStream_t>(stream)>>>(A, lda, n);
  GPTQ_CHECK_LAUNCH("symmetrize_kernel");
  return GPTQ_OK;
}
