// Column loop of one SUPER-BLOCK (up to 4 lazy-batch blocks of 128 columns, gptq.py:191-276) in ONE launch.
//
// Why: the per-block kernel (fasterquant.hip, quant_block_kernel) gives a row to a quad of lanes, so a 4096-row Linear
// occupies 256 of the chip's 1024 SIMDs with one wave each, and every block pays a launch, a 64 x 64-tile rank-128
// update launch (latency-bound: ~25 us for ~0.1 us of MFMA work per CU) and two launch boundaries: ~95 us per 128
// columns in situ (profiles/r02_solve_trace_down_proj.txt), of which the dependent quantize chain itself is ~15 us.
// Rows of W are independent problems given U (gptq.py:262-276 has no cross-row term), so here
//   * a row is spread over L = 16 or 8 lanes (lane c holds block columns L t + c): 4x / 2x the waves for the same
//     rows, a quarter / half of the in-block rank-1 update work per lane; every lane of a row runs the same
//     (bitwise identical) quantize chain for a super-step of L columns, whose current values are exchanged with
//     ds_bpermute (one instruction per column, off the dependent chain);
//   * a workgroup (256 threads = 16 or 32 rows) walks ALL blocks of the super-block: after a block it applies that
//     block's rank-128 update to the rest of the super-block's columns of ITS rows itself, on the matrix cores
//     (v_mfma_f32_16x16x4_f32: the rows are the M dimension, the B operand comes straight from L2 -- every wave owns
//     its own output columns, nothing is shared through LDS), so there is no launch, no grid-wide dependency and no
//     second pass over W per block.  Only the rank-512 updates beyond the super-block remain separate launches.
// Arithmetic: the in-block loop is the reference's IEEE fp32 sequence, operation for operation (true division,
// round-half-even, separate multiply and subtract: -ffp-contract=off), exactly as in quant_block_kernel; the rank-128
// update sums k = 0 .. 127 in ascending order from zero and subtracts the sum from W, like gemm_tile64 (an fp32 MFMA
// is an fmaf chain, whatever its shape) -- results are bit-identical to the per-block path, whatever L.
// Factor form only (gptq_fasterquant_rows: C % 128 == 0, blocksize 128): Err receives Q1 - W0.
#include <stdlib.h>

#include "common.h"

namespace gptq {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct QuantSuperArgs {
  float* W; int ldw; int R; int s0; int nb;          // columns [s0, s0 + 128 nb) of W [R, ldw]
  const float* U; int ldu;                           // what gptq_rfactor_upper left in H: U_kk inside the diagonal blocks, Rt above
  const float* scale_tab; const float* zero_tab; int tab_ld; const int32_t* col_group;   // col_group: nullable
  float maxq;
  float* Err; int lde;                               // [R, lde]: Q1 - W0 of block b at columns [128 b, 128 b + 128)
  uint8_t* codes; int ldc; const int32_t* col_map;   // codes nullable; col_map nullable (act-order: original column)
  float* loss;
  const float* w0; int ldw0;                         // the original (permuted, dead columns zeroed) weights
  int codes_wide;                                    // codes rows allow 8-byte stores and there is no column map
};

#ifdef GPTQ_DIAG   // diagnostic library only: s_memtime stamps of workgroup 0, block 0 of the LAST launch (tools/qs_phases.py)
__device__ unsigned long long qs_stamps[32];
#define QS_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && b == 0) qs_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QS_STAMP(i) do { } while (0)
#endif

template <int L>
struct QS {
  static constexpr int ROWS = 256 / L;               // rows per workgroup
  static constexpr int NT = 128 / L;                 // block columns per lane
  static constexpr int SPP = 32 / L;                 // super-steps (of L columns) per 32-column phase
  static constexpr int MT = ROWS / 16;               // 16-row MFMA tiles
  static constexpr int NCH = NT / 4;                 // float4 chunks of a lane's slice of a U row
  static constexpr int TLD = 132;                    // row stride of the transpose / A-operand tile
  static constexpr int UBUF = 32 * 128;              // floats per U-row buffer: 32 staged rows of the 128 x 128 diagonal block
  static constexpr int UD = 32 * L;                  // floats per dense diagonal sub-block buffer (SPP blocks of L x L)
  static constexpr int CLD = 144;                    // byte stride of the code tile
  // U-row buffers: two (the next phase's rows are stored while this phase still reads its own) for 16 lanes; ONE for 8
  // lanes (a second barrier per phase instead of 17 KB: 40 KB per workgroup = three per compute unit also by LDS)
  static constexpr int NBUF = L == 16 ? 2 : 1;
  static constexpr size_t LDS_BYTES = sizeof(float) * (NBUF * UBUF + NBUF * UD + ROWS * TLD) + ROWS * CLD + 2 * 128 * sizeof(int);
};

template <int L>
__device__ __forceinline__ float lane_bcast(float v, int base4, int k) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(base4 + 4 * k, __builtin_bit_cast(int, v)));
}

// quant.py:9 -- clamp(round(x / s) + z, 0, maxq)
__device__ __forceinline__ float qs_affine_code(float x, float s, float z, float maxq) {
  return fminf(fmaxf(rintf(x / s) + z, 0.f), maxq);
}

// U rows [32 ph, 32 ph + 32) of the diagonal block at (i1, i1): thread `tid` fetches 4 float4 pieces (coalesced rows).
__device__ __forceinline__ void qs_stage_fetch(const float* __restrict__ U, int ldu, int i1, int ph, int tid, f32x4 (&v)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + 256 * j, il = idx >> 5, k4 = (idx & 31) * 4;
    v[j] = *reinterpret_cast<const f32x4*>(U + (long)(i1 + 32 * ph + il) * ldu + i1 + k4);
  }
}
// ... and scatters them into the lane-major image Us[il][c][t] (element k = L t + c of row il; the float4 chunks of a
// lane's slice are XOR-swizzled by the lane's upper bits so that the 16 lanes of a ds_read_b128 group hit 64 distinct
// banks without padding) plus the dense copy Ud[ss][cc][c2] of the L x L diagonal sub-blocks (the chain reads those
// with wave-uniform addresses).
template <int L>
__device__ __forceinline__ void qs_stage_store(float* Us, float* Ud, int ph, int tid, const f32x4 (&v)[4]) {
  constexpr int NT = QS<L>::NT;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + 256 * j, il = idx >> 5, k4 = (idx & 31) * 4;
    const int i = 32 * ph + il;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k4 + e, c = k % L, t = k / L;
      const float x = (k >= i) ? v[j][e] : 0.f;                          // upper triangular block
      const int sw = (c * NT) >> 6;                                      // chunk swizzle of lane c
      Us[il * 128 + c * NT + ((((t >> 2) ^ sw) << 2) | (t & 3))] = x;
      if (k / L == i / L) Ud[(il / L) * L * L + (il % L) * L + c] = x;   // same super-step: diagonal sub-block
    }
  }
}

// One super-step: L consecutive columns 32 PH + L SS ... of the block.
template <int L, bool GROUPED, int PH, int SS>
__device__ __forceinline__ void qs_super_step(float (&w)[QS<L>::NT], float (&cd)[QS<L>::NT],
                                              const float (&psc)[QS<L>::NT], const float (&pzr)[QS<L>::NT], float sc,
                                              float zr, float maxq, const float* Usb, const float* Udb, int c, int base4,
                                              int swz, float& loss) {
  constexpr int NT = QS<L>::NT, SPP = QS<L>::SPP, NCH = QS<L>::NCH;
  constexpr int t = PH * SPP + SS;
  float cur[L], gs[L], gz[L], errs[L];
#pragma unroll
  for (int cc = 0; cc < L; ++cc) cur[cc] = lane_bcast<L>(w[t], base4, cc);
  if (GROUPED) {
#pragma unroll
    for (int cc = 0; cc < L; ++cc) {
      gs[cc] = lane_bcast<L>(psc[t], base4, cc);
      gz[cc] = lane_bcast<L>(pzr[t], base4, cc);
    }
  }
  // (a) the dependent chain of the L columns (identical in every lane of the row)
#pragma unroll
  for (int cc = 0; cc < L; ++cc) {
    // (without this the scheduler hoists the U reads of all L columns to the top of the super-step: L * L live registers)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const float* ud = Udb + SS * L * L + cc * L;               // U[i][L t + c2], c2 = 0 .. L - 1 (wave-uniform address)
    float urow[L];
#pragma unroll
    for (int k = cc / 4; k < L / 4; ++k) {
      const f32x4 u4 = *reinterpret_cast<const f32x4*>(ud + 4 * k);
      urow[4 * k] = u4[0]; urow[4 * k + 1] = u4[1]; urow[4 * k + 2] = u4[2]; urow[4 * k + 3] = u4[3];
    }
    float gsc = sc, gzr = zr;
    if (GROUPED) { gsc = gs[cc]; gzr = gz[cc]; }
    const float x = cur[cc];
    const float code = qs_affine_code(x, gsc, gzr, maxq);      // gptq.py:262-264
    const float q = gsc * (code - gzr);
    const float d = urow[cc];                                  // Hinv1[i, i]
    const float err = (x - q) / d;                             // gptq.py:269
    errs[cc] = err;
    loss += err * err;                                         // (w - q)^2 / d^2, gptq.py:267 (tolerance-level)
    if (c == cc) { w[t] = q; cd[t] = code; }
    // (pins the two selects HERE: left to itself the DAG scheduler sinks all L of them to the end of the super-step and
    //  spills every column's code and q on the way)
    asm volatile("" : "+v"(w[t]), "+v"(cd[t]));
#pragma unroll
    for (int c2 = cc + 1; c2 < L; ++c2) cur[c2] -= err * urow[c2];   // gptq.py:270
  }
  // (b) the L rank-1 updates of this lane's later columns, per element in the reference's order
  if (t + 1 < NT) {
#pragma unroll
    for (int cc = 0; cc < L; ++cc) {
      const int il = SS * L + cc;
      const float err = errs[cc];
      if (cc % 4 == 0) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int k = (t + 1) / 4; k < NCH; ++k) {
        const f32x4 u4 = *reinterpret_cast<const f32x4*>(Usb + il * 128 + c * NT + ((k ^ swz) << 2));
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * k + e > t) w[4 * k + e] -= err * u4[e];
      }
      // (pins the updates next to their U reads: the reads are ordered by the memory clobber above, the arithmetic is not,
      //  and sunk behind ALL the reads of the super-step it turns every read into a spill)
#pragma unroll
      for (int j = t + 1; j < NT; ++j) asm volatile("" : "+v"(w[j]));
    }
  }
}

template <int L, bool GROUPED, int PH>
__device__ __forceinline__ void qs_phase(const QuantSuperArgs& a, int i1_next, bool have_next, float (&w)[QS<L>::NT],
                                         const float (&psc)[QS<L>::NT], const float (&pzr)[QS<L>::NT], float sc, float zr,
                                         float* Us, float* Ud, float* T0, uint8_t* T1b, const int* cmap, int i1, int c,
                                         int row_l, long rbase, bool active, int base4, int swz, int tid, float& loss) {
  constexpr int UBUF = QS<L>::UBUF, UD = QS<L>::UD, SPP = QS<L>::SPP, NBUF = QS<L>::NBUF, NT = QS<L>::NT;
  constexpr int TLD = QS<L>::TLD, CLD = QS<L>::CLD;
  // Launder the per-lane indices once per phase: otherwise every LDS / bpermute address of all four phases (hundreds of
  // distinct base + constant values) is hoisted out of the block loop as loop-invariant and spilled (measured: 1000
  // spilled registers); laundered, an address is formed next to its use and folded into the instruction's offset field.
  asm volatile("" : "+v"(c), "+v"(base4), "+v"(swz), "+v"(tid), "+v"(row_l));
  float* Usb = Us + (NBUF == 2 ? (PH & 1) * UBUF : 0);
  float* Udb = Ud + (NBUF == 2 ? (PH & 1) * UD : 0);
  __syncthreads();                                             // this phase's rows are staged (two buffers: the other one is dead)
  // the rows of the NEXT phase (of the next block after phase 3) travel under the chain
  f32x4 nxt[4];
  const bool fetch = PH < 3 || have_next;                      // (workgroup-uniform)
  if (fetch) qs_stage_fetch(a.U, a.ldu, PH < 3 ? i1 : i1_next, (PH + 1) & 3, tid, nxt);
  float cd[NT];                                                // (only this phase's SPP entries are used)
#define QS_STEP(S) if constexpr (SPP > S) qs_super_step<L, GROUPED, PH, S>(w, cd, psc, pzr, sc, zr, a.maxq, Usb, Udb, c, base4, swz, loss)
  QS_STEP(0); QS_STEP(1); QS_STEP(2); QS_STEP(3);
#undef QS_STEP
  // This phase's columns are final: into the transpose tiles NOW (the block's retire reads them back row-contiguous).
  // Kept in registers until the block's end they were spilled to scratch by the later phases and re-loaded one by one.
#pragma unroll
  for (int ss = 0; ss < SPP; ++ss) {
    const int t = PH * SPP + ss;
    T0[row_l * TLD + L * t + c] = w[t];
    T1b[row_l * CLD + L * t + c] = (uint8_t)cd[t];
    if (a.codes && !a.codes_wide && active) a.codes[rbase * a.ldc + cmap[L * t + c]] = (uint8_t)cd[t];
  }
  if (fetch) {
    if (NBUF == 1) __syncthreads();                            // every wave is done reading the (single) buffer
    qs_stage_store<L>(Us + (NBUF == 2 ? ((PH + 1) & 1) * UBUF : 0), Ud + (NBUF == 2 ? ((PH + 1) & 1) * UD : 0), (PH + 1) & 3,
                      tid, nxt);
  }
}

// W[rows of this workgroup, cols] -= X[rows, 0:128] * Rt[i1 : i1 + 128, cols]  for the columns [c_lo, c_lo + 128 NREM):
// wave wv owns the 32 NREM columns from c_lo + 32 NREM wv, processed as segments of TS = 4 or 2 MFMA tiles of 16 columns
// (NREM = 1: one 2-tile segment, 2: one 4-tile segment, 3: a 4-tile and a 2-tile segment).  The columns of a segment's
// tiles are chosen so that what a lane holds of a B row (and of a W row) is contiguous AND the 16 lanes of a k row read
// one contiguous run: tile j holds the columns TS n + j (one float4 / float2 per lane and row).
// The B operand comes straight from L2 into registers (every wave owns its columns: nothing to share through LDS), a ring
// of 8 (4 for 32-row workgroups) k-groups deep; its first loads are issued before the block is retired (B does not depend
// on the block's results).  Measured on the way: batches of 4 k-groups 32 k cycles per update (6 k of MFMAs), a ring whose
// refills the scheduler was free to sink 26 k, pinned refills 14 k.
template <int TS>
__device__ __forceinline__ void qs_seg_load(const float* p, int n, float (&dst)[TS]) {   // p: row base + the segment's first column
  if constexpr (TS == 2) {
    const float2 v = *reinterpret_cast<const float2*>(p + 2 * n);
    dst[0] = v.x; dst[1] = v.y;
  } else {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * n);
    dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
  }
}
template <int TS>
__device__ __forceinline__ void qs_seg_store(float* p, int n, const float (&v)[TS]) {
  if constexpr (TS == 2) *reinterpret_cast<float2*>(p + 2 * n) = make_float2(v[0], v[1]);
  else *reinterpret_cast<f32x4*>(p + 4 * n) = f32x4{v[0], v[1], v[2], v[3]};
}

template <int L, int TS>
struct QsNear {
  static constexpr int MT = QS<L>::MT, TLD = QS<L>::TLD;
  static constexpr int RING = MT == 1 ? 8 : 4;                  // (32 rows per workgroup: twice the accumulators, half the ring)
  float bq[RING][TS];
  const float* Bp;
  __device__ __forceinline__ void prefetch(const QuantSuperArgs& a, int i1, long col, int tid) {   // col: the segment's first column
    const int lane = tid & 63;
    Bp = a.U + (long)(i1 + (lane >> 4)) * a.ldu + col;
#pragma unroll
    for (int u = 0; u < RING; ++u) qs_seg_load<TS>(Bp + (long)(4 * u) * a.ldu, lane & 15, bq[u]);
  }
  __device__ __forceinline__ void run(const QuantSuperArgs& a, const float* Et, long col, int row0, int tid) {
    const int lane = tid & 63;
    const int n = lane & 15, kq = lane >> 4;
    f32x4 acc[MT][TS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < TS; ++j) acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* Ap = Et + n * TLD + kq;                        // A(m, k) = Et[m][k]
#pragma unroll
    for (int kg = 0; kg < 32; ++kg) {
      float av[MT], bv[TS];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[mt] = Ap[16 * mt * TLD + 4 * kg];
#pragma unroll
      for (int j = 0; j < TS; ++j) bv[j] = bq[kg % RING][j];
      if (kg + RING < 32) qs_seg_load<TS>(Bp + (long)(4 * (kg + RING)) * a.ldu, n, bq[kg % RING]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < TS; ++j)
          acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[j], acc[mt][j], 0, 0, 0);
      // (keeps the ring's refill HERE: left alone the scheduler sinks every refill to just before its use, RING
      //  iterations later, and each k-group then pays a whole L2 round trip)
      __builtin_amdgcn_sched_barrier(0);
    }
    // old values after the product (before it: MT * 4 * TS more live registers)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float old[4][TS];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        qs_seg_load<TS>(a.W + (long)min(row0 + 16 * mt + 4 * kq + i, a.R - 1) * a.ldw + col, n, old[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = row0 + 16 * mt + 4 * kq + i;
        float out[TS];
#pragma unroll
        for (int j = 0; j < TS; ++j) out[j] = old[i][j] - acc[mt][j][i];
        if (r < a.R) qs_seg_store<TS>(a.W + (long)r * a.ldw + col, n, out);
      }
    }
  }
};

// Retire block b (Q1 -> W, Q1 - W0 -> Err and -> the A-operand tile, codes) and apply its rank-128 update to the NREM
// blocks that follow it in the super-block.
template <int L, int NREM>
__device__ __forceinline__ void qs_retire(const QuantSuperArgs& a, float* T0, const uint8_t* T1b, int i1,
                                          int b, int c, int row_l, int row0, long rbase, bool active, int tid) {
  constexpr int NT = QS<L>::NT, TLD = QS<L>::TLD, CLD = QS<L>::CLD;
    // ---- retire the block: Q1 -> W, Q1 - W0 -> Err (global, for the far updates) and -> the A-operand tile ----
    // the wave's columns of the NREM blocks that follow: [wcol, wcol + 32 NREM)
    const long wcol = (long)i1 + 128 + 32 * NREM * (tid >> 6);
    QsNear<L, NREM == 1 ? 2 : 4> near;
    QsNear<L, 2> near2;                                         // (NREM == 3: its last 32 columns)
    if constexpr (NREM > 0) near.prefetch(a, i1, wcol, tid);
    // the original weights of the columns this lane retires (c NT ... c NT + NT - 1 of the block): loaded HERE, under the
    // transposes -- held across the four phases they were spilled to scratch at 168 registers (8 lanes per row)
    f32x4 w0p[NT / 4];
#pragma unroll
    for (int k = 0; k < NT / 4; ++k)
      w0p[k] = *reinterpret_cast<const f32x4*>(a.w0 + rbase * a.ldw0 + i1 + c * NT + 4 * k);
    __syncthreads();                                           // the four phases' columns are in the transpose tiles
    {
      float* tq = T0 + row_l * TLD + c * NT;
#pragma unroll
      for (int k = 0; k < NT / 4; ++k) {
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(tq + 4 * k);
        const f32x4 x4 = q4 - w0p[k];                          // factor form: Q1 - W0
        if (active) {
          *reinterpret_cast<f32x4*>(a.W + rbase * a.ldw + i1 + c * NT + 4 * k) = q4;
          *reinterpret_cast<f32x4*>(a.Err + rbase * a.lde + 128 * b + c * NT + 4 * k) = x4;
        }
        *reinterpret_cast<f32x4*>(tq + 4 * k) = x4;            // (read and written by this lane only)
      }
      if (a.codes && a.codes_wide && active) {
        const uint8_t* tc = T1b + row_l * CLD + c * NT;
        uint8_t* dst = a.codes + rbase * a.ldc + i1 + c * NT;
        if (NT == 8) *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(tc);
        else *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(tc);
      }
    }
    QS_STAMP(6);
    if constexpr (NREM > 0) {
      __syncthreads();                                         // the A-operand tile is complete
      if constexpr (NREM == 3) near2.prefetch(a, i1, wcol + 64, tid);
      near.run(a, T0, wcol, row0, tid);
      if constexpr (NREM == 3) near2.run(a, T0, wcol + 64, row0, tid);
    }
}

template <int L, bool GROUPED, int OCC>
__global__ __launch_bounds__(256, OCC) void quant_super_kernel(QuantSuperArgs a) {
  critical_path_priority();
  constexpr int ROWS = QS<L>::ROWS, NT = QS<L>::NT, TLD = QS<L>::TLD, UBUF = QS<L>::UBUF, UD = QS<L>::UD, CLD = QS<L>::CLD;
  extern __shared__ __attribute__((aligned(16))) float qs_lds[];
  constexpr int NBUF = QS<L>::NBUF;
  float* Us = qs_lds;                                          // [NBUF][UBUF]
  float* Ud = Us + NBUF * UBUF;                                // [NBUF][UD]
  float* T0 = Ud + NBUF * UD;                                  // [ROWS][TLD] Q1 of the block, then Q1 - W0 (the update's A operand)
  uint8_t* T1b = reinterpret_cast<uint8_t*>(T0 + ROWS * TLD);  // [ROWS][CLD] codes
  int* grp = reinterpret_cast<int*>(T1b + ROWS * CLD);         // [128]
  int* cmap = grp + 128;                                       // [128]

  const int tid0 = threadIdx.x;
  const int c0 = tid0 % L, row_l0 = tid0 / L;
  const int row0 = blockIdx.x * ROWS;
  const int row = row0 + row_l0;
  const bool active = row < a.R;
  const long rbase = active ? row : a.R - 1;
  const int base4 = ((tid0 & 63) & ~(L - 1)) << 2;
  const int swz = (c0 * NT) >> 6;

  float sc = 1.f, zr = 0.f;
  if (!GROUPED) {
    sc = a.scale_tab[rbase * a.tab_ld];
    zr = a.zero_tab[rbase * a.tab_ld];
  }
  float loss_row = a.loss[rbase];

  {                                                            // U rows of block 0, phase 0
    f32x4 first[4];
    qs_stage_fetch(a.U, a.ldu, a.s0, 0, tid0, first);
    qs_stage_store<L>(Us, Ud, 0, tid0, first);
  }
#pragma unroll 1
  for (int b = 0; b < a.nb; ++b) {
    const int i1 = a.s0 + 128 * b;
    int tid = tid0, c = c0, row_l = row_l0;
    asm volatile("" : "+v"(tid), "+v"(c), "+v"(row_l));            // (see qs_phase: keeps address arithmetic inside the loop)
    QS_STAMP(0);
    if (tid < 128) {
      grp[tid] = a.col_group ? a.col_group[i1 + tid] : 0;
      cmap[tid] = a.col_map ? a.col_map[i1 + tid] : i1 + tid;
    }
    __syncthreads();                                           // grp / cmap; the previous block's updates of W are visible
    float w[NT], psc[NT], pzr[NT];
    const float* wrow = a.W + rbase * a.ldw + i1;
#pragma unroll
    for (int t = 0; t < NT; ++t) w[t] = wrow[L * t + c];
    if (GROUPED) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int g = grp[L * t + c];
        psc[t] = a.scale_tab[rbase * a.tab_ld + g];
        pzr[t] = a.zero_tab[rbase * a.tab_ld + g];
      }
    }
    float loss = 0.f;
    const bool have_next = b + 1 < a.nb;
    QS_STAMP(1);
    qs_phase<L, GROUPED, 0>(a, i1 + 128, have_next, w, psc, pzr, sc, zr, Us, Ud, T0, T1b, cmap, i1, c, row_l, rbase, active, base4, swz, tid, loss);
    QS_STAMP(2);
    qs_phase<L, GROUPED, 1>(a, i1 + 128, have_next, w, psc, pzr, sc, zr, Us, Ud, T0, T1b, cmap, i1, c, row_l, rbase, active, base4, swz, tid, loss);
    QS_STAMP(3);
    qs_phase<L, GROUPED, 2>(a, i1 + 128, have_next, w, psc, pzr, sc, zr, Us, Ud, T0, T1b, cmap, i1, c, row_l, rbase, active, base4, swz, tid, loss);
    QS_STAMP(4);
    qs_phase<L, GROUPED, 3>(a, i1 + 128, have_next, w, psc, pzr, sc, zr, Us, Ud, T0, T1b, cmap, i1, c, row_l, rbase, active, base4, swz, tid, loss);
    QS_STAMP(5);
    loss_row += 0.5f * loss;                                   // gptq.py:274

    {
      const int nrem = a.nb - 1 - b;                           // (workgroup-uniform)
      if (nrem == 3) qs_retire<L, 3>(a, T0, T1b, i1, b, c, row_l, row0, rbase, active, tid);
      else if (nrem == 2) qs_retire<L, 2>(a, T0, T1b, i1, b, c, row_l, row0, rbase, active, tid);
      else if (nrem == 1) qs_retire<L, 1>(a, T0, T1b, i1, b, c, row_l, row0, rbase, active, tid);
      else qs_retire<L, 0>(a, T0, T1b, i1, b, c, row_l, row0, rbase, active, tid);
    }
    QS_STAMP(7);
  }
  if (active && c0 == 0) a.loss[row] = loss_row;
}

}  // namespace gptq

using namespace gptq;

#ifdef GPTQ_DIAG
extern "C" int gptq_diag_qs_stamps(unsigned long long* out8) {
  GPTQ_CHECK_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(qs_stamps), sizeof(unsigned long long) * 8));
  return GPTQ_OK;
}
#endif

// 0: the per-block path; 8 / 16: lanes per row of quant_super_kernel.  Default by rows: 16 lanes up to 8192 rows (one or
// two waves per SIMD), 8 beyond (more rows per workgroup, three workgroups per compute unit).  GPTQ_QS_LANES overrides.
int quant_super_lanes(int R) {
  const char* e = getenv("GPTQ_QS_LANES");                 // (read per call: tests switch paths inside one process)
  const int env = e ? atoi(e) : -1;
  if (env == 0 || env == 8 || env == 16) return env;
  // measured (solve of R x 4096, act-order, per-block path -> this kernel): 4096 rows 4.23 -> 3.77 ms (16 lanes); 12288
  // 6.54 -> 6.05, 16384 7.51 -> 6.91, 22016 8.95 -> 8.75 (8 lanes, two workgroups per compute unit; three per compute unit
  // -- 168 registers, one U-row buffer -- measured SLOWER, 9.70: three 40 KB workgroups leave no room for the helper
  // stream's 67 KB update tiles, the two streams then take turns instead of sharing the chip)
  return R <= 8192 ? 16 : R <= 24576 ? 8 : 0;
}

// Launch for one super-block.  Preconditions (checked by the caller, gptq_fasterquant_rows): factor form, 16-byte
// aligned rows of W / W0 / Err / U, static or no groups.
int launch_quant_super(const QuantSuperArgs& a, bool grouped, int lanes, hipStream_t s) {
#define QS_LAUNCH(LL, GG, OCC)                                                                                     \
  do {                                                                                                             \
    GPTQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&quant_super_kernel<LL, GG, OCC>),           \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)QS<LL>::LDS_BYTES));     \
    quant_super_kernel<LL, GG, OCC><<<cdiv(a.R, QS<LL>::ROWS), 256, QS<LL>::LDS_BYTES, s>>>(a);                   \
  } while (0)
  if (lanes == 16) {
    if (grouped) QS_LAUNCH(16, true, 2); else QS_LAUNCH(16, false, 2);
  } else {
    if (grouped) QS_LAUNCH(8, true, 2); else QS_LAUNCH(8, false, 2);
  }
#undef QS_LAUNCH
  GPTQ_CHECK_LAUNCH("quant_super_kernel");
  return GPTQ_OK;
}
