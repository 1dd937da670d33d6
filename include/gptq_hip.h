/*
 * gptq_hip.h -- C ABI of libgptq_hip.so, the MI355X (gfx950) GPTQ hot path.
 *
 * Drop-in boundary for the reference's per-Linear GPTQ path.  Every entry point
 * names the reference interface it replaces (paths relative to the reference
 * checkout).  All pointers are DEVICE pointers borrowed for the duration of the
 * call; nothing is allocated on behalf of the caller; outputs are written in
 * place.  `stream` is a hipStream_t (NULL = default stream).  Calls only
 * enqueue work on `stream` and never synchronise the device.
 *
 * Return value: 0 = ok, GPTQ_ERR_INVALID = bad argument/shape,
 * GPTQ_ERR_HIP = HIP runtime error, GPTQ_ERR_UNSUPPORTED = option outside the
 * hot-path scope.  gptq_last_error() returns a thread-local message.
 */
#ifndef GPTQ_HIP_H
#define GPTQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPTQ_OK 0
#define GPTQ_ERR_INVALID 1
#define GPTQ_ERR_HIP 2
#define GPTQ_ERR_UNSUPPORTED 3

/* element types of activations / weights handed to the library */
#define GPTQ_F32 0
#define GPTQ_F16 1
#define GPTQ_BF16 2

typedef void* gptq_stream_t;

int gptq_hip_abi_version(void);
const char* gptq_last_error(void);

/* ---------------------------------------------------------------------------
 * Hessian running mean -- replaces GPTQ.add_batch (gptq.py:38-65):
 *   H <- H * n/(n+b) + (2/(n+b)) * X^T X,   n = nsamples_before, b = batch.
 * X: [tokens, C] row-major, leading dimension ldx elements, dtype x_dtype
 * (fp16/bf16 activations are widened to fp32 on load, gptq.py:62); H: [C, C]
 * fp32, leading dimension ldh.  Only the upper triangle (row <= col) of H is
 * maintained; call gptq_symmetrize before reading H as a full matrix.
 * ------------------------------------------------------------------------- */
int gptq_hessian_accum(float* H, int ldh, const void* X, int x_dtype, int ldx, int C, int tokens,
                       int nsamples_before, int batch, gptq_stream_t stream);

/* Same update for a batch of n_x equally shaped slabs X[i] [tokens_each, C] (X = HOST array of n_x
 * device pointers) in ONE pass over H:  H <- H * n/(n+b) + (2/(n+b)) * sum_i X[i]^T X[i],
 * b = batch_total samples -- gptq.py:42-65 with a multi-sample batch (tmp = b, gptq.py:44).
 * Deferring several hook calls into one launch divides the fp32 H read-modify-write traffic by n_x. */
int gptq_hessian_accum_multi(float* H, int ldh, const void* const* X, int n_x, int x_dtype, int ldx,
                             int C, int tokens_each, int nsamples_before, int batch_total,
                             gptq_stream_t stream);

/* Several independent problems of identical shape in one launch (the Linears of a transformer block
 * that share in_features: q/k/v/out/fc1 ... each bring too few 128x128 tiles to fill 256 CUs alone).
 * H: host array of n_prob device pointers; X: host array of n_prob*n_x device pointers (problem-major);
 * nsamples_before: host array [n_prob]. */
int gptq_hessian_accum_group(int n_prob, float* const* H, int ldh, const void* const* X, int n_x,
                             int x_dtype, int ldx, int C, int tokens_each, const int* nsamples_before,
                             int batch_total, gptq_stream_t stream);

/* Several independent problems of DIFFERENT widths in one call: every Linear of a transformer block hooked in the
 * same forward passes (GPTQ.add_batch, gptq.py:38-65, once per Linear and sample in the reference).  ldh, ldx, C,
 * nsamples_before: host arrays [n_prob]; H, X as in gptq_hessian_accum_group; all problems bring n_x slabs of
 * tokens_each rows.  Problems the 256x256-tile kernel accepts (16-bit activations, C % 256 == 0, 16-byte aligned
 * rows) share its launches, so their tiles fill the chip together; the others are dispatched per shape.
 * n_cu: work decomposition hint for THIS call -- size the launches for at most n_cu compute units (0 = the whole
 * device): the 256x256-tile kernels then launch n_cu workgroups that stride over the whole-K tiles and the K-split
 * runs.  For Hessian updates that run on a stream beside latency-bound work of other streams (the solves of other
 * Linears), which needs free compute units to make progress.  Results do not depend on it beyond fp32 rounding
 * (it moves the boundary between whole-K tiles and K-split runs). */
int gptq_hessian_accum_mixed(int n_prob, float* const* H, const int* ldh, const void* const* X, int n_x,
                             int x_dtype, const int* ldx, const int* C, int tokens_each,
                             const int* nsamples_before, int batch_total, int n_cu, gptq_stream_t stream);

/* Mirror the upper triangle of A [n, n] into the lower triangle. */
int gptq_symmetrize(float* A, int lda, int n, gptq_stream_t stream);

/* ---------------------------------------------------------------------------
 * Affine grid -- replaces Quantizer.find_params (quant.py:37-77, perchannel,
 * weight=True, mse=False) and quantize() (quant.py:6-10).
 * find_params: for every row r and every group j (columns c0 + j*gsize ...,
 * clipped to c1), writes scale[r*tab_ld + g0 + j], zero[...].
 * ------------------------------------------------------------------------- */
int gptq_find_params(const float* W, int ldw, int R, int c0, int c1, int gsize, int bits, int sym,
                     float* scale, float* zero, int tab_ld, int g0, gptq_stream_t stream);

/* Round-to-nearest onto a per-row grid (the `--nearest` baseline, opt.py:289-300):
 * X[r, c] <- scale[r] * (clamp(rint(X/scale[r]) + zero[r], 0, 2^bits-1) - zero[r]). */
int gptq_quantize_rows(float* X, int ldx, int R, int C, const float* scale, const float* zero,
                       int bits, gptq_stream_t stream);

/* ---------------------------------------------------------------------------
 * Damped inverse factor -- replaces gptq.py:174-180
 *   (H[diag] += percdamp*mean(diag H); cholesky; cholesky_inverse; cholesky(upper)).
 * In: H [C, C] fp32, upper triangle valid (dead-column fix already applied).
 * perm (nullable, int32[C]): act-order permutation, H <- H[perm][:, perm] first
 * (gptq.py:168).  Out: H is overwritten by U (upper triangular, U^T U =
 * (H + damp I)^-1, zero below the diagonal).  info (device int32[1], nullable)
 * is set non-zero if a non-positive pivot was met (torch raises LinAlgError).
 * ------------------------------------------------------------------------- */
size_t gptq_hinv_workspace_bytes(int C);
int gptq_hinv_upper(float* H, int ldh, int C, float percdamp, const int32_t* perm, int32_t* info,
                    void* workspace, size_t workspace_bytes, gptq_stream_t stream);

/* ---------------------------------------------------------------------------
 * The same chain WITHOUT the triangular inverse (half its flops), for the factor
 * form of the column loop's trailing updates (gptq_fasterquant uses it whenever
 * C % 128 == 0, blocksize == 128 and every dynamic group lies inside one block).
 * Same arguments as gptq_hinv_upper; C must be a multiple of 128.
 * Out, with R = U^-1 (H + damp I = R R^T, R upper triangular) and 128 x 128 blocks:
 *   diagonal blocks    U_kk = R_kk^-1 (the block of U the in-block loop of
 *                      gptq.py:201-271 reads), zero under their diagonals;
 *   blocks above them  Rt[B, blk] = R[B, blk] * U_blk,blk  (the factor's rows,
 *                      pre-multiplied so that  W1[:, blk] = W0[:, blk] -
 *                      sum_B (Q - W0)[:, B] Rt[B, blk]  needs no further product);
 *   blocks below them  untouched.
 * Workspace: gptq_hinv_workspace_bytes(C).
 * ------------------------------------------------------------------------- */
/* 1 if gptq_fasterquant with these parameters takes the factor form (H is left as gptq_rfactor_upper leaves it),
 * 0 if it takes the inverse form (H is left holding U). */
int gptq_fasterquant_factor_form(int C, int blocksize, int groupsize, int static_groups);
int gptq_rfactor_upper(float* H, int ldh, int C, float percdamp, const int32_t* perm, int32_t* info,
                       void* workspace, size_t workspace_bytes, gptq_stream_t stream);

/* ---------------------------------------------------------------------------
 * gptq_rfactor_upper in pieces, for a factorization whose OUTER PANELS (512 columns = 4 blocks of 128) are spread over
 * several GPUs.  The reference has no multi-GPU code; what these replace is still gptq.py:174-180.
 * Every rank holds the same H (dead-column fix applied: gptq_solve_prepare) and the same workspace layout
 * (gptq_hinv_workspace_bytes(C); C % 128 == 0): A = [C, C] floats at byte offset 0 (lower triangle: the reversed damped
 * Hessian, then its Cholesky factor), Linv = [C, C] floats at byte offset C * C * 4 (only its diagonal 128-blocks).
 *   gptq_chol_begin   builds A from H (damp = percdamp * mean(diag H); perm = act-order permutation or NULL).
 *   gptq_chol_panel   factorizes outer panel p0 (first block, a multiple of 4): rows p0 * 128 ... C of the block columns
 *                     [p0, p0 + 4) of A become the factor's, the diagonal blocks of Linv their inverses.  Its columns
 *                     must already carry the updates of every earlier outer panel.  The owning rank then sends
 *                     A[p0 * 128 :, p0 * 128 : (p0 + 4) * 128] and those four Linv blocks to the other ranks.
 *   gptq_chol_update  A[:, block columns [tn0, tn1)] -= L[:, panel p0] L[those columns' rows, panel p0]^T (rank-512, lower
 *                     tiles only), tn0 >= p0 + 4: each rank calls it for the block columns of the outer panels IT owns.
 *   gptq_chol_end     H <- Rt above the diagonal blocks, U_kk inside them (as gptq_rfactor_upper leaves it), from the
 *                     complete A / Linv.
 * With one rank, begin; { panel(p0); update(p0, p0 + 4, C / 128) } for p0 = 0, 4, ...; end  is gptq_rfactor_upper.
 * ------------------------------------------------------------------------- */
int gptq_chol_begin(float* H, int ldh, int C, float percdamp, const int32_t* perm, int32_t* info, void* workspace,
                    size_t workspace_bytes, gptq_stream_t stream);
int gptq_chol_panel(void* workspace, int C, int p0, int32_t* info, gptq_stream_t stream);
int gptq_chol_update(void* workspace, int C, int p0, int tn0, int tn1, gptq_stream_t stream);
int gptq_chol_end(float* H, int ldh, int C, void* workspace, gptq_stream_t stream);

/* ---------------------------------------------------------------------------
 * One lazy-batch block of the column loop -- replaces gptq.py:195-274 for a
 * single block [i1, i1+count) with the plain affine quantizer (gptq.py:251-264).
 * W [R, C] fp32 working weights: columns [i1, i1+count) are replaced by the
 * dequantized Q1; Err [R, blocksize] receives Err1 (zero past count).
 * U: upper factor [C, C].  The grid of column i1+i of row r is
 *   scale_tab[r*tab_ld + col_group[i1+i]]  (col_group NULL => column 0).
 * codes (nullable, uint8 [R, ldc]) receives the integer code of column i1+i at
 * codes[r*ldc + (col_map ? col_map[i1+i] : i1+i)].
 * loss [R] fp32 accumulates sum_i (w-q)^2 / d^2 / 2 per row (gptq.py:267,274).
 * Bit-exact with the reference loop when fed identical (W1, Hinv1, scale, zero).
 * ------------------------------------------------------------------------- */
int gptq_quant_block(float* W, int ldw, int R, int C, int i1, int count, int blocksize,
                     const float* U, int ldu, const float* scale_tab, const float* zero_tab,
                     int tab_ld, const int32_t* col_group, int bits, float* Err, uint8_t* codes,
                     int ldc, const int32_t* col_map, float* loss, gptq_stream_t stream);

/* ---------------------------------------------------------------------------
 * Whole solve -- replaces GPTQ.fasterquant (gptq.py:126-305), default branch.
 * W [R, C] fp32: in = layer weights (original column order), out = dequantized
 * Q in original column order (gptq.py:300-305).  H [C, C] fp32 (upper triangle
 * valid) is consumed and left holding the upper factor U (permuted order) -- or,
 * where the factor form applies (see gptq_rfactor_upper), what that call leaves.
 * scale_io/zero_io [R]: if preset != 0 they hold a ready grid (gptq.py:181);
 * on return they hold the grid left in the quantizer (last one used).
 * group_scale/group_zero (nullable, [R, n_groups]) receive the per-group grids
 * (n_groups = ceil(C/groupsize)) when groupsize > 0.
 * perm_out (nullable, int32[C]) receives the act-order permutation.
 * codes (nullable, uint8 [R, C]) receives integer codes in original column order.
 * blocksize: any value in 1 ... 256 (gptq.py:127; the drivers use 128).
 * error_out (device fp32[1]) receives sum(Losses) (gptq.py:294).
 * info (device int32[1], nullable): non-zero if the Hessian was not positive definite.
 * ------------------------------------------------------------------------- */
size_t gptq_fasterquant_workspace_bytes(int R, int C, int blocksize, int groupsize, int actorder,
                                        int static_groups);
int gptq_fasterquant(float* W, int ldw, float* H, int ldh, int R, int C, int bits, int sym,
                     int blocksize, float percdamp, int groupsize, int actorder, int static_groups,
                     float* scale_io, float* zero_io, int preset, float* group_scale,
                     float* group_zero, int32_t* perm_out, uint8_t* codes, float* error_out,
                     int32_t* info, void* workspace, size_t workspace_bytes, gptq_stream_t stream);

/* The same solve with the per-row losses as an extra output: row_loss (nullable, device fp32 [R]) receives
 * sum_j Losses[r, j] of every row (error_out = their sum).  Rows are independent problems given H (per-row grids,
 * per-row error feedback, gptq.py:262-276), so the R rows may be the CONCATENATION of several Linears that were fed
 * the same inputs (q/k/v, gate/up: same H): one factorization chain serves all of them, the column loop runs over
 * all their rows at once, and row_loss lets the caller report each Linear's own `error`. */
int gptq_fasterquant_rows(float* W, int ldw, float* H, int ldh, int R, int C, int bits, int sym,
                          int blocksize, float percdamp, int groupsize, int actorder, int static_groups,
                          float* scale_io, float* zero_io, int preset, float* group_scale,
                          float* group_zero, int32_t* perm_out, uint8_t* codes, float* error_out,
                          float* row_loss, int32_t* info, void* workspace, size_t workspace_bytes,
                          gptq_stream_t stream);

/* The head of the solve alone: dead-column fix of H's diagonal (gptq.py:143-145; dead_out [C] int32 flags) and, with
 * actorder, perm_out = argsort(diag H, descending) (gptq.py:166).  scratch: [C] floats.  What a factorization spread over
 * several GPUs (gptq_chol_*) needs before it starts. */
int gptq_solve_prepare(float* H, int ldh, int C, int actorder, int32_t* dead_out, int32_t* perm_out, float* scratch,
                       gptq_stream_t stream);
/* gptq_fasterquant_rows for an H that ALREADY holds what gptq_rfactor_upper leaves (factor form only:
 * gptq_fasterquant_factor_form must say 1), factorized elsewhere from the H that gptq_solve_prepare fixed, with that call's
 * dead flags and (actorder) permutation: everything of gptq.py:126-305 but lines 143-144, 166 and 174-180. */
int gptq_fasterquant_rows_factored(float* W, int ldw, float* H, int ldh, int R, int C, int bits, int sym, int blocksize,
                                   int groupsize, int actorder, int static_groups, float* scale_io, float* zero_io,
                                   int preset, float* group_scale, float* group_zero, const int32_t* dead_in,
                                   const int32_t* perm_in, uint8_t* codes, float* error_out, float* row_loss,
                                   int32_t* info, void* workspace, size_t workspace_bytes, gptq_stream_t stream);

/* ---------------------------------------------------------------------------
 * Bit packing -- replaces Quant3Linear.pack (quant.py:152-187) and the int4
 * layout of Quant4Linear.__init__ (zeroShot/models/quant.py:176-185).
 * weight: [out, in] row-major (dtype w_dtype, leading dimension ldw);
 * scales [out] fp32; zeros [out] fp32 = zero*scale (quant.py:153);
 * qweight: int32 [in/32*bits, out] (bits = 3: 96-bit little-endian stream per
 * 32 codes; bits = 4: nibble i%8 of word i/8).  in % 32 == 0 (3-bit) / in % 8 == 0.
 * gptq_pack_codes packs ready integer codes (uint8 [out, in]) instead.
 * ------------------------------------------------------------------------- */
int gptq_pack_weights(const void* weight, int w_dtype, int ldw, int out_features, int in_features,
                      const float* scales, const float* zeros, int bits, int32_t* qweight,
                      gptq_stream_t stream);
int gptq_pack_codes(const uint8_t* codes, int ldc, int out_features, int in_features, int bits,
                    int32_t* qweight, gptq_stream_t stream);

/* Packed -> dense weights: W[o][i] = scale[g][o] * (code - zero[g][o]), g = i / groupsize, tables
 * [in/groupsize, out] with `zero` the INTEGER zero point (quant.py:10 arithmetic); groupsize <= 0 means one
 * group.  Used to rebuild the Linears other ranks quantized and packed checkpoints. */
int gptq_dequant_packed(const int32_t* qweight, const float* scale, const float* zero, int out_features,
                        int in_features, int bits, int groupsize, void* weight, int w_dtype, int ldw,
                        gptq_stream_t stream);

/* ---------------------------------------------------------------------------
 * Packed dequant mat-vec -- replaces quant_cuda.vecquant3matmul /
 * vecquant3matmul_faster (quant_cuda.cpp:15-29, quant_cuda_kernel.cu:31-244):
 *   mul[col] += sum_k (scales[col]*q[k,col] - zeros[col]) * vec[k]
 * vec: [in] (vec_dtype GPTQ_F32, or GPTQ_F16 for the "faster" form); mat int32
 * [height, width], height = in/32*bits, width = out; mul/scales/zeros fp32 [width].
 * vecquant4matmul has no reference kernel (only the layout is pinned, see above).
 * ------------------------------------------------------------------------- */
int gptq_vecquant3matmul(const void* vec, int vec_dtype, const int32_t* mat, float* mul,
                         const float* scales, const float* zeros, int height, int width,
                         gptq_stream_t stream);
int gptq_vecquant4matmul(const void* vec, int vec_dtype, const int32_t* mat, float* mul,
                         const float* scales, const float* zeros, int height, int width,
                         gptq_stream_t stream);

/* Grouped grids (groupsize = 128 etc.; SURVEY row f4 -- the reference cannot pack grouped models at all,
 * Quant3Linear holds per-row scalars only, quant.py:144-145): scales / zeros are [in/groupsize, width]
 * (row = group of ORIGINAL input columns, zeros = zero*scale), groupsize % 32 == 0:
 *   mul[col] += sum_k (scales[k/groupsize][col]*q[k,col] - zeros[k/groupsize][col]) * vec[k]. */
int gptq_vecquant_matmul_grouped(const void* vec, int vec_dtype, const int32_t* mat, float* mul,
                                 const float* scales, const float* zeros, int height, int width,
                                 int bits, int groupsize, gptq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GPTQ_HIP_H */
