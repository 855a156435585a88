/* C ABI of libbasd_hip.so -- the MI355X (gfx950) kernels behind the BASD loss path.
 *
 * The reference (indrajeetadityaroy9/vit-inductive-bias-distillation) is pure Python: its
 * "FFI" for this path is the set of torch operator call sites in src/losses/<module>.py.  Each entry
 * point below replaces one (or a fused group) of those call sites; the file:line it stands in
 * for is quoted on every declaration (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is DEVICE memory unless marked host; sizes/strides are in ELEMENTS;
 *   - no entry point allocates, frees or synchronises; all work is queued on `stream`
 *     (pass torch.cuda.current_stream().cuda_stream); scratch buffers are caller-provided;
 *   - return value: 0 = ok, <0 = invalid argument / unsupported shape (BASD_E*), >0 = hipError_t;
 *   - dtype codes: 0 = fp32, 1 = bf16 (inputs only; all arithmetic and outputs are fp32/fp64);
 *   - entry points are re-entrant and keep no global mutable state, with a few process-wide test / tuning hooks as the
 *     only exceptions: basd_tridiag_tuning, basd_jacobi_tuning, basd_jacobi_ordering, basd_gemm_tuning, basd_procrustes_tuning (none is called by the
 *     loss).
 */
#ifndef BASD_HIP_H
#define BASD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#define BASD_OK 0
#define BASD_EINVAL (-1)
#define BASD_EUNSUPPORTED (-2)
#define BASD_DTYPE_F32 0
#define BASD_DTYPE_BF16 1

/* ---- dense contractions (fp32 MFMA, v_mfma_f32_32x32x2_f32) ------------------------------- */

/* C[z] (M x N, ldc) = beta * C[z] + scale * A[z] (M x K) * B[z]^T (N x K) - 1 bias^T   (bias nullable).
 * replaces: `tokens.reshape(-1, D_t) @ proj_t.T`  layer_selector.py:72,:135;
 *           `features @ features.T / M`           layer_selector.py:15;
 *           `U_s.T @ U_t`                          layer_selector.py:99.
 * A element (m, k): a + z*a_batch_stride + (m / a_rows_per_batch)*a_sb + (m % a_rows_per_batch)*a_sn + k*a_sd
 * (so (B, N, D) token views, CLS-sliced or channel-major, are consumed in place). B: fp32 row-major.
 * colsum_part (nullable, needs beta == 0): batch * ceil(M/128) * N floats; the kernel's epilogue leaves the column
 * sums of every 128-row tile of C there.  col_mean (nullable, needs colsum_part): batch * N floats, the column means
 * of C folded from those -- `z.mean(dim=0)` of layer_selector.py:35 without another pass over z. */
int basd_gemm_nt(const void* a, int a_dtype, long a_sb, long a_sn, long a_sd, int a_rows_per_batch,
                 long a_batch_stride, const float* b, long ldb, long b_batch_stride, int M, int N, int K, int batch,
                 float* c, long ldc, long c_batch_stride, float scale, const float* bias, float beta,
                 float* colsum_part, float* col_mean, hipStream_t stream);

/* Suggested split count over the contraction rows for basd_gemm_tn. */
int basd_gemm_tn_splits(int krows);

/* C[z] (M x N) = scale * (A[z] - 1 mean_a^T)^T (B[z] - 1 mean_b^T), contraction over `krows` rows.
 * replaces: `features.T @ features / M`  layer_selector.py:13  (mean = NULL, scale = 1/M) and the
 *           centred z^T z behind `torch.linalg.svd(z - z.mean(0))`  layer_selector.py:35-36, :90-92;
 *           also the per-sample K' A' product of the Procrustes backward (batch > 1, splits = 1).
 * `slabs`: batch*splits*M*N floats of scratch when splits > 1 (deterministic split-K reduction). */
int basd_gemm_tn(const void* a, const void* b, int dtype, long a_sb, long a_sn, long a_sd, long a_batch_stride,
                 long b_sb, long b_sn, long b_sd, long b_batch_stride, int rows_per_batch, int krows, int M, int N,
                 int batch, const float* mean_a, const float* mean_b, int splits, float* slabs, float* c, long ldc,
                 long c_batch_stride, float scale, hipStream_t stream);

int basd_colmean_parts(int rows);

/* mean[z][c] = (1/rows) * sum_r X[z](r, c).   replaces `z.mean(dim=0)`  layer_selector.py:35, :91.
 * `partial`: batch*parts*cols floats of scratch. */
int basd_colmean(const void* x, int dtype, long sb, long sn, long sd, int rows_per_batch, long batch_stride,
                 int rows, int cols, int batch, int parts, float* partial, float* mean, hipStream_t stream);

/* Test / tuning hook: 1 (default) = the Gram launches (basd_syrk_multi) run on the bf16 matrix cores with every staged fp32
 * value cut into three bf16 pieces (exact) and the six leading piece products accumulated in fp32 -- fp32-grade results at
 * 2.7x the fp32 MFMA ceiling; 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32). */
int basd_gemm_tuning(int split_bf16);
int basd_gemm_tuning_get(void);       /* the current setting */

/* means[z][c] for a DEVICE table of n_mats same-layout matrices (one launch for all extraction layers of
 * layer_selector.py:88-91).  `partial`: n_mats*parts*cols floats of scratch.  vec_ok: caller asserts every base
 * pointer is 16-byte aligned. */
int basd_colmean_multi(const void* const* x_ptrs, int dtype, long sb, long sn, long sd, int rows_per_batch, int rows,
                       int cols, int n_mats, int parts, float* partial, float* means, int vec_ok,
                       hipStream_t stream);

int basd_syrk_splits(int krows, int cols, int n_mats);

/* out[z] (cols x cols, symmetric) = scales[z] * (X_z - 1 means[z]^T)^T (X_z - 1 means[z]^T),  z < n_mats.
 * replaces the per-layer Gram matrices of the selector in one launch: `features.T @ features / M`
 * layer_selector.py:13 (means row = 0, scale 1/M) and the centred z^T z behind the thin SVD :35-36, :90-92.
 * Only lower-triangular 128x128 tiles are computed (half the operand traffic of basd_gemm_tn), then mirrored.
 * x_ptrs: DEVICE array of base pointers; element (k, c) of X_z at x_ptrs[z] + (k / rows_per_batch)*sb +
 * (k % rows_per_batch)*sn + c*sd.  means / scales (nullable): n_mats*cols / n_mats floats on the device.
 * slabs: n_mats*splits*cols*cols floats of scratch (splits from basd_syrk_splits).  vec_ok: caller asserts
 * every base pointer is 16-byte aligned.  fold (nullable): (n_mats - fold_from) x fold_parts x cols row-tile column
 * sums as left by basd_gemm_nt's epilogue; matrices z >= fold_from take their means from it (sum / krows) instead
 * of `means`, folded inside the kernel. */
int basd_syrk_multi(const void* const* x_ptrs, int dtype, long sb, long sn, long sd, int rows_per_batch, int krows,
                    int cols, int n_mats, const float* means, const float* scales, int splits, float* slabs,
                    float* out, long out_stride, int vec_ok, const float* fold, int fold_parts, int fold_from,
                    hipStream_t stream);

/* ---- one-sided Jacobi SVD / symmetric eigensolver ------------------------------------------ */

int basd_jacobi_workspace_ints(int batch, int max_sweeps);
/* 1 when basd_jacobi_onesided takes large batches of plain n x n matrices with 4 lanes per column pair in LDS (orders
 * 40..144): basd_procrustes_forward_fused then solves the TRANSPOSED cores, without riding rows. */
int basd_jacobi_plain4_fits(int n);

/* In-place one-sided Jacobi on `batch` column-major matrices (rows_tot x n, leading dim rows_tot):
 * right rotations orthogonalise the first rows_dot rows; colnorm receives the column norms.
 * replaces the LAPACK calls behind torch.linalg.eigvalsh (layer_selector.py:16), torch.linalg.svd
 * (:36, :92), torch.linalg.svdvals (:99) and torch.linalg.matrix_norm(ord="nuc") (relational.py:48).
 * n_arr (nullable): per-matrix order for square problems.  flags: basd_jacobi_workspace_ints() ints.
 * tol_cos: stop when every pair has |cos| <= tol_cos over a full sweep (<= 0: eps * sqrt(rows_dot)). */
/* Test / tuning hook: lanes per column pair of the LDS-resident solver; 0 = automatic (batches >= 256 of stacked
 * matrices: 4 lanes for n >= 40, 8 for n >= 16 -- fewer instruction issues per matrix round; else one DPP row of 16),
 * 4 / 8 / 16 = forced where the shape allows. */
int basd_jacobi_tuning(int lanes_per_pair);
/* Test / tuning hook: ordering of the plain batched solver (no riding rows, batches >= 128 of orders 40..200: the
 * transposed Procrustes cores) -- 1 (default) = odd-even ordering with the columns held in registers and one column per
 * pair-step passed through LDS (two 196 x 196 matrices share a CU), 0 = round-robin ordering through LDS. */
int basd_jacobi_ordering(int odd_even);

int basd_jacobi_onesided(float* W, long batch_stride, int rows_dot, int rows_tot, int n, int batch,
                         const int* n_arr, float* colnorm, int colnorm_stride, int max_sweeps, float tol_cos,
                         int* flags, int* sweeps_out, hipStream_t stream);

/* The same for stacked square cores [M; L_b] (2n x n, leading dimension 2n; the Procrustes cores of relational.py:48)
 * whose 2n rows do not fit LDS while n rows do (n = 129..196: cfg-5's 144, cfg-4's 196): pass 1 solves the top n x n
 * in LDS and logs every rotation, pass 2 replays the log on the riding rows in LDS -- the same rotations in the same
 * order as the one-kernel solver, without the block solver's trips through L2 / HBM.  workspace: device memory of
 * basd_jacobi_twopass_workspace_bytes() bytes, 8-byte aligned (0 bytes = shape not covered: BASD_EUNSUPPORTED, use
 * basd_jacobi_onesided). */
long basd_jacobi_twopass_workspace_bytes(int n, int batch, int max_sweeps);
int basd_jacobi_stacked_twopass(float* W, long batch_stride, int n, int batch, float* colnorm, int colnorm_stride,
                                int max_sweeps, float tol_cos, void* workspace, int* sweeps_out, hipStream_t stream);

/* Sort column norms descending; optionally emit the top-kmax normalised columns as rows
 * (`Vt[:k]` of layer_selector.py:37, :97). */
int basd_sort_extract(const float* W, long batch_stride, int rows, int rows_tot, int n, int batch,
                      const float* colnorm, int colnorm_stride, float* vals_desc, float* vecs, int kmax,
                      hipStream_t stream);

/* ---- tridiagonalisation-based symmetric eigen-solver (eigenvalues / leading eigenvectors only) ------- */

/* Householder tridiagonalisation A = Q T Q^T of `batch` symmetric n x n matrices (A destroyed):
 * first stage of the LAPACK path behind torch.linalg.eigvalsh (layer_selector.py:16) and the
 * Vt[:k] / S[:k] part of torch.linalg.svd (:36, :92).  d, e, tau: (batch, n); vh: (batch, n, n). */
long basd_tridiag_workspace_bytes(int n, int batch);

/* `work`: basd_tridiag_workspace_bytes(n, batch) bytes of 16-byte aligned device scratch.
 * Two stages: the trailing block of order <= 256 is factored in the registers of ONE CU per matrix (n <= 256: the
 * whole factorisation -- no workgroup waits for another); for n > 256 the first n - 256 steps run with the matrix
 * shared by up to 16 workgroups that exchange one 16-byte granule per row and step through `work` (all workgroups of
 * that launch must be resident together: at most half of what an occupancy query says the device holds of the kernel,
 * and never more than 128; beyond that the member count per matrix shrinks, down to 1 = no exchange).  Its last 32
 * bytes hold a status word that is non-zero afterwards if a workgroup of the shared stage gave up waiting for its
 * partners, and a trace of the first give-up (step + 1, row, member | matrix << 8, tag seen, tag wanted). */
int basd_tridiag(float* a, long a_batch_stride, int n, int batch, float* d, float* e, float* tau, float* vh,
                 void* work, hipStream_t stream);

/* basd_tridiag + the Marchenko-Pastur ranks (see basd_tridiag_mp_rank; layer_selector.py:16-19, :74) of the FIRST
 * rank_count matrices of the batch, computed by the workgroup that finishes each factorisation -- no separate launch on
 * the path the host waits for.  host_mirror (nullable): pinned host memory of rank_count + 8 ints that receives the
 * ranks and then the factorisation's 8 status words. */
int basd_tridiag_ranked(float* a, long a_batch_stride, int n, int batch, float* d, float* e, float* tau, float* vh,
                        void* work, int rank_count, double factor, int cap, int* rank_out, int* host_mirror,
                        void* mid_event /* nullable hipEvent_t: recorded behind the multi-workgroup stage */,
                        hipStream_t stream);

/* Test / tuning hook for basd_tridiag (the only process-wide setting of the library; its defaults come from the
 * BASD_TRIDIAG_{MEMBERS,PAD,LAG,THREADS,TAIL} environment variables, read ONCE when the library is loaded -- no
 * entry point calls getenv).  Negative = keep; reset != 0 restores the load-time values first.
 * members: workgroups per matrix in the shared stage; pad: workgroup-id padding between matrices; lag: member that
 * sleeps every step; threads: per member; tail: 0 = shared stage for the whole factorisation, 1 = the two-barrier
 * register-resident tail kernel (default), 2 = its four-barrier first form (kept for comparison tests). */
int basd_tridiag_tuning(int members, int pad, int lag, int threads, int tail, int reset);

/* All eigenvalues (descending) of the tridiagonals by Sturm-sequence bisection. */
int basd_tridiag_eigenvalues(const float* d, const float* e, int n, int batch, float* vals_desc, hipStream_t stream);

/* out (batch, k_stride, n) rows = Q x (transpose = 0) or Q^T x (transpose = 1) for the rows of x (batch, k, n), Q the
 * product of the reflectors of basd_tridiag.  With basd_tridiag_shifted_solve this gives (A - lambda I)^{-1} on the
 * orthogonal complement of the leading eigenvectors: the part of the backward of `torch.linalg.svd(...).Vh[:k]`
 * (layer_selector.py:92 under autograd) that involves the eigenvectors NOT computed. */
int basd_tridiag_apply_q(const float* tau, const float* vh, int n, int k, int batch, const float* x, float* out,
                         int k_stride, int transpose, hipStream_t stream);

/* x[z][t] = (T_z - shifts[z][t] I)^{-1} rhs[z][t], t < k: LU with partial pivoting + one solve with perturbed tiny
 * pivots (shifts are eigenvalues; the caller projects the singular direction out).  rhs, x: (batch, k, n). */
int basd_tridiag_shifted_solve(const float* d, const float* e, const float* shifts, int shift_stride, int n, int k,
                               int batch, const float* rhs, float* x, hipStream_t stream);

/* Marchenko-Pastur rank of each tridiagonal without its spectrum: the lower median eigenvalue by 1025-section
 * Sturm counts, the threshold median * factor rounded to fp32, and one Sturm count at the threshold
 * (layer_selector.py:16-19; `factor` = (1 + sqrt(D/M))^2 in float64 from the host, `cap` as :74).
 * This is the quantity the host reads back every step: host_mirror (nullable) is device-visible pinned host memory
 * of batch + 8 ints; the kernel writes the ranks and the 8 words at `status` (nullable: the status area of
 * basd_tridiag's workspace) straight into it. */
int basd_tridiag_mp_rank(const float* d, const float* e, int n, int batch, double factor, int cap, int* rank_out,
                         float* thr_out, const int* status, int* host_mirror, hipStream_t stream);

/* Leading k eigenvectors of the original matrices (inverse iteration on T, cluster re-orthogonalisation,
 * back-transformation with the reflectors), written as rows of vecs (batch, k_stride, n).
 * z: batch*k*n floats of scratch. */
int basd_tridiag_eigenvectors(const float* d, const float* e, const float* tau, const float* vh,
                              const float* vals_desc, int n, int k, int batch, float* z, float* vecs, int k_stride,
                              hipStream_t stream);

/* ---- selector epilogues --------------------------------------------------------------------- */

/* rank = min(#{eig > fp32(median_lower(eig) * factor)}, cap).  layer_selector.py:17-19, :74.
 * factor = (1 + (D/M)**0.5)**2 evaluated by the HOST in float64 (layer_selector.py:11,18). */
int basd_mp_rank(const float* vals_desc, int n, int batch, double factor, int cap, int* rank_out, float* thr_out,
                 hipStream_t stream);

/* d = sum(sw * acos(min(sigma, 1-eps))^2) / sum(sw) per (student layer, teacher layer) item.
 * layer_selector.py:99-105. */
int basd_grassmann_distance(const float* colnorm, int stride, const int* k_arr, const float* sw, int sw_stride,
                            const int* sw_index, int items, float* d_out, float* theta_out, hipStream_t stream);

/* The same for matrices that were zero-padded to a common order n_valid (everything outside the leading k x k block zero,
 * basd_selector_chain_tail): colnorm holds n_valid values per item in any order, the k largest are the cosines. */
int basd_grassmann_distance_padded(const float* colnorm, int stride, int n_valid, const int* k_arr, const float* sw,
                                   int sw_stride, const int* sw_index, int items, float* d_out, hipStream_t stream);

int basd_sqrt_clamp(const float* in, float* out, long count, hipStream_t stream);

/* Both Grams of the projected teacher tokens z = tokens proj_t^T (layer_selector.py:72 -> :13 and :35) from the centred
 * Gram of the tokens in THEIR OWN space, for teachers about as wide as the student (D_t <~ 1.5 D_s: ViT teachers) -- z is
 * never formed: c (batch, n, n) = proj_t G_c proj_t^T (two basd_gemm_nt), zbar (batch, n) = proj_t tbar;
 * out_c (nullable: c is symmetric already) = sym(c) (the centred Gram of z), out_u (nullable: the rank-one route of the MP
 * rank, basd_tridiag_mp_rank_rank1, never forms it) = (sym(c) + M zbar zbar^T) / M (its uncentred Gram / M, M = m_rows).  Also the second Gram of basd_selector_chain: c = the centred Gram of the projected
 * tokens themselves, zbar their column means -- one symmetric launch per layer instead of two. */
int basd_gram_finish(const float* c, const float* zbar, int n, int batch, long m_rows, float* out_u, float* out_c,
                     hipStream_t stream);

/* Everything of the selector that follows the rank read-back, queued by one call (layer_selector.py:36-37, :92,
 * :95-105): leading kmax eigenvectors of the E student / L centred teacher Grams from their basd_tridiag
 * factorisations (t_* / s_*: d, e, tau, vh, vals), S[:k], the teacher bases rotated by proj_s^T, the E x L cosine
 * matrices, their singular values and d_grass_sq (E, L) -> d_out.  Scratch (floats unless noted): z_s, v_s
 * (E kmax d_s), z_t, u_t, u_rot (L kmax d_s), sw (L kmax), cos (E L kmax^2), k_arr (E L ints), sigma (E L kmax),
 * flags (basd_jacobi_workspace_ints(E L, 20) ints); sw_index: (E L) ints, item e*L+l -> l. */
int basd_selector_tail(const float* t_d, const float* t_e, const float* t_tau, const float* t_vh, const float* t_vals,
                       const float* s_d, const float* s_e, const float* s_tau, const float* s_vh, const float* s_vals,
                       int d_s, int E, int L, int kmax, const int* ranks, const float* proj_s_t, float* z_s,
                       float* v_s, float* z_t, float* u_t, float* u_rot, float* sw, float* cos, int* k_arr,
                       const int* sw_index, float* sigma, int* flags, float* d_out, hipStream_t stream);

/* Marchenko-Pastur ranks of the UNCENTRED Grams from the factorisations of the CENTRED ones (layer_selector.py:13-19 from
 * the matrices of :35): z^T z = G_c + M zbar zbar^T and G_c = Q T Q^T give z^T z = Q (T + M w w^T) Q^T with w = Q^T zbar
 * (basd_tridiag_apply_q, transposed), and the eigenvalues of a rank-one modification are COUNTED without another
 * factorisation: #{below x} = #{negative pivots of T - x I} + [-1/M - w^T (T - x I)^-1 w < 0] - 1, in one fp64 Sturm
 * recurrence.  A teacher layer then needs one factorisation instead of two.  d, e: (batch, n); w: (batch, n); rho = M;
 * factor, cap, rank_out, status, host_mirror as basd_tridiag_mp_rank. */
int basd_tridiag_mp_rank_rank1(const float* d, const float* e, const float* w, int n, int batch, double rho,
                               double factor, int cap, int* rank_out, const int* status, int* host_mirror,
                               hipStream_t stream);

/* basd_tridiag_ranked queued BEFORE its input exists (orders 257..384, the one-kernel factorisation; BASD_EUNSUPPORTED
 * otherwise): its whole-CU workgroups take their CUs while the chip is still quiet and wait, asleep and bounded, until
 * *go_flag == go_value; set that word with basd_flag_set on the stream that produces the matrices, behind them.  If the
 * word does not arrive within go_budget polls of ~1.7 us (e.g. a profiler serialising kernels) the workgroups give up:
 * status word 0 = 2 (also in host_mirror[rank_count]); queue the plain basd_tridiag_ranked then.
 * replaces: the same call sites as basd_tridiag_ranked (layer_selector.py:16-19, :72-74). */
int basd_tridiag_ranked_gated(float* a, long a_batch_stride, int n, int batch, float* d, float* e, float* tau, float* vh,
                              void* work, int rank_count, double factor, int cap, int* rank_out, int* host_mirror,
                              const unsigned* go_flag, unsigned go_value, int go_budget, hipStream_t stream);
int basd_flag_set(unsigned* flag, unsigned value, hipStream_t stream);

/* The selector of one loss step queued by ONE call over three streams (layer_selector.py:69-74 `_estimate_ranks`,
 * :131-138 teacher subspaces, :86-105 `_mix_for_student_layer` up to d_grass_sq): teacher projections z_l =
 * tokens_l proj_t^T, the centred Gram of every z_l and its uncentred Gram / M from it (+ M zbar zbar^T: an addition, no
 * second symmetric launch), centred Grams of the E student layers, the
 * Householder tridiagonalisation of all 2L + E matrices with the Marchenko-Pastur ranks of the L uncentred ones
 * written to `ranks` (device) and `host_mirror` (pinned host: L ranks + the 8 status words of basd_tridiag_ranked)
 * by the kernel that finishes the factorisation, then -- on tail_stream, with the ranks taken from DEVICE memory --
 * what basd_selector_tail does.  The host only waits for ev_ranks (basd_event_synchronize) to refresh
 * `subspace_ranks` and to raise on rank 0 like the reference.
 *   kmax: eigenvectors computed per matrix in the tail, a HINT (the previous step's largest rank); any value >= the
 *         largest rank of this step gives the same d_grass_sq; if the ranks read back exceed it, call
 *         basd_selector_chain_tail(args, larger kmax) again.  0: do not queue the tail (first step: no hint yet).
 *   mode: 0 = one factorisation launch over all 2L + E matrices (student Grams formed on student_stream beside the
 *             teacher's projections);
 *         1 = teacher matrices first; the student side (Grams + factorisation, student_stream != chain_stream) is
 *             held back until the ranks are out;   2 = the same, not held back;
 *         3 = the same, held back until the teacher's Grams are done (ev_tg0) plus `release_delay` rounds of ~3.4 us:
 *             the student Grams -- the largest MFMA launch of the step -- then run beside the teacher's
 *             factorisation (two CUs) instead of beside its projection and Grams, and that factorisation's whole-CU
 *             workgroups have been placed before the student side's launches refill every free slot.
 *   streams: main_stream = the caller's (inputs are ready there; NULL: the caller has recorded ev_fork on it already,
 *         e.g. before it queued other work the chain need not wait for); events are opaque handles of basd_event_create:
 *         ev_fork / ev_student / ev_ranks / ev_tail / ev_tgram / ev_tg0 are recorded by the call; ev_slot_free (nullable) is waited for
 *         before anything is written: the ev_tail of the call that used these buffers last.
 *   Every field is 8 bytes wide.  Device buffers (floats unless noted), n = d_s, M_t = B n_t, nt = ceil(M_t / 128):
 *         z (L, M_t, n); z_sums (L, nt, n); z_means (L, n); z_ptrs: device table of L pointers [z_0..z_{L-1}];
 *         t_slabs (L t_splits n n), t_splits = basd_syrk_splits(M_t, n, L);
 *         s_partial (E s_parts n), s_parts = basd_colmean_parts(B n_s); s_means (E, n); s_slabs (E s_splits n n);
 *         grams, vh (2L + E, n, n); d, e, tau, vals (2L + E, n); tri_work: basd_tridiag_workspace_bytes(n, 2L + E)
 *         bytes (modes 1, 2: (n, 2L) and tri_work_s (n, E)); ranks (L ints);
 *         tail, K = kmax_cap >= kmax: zv, vecs ((L + E) K n); u_rot (L K n); sw (L K); cos (E L K K); sigma (E L K);
 *         d_out (E L); k_arr (E L ints); sw_index (E L ints, item e L + l -> l);
 *         jflags (basd_jacobi_workspace_ints(E L, 20) ints).
 * Returns BASD_EUNSUPPORTED when B n_t < d_s (the reference then forms the token-side Gram, layer_selector.py:14-15):
 * take the per-kernel entry points. */
typedef struct BasdSelectorChain {
    const void* const* teacher_host_ptrs;  /* HOST array of L device pointers: (B, n_t, d_t) views with common strides */
    long t_dtype, t_sb, t_sn, t_sd;
    const void* const* student_ptrs;       /* device table of E pointers: (B, n_s, d_s) views with common strides */
    long s_dtype, s_sb, s_sn, s_sd, s_vec_ok;
    const float* proj_t;                   /* (d_s, d_t) fp32 row-major */
    const float* proj_s_t;                 /* (d_s, d_s) fp32: proj_s^T */
    long E, L, B, n_s, n_t, d_s, d_t;
    double mp_factor;                      /* (1 + sqrt(d_s / M_t))^2, float64 on the host as layer_selector.py:11,18 */
    long rank_cap;                         /* d_s - 1 (layer_selector.py:74) */
    long kmax, kmax_cap, mode;
    float* z; float* z_sums; const void* const* z_ptrs; float* z_means; float* t_slabs; long t_splits;
    float* s_partial; float* s_means; float* s_slabs; long s_splits, s_parts;
    float* grams; float* d; float* e; float* tau; float* vh; float* vals; void* tri_work; void* tri_work_s;
    int* ranks; int* host_mirror;          /* host_mirror: pinned host memory, L + 8 ints (nullable) */
    int* student_status_mirror;            /* modes 1, 2: pinned host memory, 8 ints (nullable) */
    float* zv; float* vecs; float* u_rot; float* sw; float* cos; float* sigma; float* d_out;
    int* k_arr; const int* sw_index; int* jflags;
    hipStream_t main_stream, chain_stream, student_stream, tail_stream;
    void* ev_fork; void* ev_student; void* ev_ranks; void* ev_tail; void* ev_slot_free; void* ev_tgram; void* ev_tg0;
    long release_delay;                    /* mode 3: rounds of ~3.4 us between the teacher's factorisation launch and the student side's release */
    /* mode 3, nullable: the teacher's factorisation is queued FIRST, on fact_stream, and waits for go_flag (one device
     * word of the slot, set behind the teacher Grams to go_value != 0): see basd_tridiag_ranked_gated.  ev_ranks is then
     * recorded on fact_stream. */
    hipStream_t fact_stream; unsigned* go_flag; long go_value; long go_budget;
    /* measurement (all nullable; timed events of basd_event_create_timed, recorded on the stream of the launch they
     * bracket): tm_proj = behind the projections, tm_tgram = behind the teacher Grams, tm_scol0 / tm_scol1 = around the
     * student column means, tm_sgram = behind the student Grams, tm_tri0 = in front of the factorisation (ev_ranks ends
     * it), tm_mid = between its two stages, tm_spec = behind the spectra at the head of the tail */
    void* tm_proj; void* tm_tgram; void* tm_scol0; void* tm_scol1; void* tm_sgram; void* tm_tri0; void* tm_mid; void* tm_spec;
    /* Rank certificate (all four nullable together).  The reference raises inside forward when a teacher layer has MP
     * rank 0 (layer_selector.py:16-19 -> NaN weights -> torch.linalg.svd raises), which is what the host waits ~1.5 ms
     * for.  Behind the teacher Grams -- long before the factorisation -- one small kernel on cert_stream proves
     * "every rank >= 1" where it can: with eigenvalues l_1 >= ... >= l_n >= 0 of the uncentred Gram G = A + zbar zbar^T,
     *     l_1 >= max(|zbar|^2, ||G||_F^2 / tr G)   and   median <= min(tr G / c, tr A / (c - 1)),  c = n - (n-1)/2,
     * so "left > 1.5 factor right" (the 1.5: fp32 eigenvalue errors ~2e-5 l_1, entries slightly off PSD)
     * leaves no way for l_1 > fp32(median factor) to fail.  cert_mirror (pinned host, 1 int) = 1 if proven for ALL
     * layers, else 0 (flat spectra, NaN: the caller then waits for the ranks as before); ev_cert is recorded behind it.
     * cert_stream may be chain_stream (the kernel then sits between the Grams and the factorisation; measured best). */
    hipStream_t cert_stream; int* cert_mirror; void* ev_cert;
    double* cert_scratch;                  /* basd_rank_certificate_scratch_bytes(L) bytes, ZEROED once (the kernel leaves it zeroed) */
} BasdSelectorChain;
int basd_selector_chain(const BasdSelectorChain* args);
/* The certificate kernel of BasdSelectorChain.cert_mirror on its own: *flag (device or pinned host memory) = 1 iff
 *     max(|zbar|^2, ||G||_F^2 / tr G) > 1.5 factor min(tr G / c, (tr G - |zbar|^2) / (c - 1)),   c = n - (n-1)/2,
 * holds for every one of the `batch` symmetric n x n matrices G = A + zbar zbar^T (A PSD; zbar (batch, n) nullable = 0)
 * -- a sufficient condition for "every Marchenko-Pastur rank (layer_selector.py:16-19) is >= 1"; else 0. */
int basd_rank_certificate(const float* grams, const float* zbar, int n, int batch, double factor, double* scratch,
                          int* flag, hipStream_t stream);
/* scratch: basd_rank_certificate_scratch_bytes(batch) bytes of device memory, zeroed before the FIRST launch that uses
 * it (partial sums + a ticket; the launch leaves it zeroed again). */
long basd_rank_certificate_scratch_bytes(int batch);
/* Test hook: fills every CU's LDS with `pattern` (a NaN, say): no kernel may depend on what its CU's previous tenant left. */
int basd_debug_fill_lds(unsigned pattern, hipStream_t stream);
/* exact_k != 0: the caller has READ the ranks and every one of them equals kmax (one teacher layer): the principal-angle
 * matrices then have one common order and may leave LDS (orders past basd_jacobi_lds_square_fits; a speculative kmax
 * past it returns BASD_EUNSUPPORTED). */
int basd_selector_chain_tail(const BasdSelectorChain* args, int kmax, int exact_k);
int basd_jacobi_lds_square_fits(int n);

/* ---- attention-weighted Procrustes loss ------------------------------------------------------ */

/* Token weights from the layer-mixed attention.  relational.py:22-34 + layer_selector.py:112.
 * attn_ptrs: device array of L pointers to (B, H, A, A) tensors sharing strides (sb, sh, sq, sk),
 * A = n_a + has_cls.  atap*: weight interpolation n_a -> n_s; tap*: token interpolation n_t -> n_s
 * (both NULL when the grids coincide). */
int basd_token_weights(const void* const* attn_ptrs, int dtype, const float* mix, int L, long sb, long sh, long sq,
                       long sk, int B, int H, int A, int has_cls, int n_a, int n_t, int n_s, const int* atap0,
                       const int* atap1, const float* alam, const int* tap0, const int* tap1, const float* lam,
                       float* omega, float* omega_t, float* raw_out, hipStream_t stream);

/* Student half of relational.py:36-45 and the adjoint of combined.py:9-14:
 * mu = sum_n w_n x_n, tr_s = sum_n w_n |x_n - mu|^2 (as (B, ceil(D/64)) per-slab partials),
 * A' = I_interp^T diag(w) (X - 1 mu^T). */
int basd_student_project(const void* x, int dtype, long sb, long sn, int B, int n_s, int n_t, int D,
                         const float* omega, const int* tap0, const int* tap1, const float* lam, const int* range0,
                         const int* range1, float* mu, float* tr_s, float* a_prime, hipStream_t stream);

/* Teacher half: soft layer mixing (layer_selector.py:110-111), optional resampling to the core grid of
 * n tokens (combined.py:9-14, only when the teacher grid is finer than the student's; g0/g1/glam gather
 * taps, else NULL) and weighted centring (relational.py:37,39).  tok_ptrs: device array of L pointers. */
int basd_teacher_center(const void* const* tok_ptrs, int dtype, const float* mix, int L, long sb, long sn, long sd,
                        int B, int n, int D, const int* g0, const int* g1, const float* glam, const float* omega_t,
                        float* mu, float* tc, hipStream_t stream);
/* The same for G <= 4 groups of mixing weights (mix: (G, L); multi-layer teachers: one group per extraction layer) in
 * ONE pass over the teacher layers: omega_t (G, B, n), mu (G, B, D), tc (G, B, n, D).  Returns BASD_EUNSUPPORTED for
 * more groups or tiles past LDS: call basd_teacher_center per group. */
int basd_teacher_center_multi(const void* const* tok_ptrs, int dtype, const float* mix, int L, int G, long sb, long sn,
                              long sd, int B, int n, int D, const int* g0, const int* g1, const float* glam,
                              const float* omega_t, float* mu, float* tc, hipStream_t stream);
/* Streaming form of the same for row-major teacher tokens and many layers (no LDS tile: full rows are read; the mixed
 * tokens are written uncentred with per-chunk weighted column sums, a second pass folds the sums in a fixed order and
 * subtracts the mean in place).  scratch: basd_teacher_center_stream_scratch_floats() floats.  BASD_EUNSUPPORTED: more
 * than 4 groups or features not contiguous (sd != 1). */
long basd_teacher_center_stream_scratch_floats(int G, int B, int n, int D);
int basd_teacher_center_stream(const void* const* tok_ptrs, int dtype, const float* mix, int L, int G, long sb, long sn,
                               long sd, int B, int n, int D, const int* g0, const int* g1, const float* glam,
                               const float* omega_t, float* mu, float* tc, float* scratch, hipStream_t stream);

/* G[b] = P[b] P[b]^T accumulated in fp64 on v_mfma_f64_16x16x4_f64 (the bmm of relational.py:47,
 * reduced to the teacher grid). */
int basd_gram_f64(const float* p, long p_batch_stride, int n, int D, int batch, double* g, long g_batch_stride,
                  hipStream_t stream);

/* fp64 Cholesky of (possibly singular) PSD matrices; L full row-major. */
/* basd_gram_f64 with the contraction split over `splits` workgroups per matrix (few matrices, long feature axis);
 * slabs: splits * batch * n * n doubles of scratch, folded in a fixed order; g contiguous (batch, n, n). */
int basd_gram_f64_split(const float* p, long p_batch_stride, int n, int D, int batch, int splits, double* slabs,
                        double* g, hipStream_t stream);

int basd_chol_f64(const double* g, long g_batch_stride, int n, int batch, double* l, long l_batch_stride,
                  hipStream_t stream);

/* W[b] = [L_a^T L_b[b % lb_period] ; L_b[b % lb_period]]  (2n x n column-major fp32) -- the input of
 * basd_jacobi_onesided.  lb_period < batch: one teacher factor per sample shared by all extraction layers. */
int basd_stack_product(const double* la, const double* lb, long l_batch_stride, int n, int batch, int lb_period,
                       float* w, long w_batch_stride, hipStream_t stream);

/* Per sample: tr_t, nuclear norm (relational.py:48), loss_b = tr_s + tr_t - 2 nuc (relational.py:50),
 * and K' = Y Sigma^+ Y^T for the backward (nullable). */
int basd_procrustes_finalize(const float* w, long w_batch_stride, const float* sigma, int n, int n_s, int batch,
                             int t_period /* gb, omega indexed by b % t_period */,
                             const double* gb, long g_batch_stride, const float* omega, const int* tap0,
                             const int* tap1, const float* lam, const float* tr_s_part, int tr_slabs, float* tr_s,
                             float* tr_t, float* nuc, float* loss, float* k_prime, hipStream_t stream);

/* dX = (*scale_ptr * scale_const) * w_s * ((x_s - mu) - interp(K' A')[s]): autograd of relational.py:36-50
 * with respect to the student tokens. */
int basd_student_grad(const void* x, int dtype, long sb, long sn, int B, int n_s, int n_t, int D, const float* omega,
                      const float* mu, const float* h, const int* tap0, const int* tap1, const float* lam,
                      const float* scale_ptr, float scale_const, float* dx, const float* tnorm2, float* gomega,
                      hipStream_t stream);

/* basd_student_project / basd_student_grad for all E extraction layers in ONE launch each.  x_ptrs: device table of
 * E base pointers (common strides); omega + e * omega_e_stride (0 = one weight vector for all layers); the other
 * operands are laid out (E, B, ...); scale_ptr[e] = upstream gradient of layer e.
 * basd_student_project_multi returns BASD_EUNSUPPORTED where its vectorised LDS-staged kernel does not apply (rows not
 * 16-byte aligned, slab over 64 KB): fall back to basd_student_project per layer. */
int basd_student_project_multi(const void* const* x_ptrs, int dtype, long sb, long sn, int E, int B, int n_s, int n_t,
                               int D, int ptrs_16B_aligned, const float* omega, long omega_e_stride, const int* tap0,
                               const int* tap1, const float* lam, const int* range0, const int* range1, float* mu,
                               float* tr_s, float* a_prime, hipStream_t stream);
int basd_student_grad_multi(const void* const* x_ptrs, int dtype, long sb, long sn, int E, int B, int n_s, int n_t,
                            int D, const float* omega, long omega_e_stride, const float* mu, const float* h,
                            const int* tap0, const int* tap1, const float* lam, const float* scale_ptr,
                            float scale_const, float* dx, const float* tnorm2, float* gomega, hipStream_t stream);
/* H = K' A' (basd_gemm_tn) and basd_student_grad_multi in ONE launch for cores of n_t <= 64 tokens: each workgroup
 * forms its 64-feature slab of H in LDS (K' through the scalar cache) and streams the sample's token rows once, so H
 * never goes to memory.  k_prime (E, B, n_t, n_t) as basd_procrustes_finalize writes it (symmetric), a_prime
 * (E, B, n_t, D).  Returns BASD_EUNSUPPORTED for larger cores or rows that are not 16-byte aligned: use the two-call
 * form (autograd of relational.py:36-50 either way). */
int basd_student_grad_fused(const void* const* x_ptrs, int dtype, long sb, long sn, int E, int B, int n_s, int n_t,
                            int D, int ptrs_16B_aligned, const float* omega, long omega_e_stride, const float* mu,
                            const float* k_prime, const float* a_prime, const int* tap0, const int* tap1,
                            const float* lam, const float* scale_ptr, float scale_const, float* dx,
                            hipStream_t stream);

/* The whole forward of relational.py:22-50 for E extraction layers against the (mixed) teacher -- and, when `dx` is
 * set, the student-token gradients for the upstream gradients `grad_layers` -- queued by ONE call: the launches of
 * basd_token_weights, basd_teacher_center, basd_student_project(_multi), basd_gram_f64 x2, basd_chol_f64,
 * basd_stack_product, basd_jacobi_onesided, basd_procrustes_finalize [, basd_student_grad_fused or basd_gemm_tn + basd_student_grad_multi] in
 * that order (a dozen FFI calls from Python cost several times the launches themselves, and that host time sat in
 * front of the caller's stream).  All fields are 8 bytes wide; pointers are device memory except student_host_ptrs.
 *   G = 1: the teacher side is shared by all layers (one teacher layer: mixing weights exactly 1), else G = E.
 *   n = min(n_s, n_t): the core grid.  Arrays: omega (G,B,n_s), omega_t (G,B,n), raw (G,B,n_a) nullable,
 *   mu_t (G,B,d_t), tc (G,B,n,d_t), mu_s (E,B,d_s), tr_part (E,B,ceil(d_s/32)), tr_s/tr_t/nuc/loss_b (E,B),
 *   a_prime (E,B,n,d_s), g_all/l_all ((E+G)B,n,n) fp64, W (EB,n,2n), sigma (EB,n),
 *   jflags basd_jacobi_workspace_ints(EB, max_sweeps) ints, sweeps (EB) ints nullable, k_prime (EB,n,n) nullable,
 *   h (EB,n,d_s) + dx (E,B,n_s,d_s) + grad_layers (E): only for the gradients. */
typedef struct BasdProcrustesArgs {
    const void* const* student_ptrs;       /* device table of E pointers */
    const void* const* student_host_ptrs;  /* the same E pointers in host memory (per-layer fallback) */
    long s_dtype, s_sb, s_sn, s_aligned;
    const void* const* tok_ptrs;           /* device table of L pointers */
    long t_dtype, t_sb, t_sn, t_sd;
    const void* const* attn_ptrs;          /* device table of L pointers */
    long a_dtype, a_sb, a_sh, a_sq, a_sk;
    const float* mix;                      /* (G, L) */
    long E, L, G, B, n_s, n_t, d_s, d_t, H, A, has_cls, n_a, n, max_sweeps;
    const int* atap0; const int* atap1; const float* alam;                                  /* n_a -> n_s */
    const int* tap0; const int* tap1; const float* lam; const int* range0; const int* range1;   /* n -> n_s, student side */
    const int* g0; const int* g1; const float* glam;                                        /* teacher gather n_t -> n */
    float* omega; float* omega_t; float* raw; float* mu_t; float* tc; float* mu_s; float* tr_part; float* tr_s;
    float* a_prime;
    double* g_all; double* l_all;
    float* W; float* sigma; int* jflags; int* sweeps;
    float* tr_t; float* nuc; float* loss_b; float* k_prime;
    float* h; float* dx; const float* grad_layers;
    double* g_slabs; long g_splits;     /* nullable / <= 1: teacher Gram unsplit; else g_splits * G*B*n*n doubles of scratch */
    /* UW-SO combination (combined.py:76-85) inside the call, nullable: uw_ce = the base loss (one fp32 on the device);
     * uw_out (4 + 2 E floats) receives w_ce, w_geo, total, geo, then E x w_geo / E, then the E per-layer means.  The
     * student gradients (dx) are then those of `total` for a unit upstream gradient and grad_layers is ignored. */
    const float* uw_ce; float* uw_out;
    /* nullable: basd_jacobi_twopass_workspace_bytes(n, E*B, max_sweeps) bytes; the SVD then takes
     * basd_jacobi_stacked_twopass where that covers the shape */
    void* jac_ws;
    /* nullable, only read when `raw` is set (gradients through the mixing weights): E*B * 2n*n floats; the transposed
     * route is then taken where basd_jacobi_plain4_fits(n) and U Sigma is rebuilt here (basd_ustack_from_transposed);
     * sigma_u: E*B * n floats, the norms of its columns (what the backward pairs with w_stack) */
    float* w_stack; float* sigma_u;
} BasdProcrustesArgs;
int basd_procrustes_forward_fused(const BasdProcrustesArgs* args, hipStream_t stream);
/* Test / tuning hook: 0 = always the stacked cores [M; L_b] (riding rows); 1 = the transposed cores where they apply
 * (default: no gradient through the mixing weights, >= 128 cores, basd_jacobi_plain4_fits(n)); 2 = the same for any
 * batch; negative = keep.  Process-wide. */
int basd_procrustes_tuning(int transposed_cores);

/* The two pieces of the transposed route (relational.py:47-48 and its autograd): M = L_a^T L_b alone, row-major and
 * compact at w + b * w_batch_stride -- as a column-major matrix that is M^T, whose one-sided Jacobi leaves V Sigma in
 * the columns -- and K' = Y Sigma^+ Y^T with Y = L_b V formed from those columns (z: n x n floats of scratch per
 * matrix at z + b * z_batch_stride). */
int basd_stack_product_t(const double* la, const double* lb, long l_batch_stride, int n, int batch, int lb_period,
                         float* w, long w_batch_stride, hipStream_t stream);
int basd_kprime_from_transposed(const float* w, long w_batch_stride, const float* sigma, int n, int batch,
                                const double* lb, long l_batch_stride, int lb_period, float* z, long z_batch_stride,
                                float* k_prime, hipStream_t stream);
/* The transposed route for steps whose backward goes through the mixing weights (multi-layer teachers; autograd of
 * relational.py:47-48 w.r.t. the teacher side reads U Sigma, the top half of the stacked cores: basd_teacher_factor*):
 * basd_ustack_stash, between basd_stack_product_t and the Jacobi, copies M (row-major, compact at wc) into the unused
 * bottom half of the stacked buffer w_stack (E*B x 2n*n floats); basd_ustack_from_transposed, behind the Jacobi, forms
 * M V from it and X = V Sigma (compact at x, what the Jacobi left) in `scratch` (n*n floats per core), makes its columns
 * orthogonal relative to their own norms with a short one-sided Jacobi (orthogonality was enforced on V Sigma: M V has
 * U Sigma's columns to tol sigma_max only; sigma_u receives their norms) and places them in the top half, stacked layout.
 * The backward reads w_stack with sigma_u. */
int basd_ustack_stash(const float* wc, long wc_batch_stride, int n, int batch, float* w_stack, long w_stack_stride,
                      hipStream_t stream);
int basd_ustack_from_transposed(const float* x, long x_batch_stride, const float* sigma, int n, int batch,
                                float* w_stack, long w_stack_stride, float* scratch, long scratch_batch_stride,
                                float* sigma_u, int max_sweeps, int* jflags, hipStream_t stream);

/* x *= num / den unless the ratio is exactly 1 (then the launch returns at once): the upstream gradient of a loss that
 * was differentiated for a unit one.  num, den: one fp32 each on the device, den nullable (= 1). */
int basd_scale_unless_one(float* x, long count, const float* num, const float* den, hipStream_t stream);

/* The base criterion of BASDLoss (combined.py:56) when it is the stock torch.nn.CrossEntropyLoss (mean reduction, no
 * class weights): per-row losses (already divided by the number of rows that are not ignored; loss = their sum) and
 * d loss / d logits in ONE launch.  Exactly one of labels (int64 class indices, B) / probs (fp32 soft targets (B, C),
 * row stride pld); label smoothing as in torch: t' = (1 - eps) t + eps / C. */
int basd_cross_entropy(const void* logits, int dtype, long ld, int B, int C, const long* labels, const float* probs,
                       long pld, float label_smoothing, long ignore_index, float* row_loss, float* dlogits,
                       hipStream_t stream);

/* ---- stream ordering helpers ----------------------------------------------------------------------- */

/* Events to order two streams from inside a library call (basd_tridiag_ranked's mid_event): opaque hipEvent_t handles. */
int basd_event_create(void** out);
int basd_event_destroy(void* event);
int basd_stream_wait_event(hipStream_t stream, void* event);
int basd_event_record(void* event, hipStream_t stream);
int basd_event_synchronize(void* event);      /* blocks the calling host thread */
int basd_event_query(void* event);            /* 1 = reached, 0 = not yet, < 0 = invalid */
/* A stream of priority level -1 (highest of the device's range), 0 (default) or +1 (lowest). */
int basd_stream_create_priority(void** out, int level);
/* diagnostics: events that carry a time stamp (basd_event_create makes them without), and the time between two */
int basd_event_create_timed(void** out);
int basd_event_elapsed_ms(void* from, void* to, float* ms_out);

/* ---- multi-layer teachers only: gradients through the mixing weights and the principal angles ------ */

/* Kt = Q - K'' with d loss_b / d T_c = 2 Kt T_c (autograd of relational.py:36-50 w.r.t. the mixed teacher
 * tokens of layer_selector.py:111), and |t_hat_c[s]|^2 per student token. */
int basd_teacher_factor(const float* w, long w_batch_stride, const float* sigma, int n, int n_s, int batch,
                        const double* la, const double* gb, long g_batch_stride, const float* omega,
                        const int* tap0, const int* tap1, const float* lam, const int* range0, const int* range1,
                        float* kt, float* tnorm2, hipStream_t stream);
/* The same for cores past LDS (basd_teacher_factor returns BASD_EUNSUPPORTED for n > 199): both contractions tiled,
 * Z = L_a U Sigma^-1/2 through z_scratch, (batch, n, n) floats of device scratch. */
int basd_teacher_factor_tiled(const float* w, long w_batch_stride, const float* sigma, int n, int n_s, int batch,
                              const double* la, const double* gb, long g_batch_stride, const float* omega,
                              const int* tap0, const int* tap1, const float* lam, const int* range0, const int* range1,
                              float* kt, float* tnorm2, float* z_scratch, hipStream_t stream);

/* partial[e][b][l] = <R[e][b], teacher layer l on the core grid>: d loss / d mix_l through the tokens
 * (layer_selector.py:111). */
int basd_mix_grad_tokens(const float* r, const void* const* tok_ptrs, int dtype, int L, long sb, long sn, long sd,
                         int E, int B, int n, int D, const int* g0, const int* g1, const float* glam, float* partial,
                         hipStream_t stream);

/* The same in ONE pass over the teacher layers for all E <= 4 extraction layers (BASD_EUNSUPPORTED beyond).  scratch:
 * basd_mix_grad_tokens_scratch_floats() floats of device memory; per-chunk sums are folded in a fixed order. */
long basd_mix_grad_tokens_scratch_floats(int E, int B, int L, int n);
int basd_mix_grad_tokens_onepass(const float* r, const void* const* tok_ptrs, int dtype, int L, long sb, long sn, long sd,
                                 int E, int B, int n, int D, const int* g0, const int* g1, const float* glam,
                                 float* partial, float* scratch, hipStream_t stream);

/* partial[e][b][l] = d loss / d mix_l through the attention-derived token weights
 * (layer_selector.py:112 + relational.py:22-34). */
int basd_token_weight_bwd(const float* gomega, const float* raw, int E, int B, int n_a, int n_s, const int* atap0,
                          const int* atap1, const float* alam, const int* arange0, const int* arange1,
                          const void* const* attn_ptrs, int dtype, int L, long sb, long sh, long sq, long sk, int H,
                          int A, int has_cls, float* partial,
                          float* graw_out /* nullable: (E, B, n_a) d loss_b / d raw attention-grid weights */,
                          hipStream_t stream);

/* [masked cosine matrix ; identity] stacks (2 kmax x kmax, column-major) for the principal-angle SVD with
 * right singular vectors (backward of layer_selector.py:99). */
int basd_build_angle_stack(const float* cos, int kmax, const int* k_arr, int items, float* out, hipStream_t stream);

/* gWt[item][j][i] = d (d_grass_sq[item]) / d W[i][j] * gd[item]  (backward of layer_selector.py:99-105). */
int basd_grassmann_distance_bwd(const float* stack, const float* colnorm, int kmax, const int* k_arr, const float* sw,
                                int sw_stride, const int* sw_index, const float* gd, int items, float* gwt,
                                hipStream_t stream);

/* K2[i][j] = (M[i][j] - M[j][i]) / (lam_j - lam_i): eigenvector perturbation of the student Gram
 * (backward of torch.linalg.svd at layer_selector.py:92). */
int basd_eigvec_k2(const float* m, const float* lam, int D, int kmax, int batch, float* k2, hipStream_t stream);

/* Stand-alone `_align_token_count` (combined.py:9-14): out (B, n_out, D) fp32 contiguous; and its adjoint. */
int basd_resample_tokens(const void* x, int dtype, long sb, long sn, long sd, int B, int n_in, int n_out, int D,
                         const int* tap0, const int* tap1, const float* lam, float* out, hipStream_t stream);
int basd_resample_tokens_adjoint(const float* dy, int B, int n_in, int n_out, int D, const int* tap0,
                                 const int* tap1, const float* lam, const int* range0, const int* range1, float* dx,
                                 hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BASD_HIP_H */
