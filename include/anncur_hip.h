/* anncur_hip.h -- C ABI of libanncur_hip.so: the MI355X (gfx950) kernels behind the
 * CUR nearest-neighbour path of iesl/anncur.
 *
 * The reference is pure Python with no FFI of its own; the symbols below are what a
 * binding for this path binds (ctypes).  Each entry point cites
 * the reference interface (file:line, relative to the upstream repo) it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error (ANNCUR_E_*); the message is
 *    available from anncur_last_error() (thread-local);
 *  - the caller owns every buffer; pointers are DEVICE pointers unless stated;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *    asynchronous on that stream; no hidden allocation, no host synchronisation:
 *    scratch comes from an explicit workspace sized by the *_workspace_bytes() query;
 *  - matrices are row-major; `ld*` are leading dimensions in ELEMENTS;
 *  - dtype codes: ANNCUR_F32 (float) / ANNCUR_BF16 (bfloat16, 2 bytes);
 *  - top-k results are sorted by score descending, ties broken by the smaller index
 *    (torch.topk leaves tie order unspecified; this is one valid order);
 *  - NaN scores are never selected.
 */
#ifndef ANNCUR_HIP_H
#define ANNCUR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANNCUR_F32  0
#define ANNCUR_BF16 1
#define ANNCUR_F64  2            /* only the fp64 helpers of the on-device pseudo-inverse take it */

#define ANNCUR_OK            0
#define ANNCUR_E_INVALID    -1   /* bad argument (shape, dtype, alignment, k range) */
#define ANNCUR_E_WORKSPACE  -2   /* workspace missing or too small */
#define ANNCUR_E_HIP        -3   /* a HIP runtime call / kernel launch failed */
#define ANNCUR_E_UNSUPPORTED -4  /* valid request outside what this build implements */

#define ANNCUR_MAX_TOPK 2048     /* largest k accepted by the top-k entry points */

/* library / diagnostics ------------------------------------------------------------ */
int         anncur_version(void);            /* 1000*major + minor */
const char *anncur_last_error(void);         /* thread-local, never NULL */
int         anncur_device_info(int *n_cu, int *wave_size, char *arch_name, int arch_name_len);

/* a2: anchor row / column gather ---------------------------------------------------
 * rows = A[row_idxs,:], cols = A[:,col_idxs]
 *   eval/run_retrieval_eval_wrt_exact_crossenc.py:73-74
 *   eval/run_retrieval_eval_wrt_exact_crossenc_w_fixed_train_test_splits.py:297,300
 * src and dst share `dtype` unless dst_dtype says otherwise (F32<->BF16 convert on the fly).
 * idx are int32 device arrays; out-of-range indices are an error reported lazily as
 * zeros (the Python layer validates indices on the host before calling). */
int anncur_gather_cols(const void *A, int dtype, int64_t n_rows, int64_t n_cols, int64_t lda,
                       const int32_t *col_idx, int32_t n_idx,
                       void *out, int dst_dtype, int64_t ldo, void *stream);
int anncur_gather_rows(const void *A, int dtype, int64_t n_rows, int64_t n_cols, int64_t lda,
                       const int32_t *row_idx, int32_t n_idx,
                       void *out, int dst_dtype, int64_t ldo, void *stream);

/* a4/a5/a6: strided GEMM with exact-fp32 accumulation ------------------------------
 * C(m,n) = sum_k A(m,k) * B(k,n), element (i,j) of an operand at base + i*s0 + j*s1
 * (strides in elements), so NN / NT / TN / transposed-output products are one call.
 * Inputs F32 or BF16 (up-converted exactly), products and sums in fp32 on the matrix
 * cores (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain).  Replaces
 *   latent_cols = U @ R                      eval/matrix_approx_zeshel.py:65
 *   latent_rows = C @ U                      eval/matrix_approx_zeshel.py:61
 *   get / get_rows / get_cols / get_complete_row / get_complete_col
 *                                             eval/matrix_approx_zeshel.py:74,79,85,97,118
 *   pinv(C) @ A @ pinv(R)                    eval/matrix_approx_zeshel.py:47 (the two products)
 *   mention_embeds @ label_embeds.T          ..._w_fixed_train_test_splits.py:283 */
int anncur_gemm(const void *A, int a_dtype, int64_t a_sm, int64_t a_sk,
                const void *B, int b_dtype, int64_t b_sk, int64_t b_sn,
                void *C, int c_dtype, int64_t c_sm, int64_t c_sn,
                int64_t M, int64_t N, int64_t K, void *stream);

/* Same product with scaling and accumulation: C = alpha * A.B + beta * Cin (Cin fp32 with its own strides, may be NULL; it may
 * alias C element for element).  Building block of the on-device pseudo-inverse (Newton-Schulz X <- 2X - (X W) X), the
 * alternative to the host numpy.linalg.pinv of eval/matrix_approx_zeshel.py:47,49 for large anchor counts. */
int anncur_gemm_ex(const void *A, int a_dtype, int64_t a_sm, int64_t a_sk,
                   const void *B, int b_dtype, int64_t b_sk, int64_t b_sn,
                   void *C, int c_dtype, int64_t c_sm, int64_t c_sn,
                   int64_t M, int64_t N, int64_t K, float alpha, float beta,
                   const float *Cin, int64_t i_sm, int64_t i_sn, void *stream);
/* out[0] = sum of squares of an fp32 matrix (Frobenius norm squared); dst(i,j) = alpha / (divide_by ? divide_by[0] : 1) * src(i,j)
 * with arbitrary strides (scaled transpose).  Helpers of the on-device pseudo-inverse; no host synchronisation. */
int anncur_sumsq(const float *A, int64_t n_rows, int64_t n_cols, int64_t lda, float *out, void *stream);
int anncur_scale_copy(const float *src, int64_t s0, int64_t s1, float *dst, int64_t d0, int64_t d1, int64_t M, int64_t N,
                      float alpha, const float *divide_by, void *stream);

/* fp64 building blocks of the PARITY-GRADE on-device pseudo-inverse (anncur_amd/pinv.py, Newton-Schulz X <- X (2 I - W X) in
 * double precision, rounded to fp32 once at the end): replaces the host call
 *   U = numpy.linalg.pinv(W)                 eval/matrix_approx_zeshel.py:47,49
 * when the caller asks for it (pinv_backend "device" / "auto").  Strided like anncur_gemm_ex; products and sums in fp64 on
 * v_mfma_f64_16x16x4_f64.  anncur_convert_f64: dst(i,j) = alpha / (divide_by ? divide_by[0] : 1) * src(i,j) for the dtype pairs
 * F32->F64, BF16->F64, F64->F64 (scaled copy / transpose) and F64->F32 (one rounding).  anncur_diff_sumsq_f64:
 * out2 = {sum (x - y)^2, sum x^2} over n contiguous doubles (y may be NULL), device doubles, no host synchronisation. */
int anncur_gemm_f64(const double *A, int64_t a_sm, int64_t a_sk, const double *B, int64_t b_sk, int64_t b_sn,
                    double *C, int64_t c_sm, int64_t c_sn, int64_t M, int64_t N, int64_t K,
                    double alpha, double beta, const double *Cin, int64_t i_sm, int64_t i_sn, void *stream);
int anncur_convert_f64(const void *src, int src_dtype, int64_t s0, int64_t s1, void *dst, int dst_dtype, int64_t d0, int64_t d1,
                       int64_t M, int64_t N, double alpha, const double *divide_by, void *stream);
int anncur_diff_sumsq_f64(const double *X, const double *Y, int64_t n, double *out2, void *stream);

/* a11: approximation error without materialising S_hat ------------------------------
 * err_sq[q] = sum_i (X[q,:].E[:,i] - A[q,i])^2 , norm_sq[q] = sum_i A[q,i]^2
 *   eval/run_retrieval_eval_wrt_exact_crossenc.py:146-147 (torch.norm over row subsets;
 *   the host sums the per-row terms over anchor / non-anchor / all rows in fp64).
 * X [Q x K] (x_dtype), Et [I x K] = E transposed (e_dtype), A [Q x I] (a_dtype).
 * err_sq / norm_sq: float[Q], overwritten. */
int anncur_approx_error(const void *X, int x_dtype, int64_t ldx,
                        const void *Et, int e_dtype, int64_t lde,
                        const void *A, int a_dtype, int64_t lda,
                        int64_t Q, int64_t I, int64_t K,
                        float *err_sq, float *norm_sq, void *stream);

/* The same on the fused path's packed bf16 operands (layouts of anncur_score_topk: X [Q x Kp], Et [ceil32(I) x Kp], Kp in
 * {64,128,256,512}), on the bf16 MFMA loop of the sweep instead of the strided fp32 GEMM: ~25x faster at Q=10k, I=100k, Kp=256.
 * A: fp32 or bf16, 16-byte aligned, lda a multiple of 4 (ANNCUR_E_UNSUPPORTED otherwise: use anncur_approx_error). */
int anncur_approx_error_packed(const void *X, int64_t ldx, const void *Et, int64_t lde,
                               const void *A, int a_dtype, int64_t lda,
                               int64_t Q, int64_t I, int32_t Kp,
                               float *err_sq, float *norm_sq, void *stream);

/* a8 (retrieval half) + a11 of one grid cell of entry point A in ONE sweep (SURVEY 8b.6 `anncur_eval_fused`) ---------------------------
 *   approx = CURApprox(...).get(rows, cols); approx.topk(top_k_retvr); torch.norm((approx - A)[rows]) ; torch.norm(A[rows])
 *                                                                     eval/run_retrieval_eval_wrt_exact_crossenc.py:84,106,146-147
 * = anncur_score_topk(X, Et, k) + anncur_approx_error_packed(X, Et, A): out_val / out_idx = the exact top-k of S_hat = X . E (values
 * and index sets those of anncur_score_topk_ex with ANNCUR_TOPK_MFMA32: same MFMAs in the same order), err_sq[q] = sum_i (S_hat - A)^2,
 * norm_sq[q] = sum_i A^2 -- with ONE S_hat GEMM per sweep stage instead of two: the kernel that streams the exact tile through LDS
 * beside the MFMA chain also runs the threshold filter on the accumulator it holds (csrc/score_evalf.hpp).
 * X [Q x Kp], Et [ceil32(I) x Kp] packed bf16 in ITEM order (a tile of Et faces the same 32 columns of A), Kp in {64,128,256};
 * A [Q x I] bf16, 16-byte aligned, lda a multiple of 8 (ANNCUR_E_UNSUPPORTED otherwise: the two calls above).  Workspace:
 * anncur_eval_fused_workspace_bytes (0: shape outside the fused path). */
size_t anncur_eval_fused_workspace_bytes(int64_t Q, int64_t I, int32_t Kp, int32_t k);
int anncur_eval_fused(const void *X, int64_t ldx, const void *Et, int64_t lde, const void *A, int a_dtype, int64_t lda,
                      int64_t Q, int64_t I, int32_t Kp, int32_t k, float *out_val, int32_t *out_idx, float *err_sq, float *norm_sq,
                      void *workspace, size_t workspace_bytes, void *stream);
/* The same with a HINT for the first threshold (round 5): Et_hint = a copy of Et's rows in any other order, same shape and pitch -- CURApprox
 * keeps one in descending-norm order for anncur_score_topk_ex (ANNCUR_TOPK_LEADING_SAMPLE) -- whose LEADING tiles the prepass samples instead of a
 * strided sample of Et: with the largest-norm items in the sample the first threshold is tighter and the sweep keeps fewer candidates.  The
 * sweep itself runs on Et (item order).  Same results bit for bit (any subset of the items yields a valid threshold); null = anncur_eval_fused. */
int anncur_eval_fused_ex(const void *X, int64_t ldx, const void *Et, int64_t lde, const void *Et_hint, const void *A, int a_dtype, int64_t lda,
                         int64_t Q, int64_t I, int32_t Kp, int32_t k, float *out_val, int32_t *out_idx, float *err_sq, float *norm_sq,
                         void *workspace, size_t workspace_bytes, void *stream);

/* a7/a8: exact row-wise top-k of a stored matrix (HBM-streaming scan) ---------------
 *   torch.topk(S, k, dim=1)                  eval/matrix_approx_zeshel.py:106,126
 *   curr_ment_scores.topk(top_k)             ...crossenc.py:103 ; ..._splits.py:86
 * A [Q x I] (dtype), out_val float[Q x k] (ldo = k), out_idx int32[Q x k].  1 <= k <= min(I, ANNCUR_MAX_TOPK).
 * Order: score descending, equal scores by ascending index (torch leaves ties unspecified).  -inf entries are ordinary
 * candidates; NaN is never selected (torch ranks NaN first): a row with fewer than k non-NaN entries is padded with (-inf, -1). */
int anncur_rowwise_topk(const void *A, int dtype, int64_t Q, int64_t I, int64_t lda, int32_t k,
                        float *out_val, int32_t *out_idx, void *stream);

/* The same scan over RAGGED rows (round 5; the batched IVF search's packed score rows): row q of A holds row_len[q] elements,
 * k <= row_len[q] <= I_max <= lda (a length outside 0..I_max is clamped; a row shorter than k is padded with (-inf, -1)).  k <= 128 (ANNCUR_E_UNSUPPORTED above). */
int anncur_rowwise_topk_ragged(const void *A, int dtype, int64_t Q, int64_t I_max, int64_t lda, const int32_t *row_len, int32_t k,
                               float *out_val, int32_t *out_idx, void *stream);

/* a2 folded into a8's pass over A (SURVEY a2 "fold into the first pass"; reference ..._splits.py:297,300 + 86): the scan of
 * anncur_rowwise_topk (k <= 128) also copies the anchor columns, cq[q, j] = A[q, col_idx[j]], as their 16-byte vectors stream past --
 * C_q costs no second read of the rows' sectors.  col_idx: n_idx (<= 65535) ascending, distinct columns in [0, I);
 * vec_tab uint32[ceil(I / V)] (V = 8 elements for bf16, 4 for fp32; per 16-byte vector: which elements are anchors, and how many
 * anchors lie before it) from anncur_gather_tables for the same (col_idx, I, dtype) -- built once per anchor set; cq [Q x ldo] has A's
 * element type.  A and its rows must be 16-byte aligned
 * (ANNCUR_E_UNSUPPORTED otherwise: use anncur_gather_cols + anncur_rowwise_topk).  Same top-k result as anncur_rowwise_topk.
 * Measured on MI355X (cfg2: 10 000 x 100 000 bf16, 256 anchors): 0.50 ms against 0.357 + 0.049 ms for the two separate kernels -- one HBM
 * pass less, but the per-vector extraction costs more than the sectors it saves; callers that are HBM-capacity- rather than
 * time-bound may still prefer it. */
int anncur_gather_tables(const int32_t *col_idx, int32_t n_idx, int64_t I, int dtype, uint32_t *vec_tab, void *stream);
int anncur_rowwise_topk_gather(const void *A, int dtype, int64_t Q, int64_t I, int64_t lda, int32_t k, float *out_val, int32_t *out_idx,
                               const int32_t *col_idx, int32_t n_idx, const uint32_t *vec_tab, void *cq, int64_t ldo, void *stream);

/* a6+a7 fused: S_hat = X . E never written; per-query top-k ---------------------------
 *   CURApprox.topk_in_row                    eval/matrix_approx_zeshel.py:121-126
 *   approx_curr_ment_scores.topk(top_k_retvr) ...crossenc.py:106 ; ..._splits.py:89
 *   faiss IndexFlatIP.add / .search          models/nearest_nbr.py:36-38 (+ utils/data_process.py:351,397)
 * X  [Q x Kp]  bf16 row-major (queries' anchor-item scores / query embeddings),
 * Et [Ip x Kp] bf16 row-major (item embeddings E^T), Kp in {64,128,256,512} with the
 * logical K zero-padded up to Kp, Ip = I rounded up to a multiple of 32 with zero rows.
 * Products are exact (bf16 x bf16 in fp32), sums fp32 on v_mfma_f32_32x32x16_bf16.
 * Launches on `stream`: (1) group-max pre-pass over a strided sample of item tiles ->
 * (2) per-query lower bound tau on the k-th best; (3) full sweep in 1-3 stages, survivors
 * >= tau appended to per-lane candidate segments, tau raised between stages; (4) per-query
 * select + sort (a query whose segments overflowed is repaired in the same call by
 * recomputing the affected item range: the result is always the exact top-k of S_hat).
 * out_val float[Q x k], out_idx int32[Q x k]. */
size_t anncur_score_topk_workspace_bytes(int64_t Q, int64_t I, int32_t Kp, int32_t k);
int anncur_score_topk_supported(int64_t Q, int64_t I, int32_t Kp, int32_t k);   /* 1 / 0 */
int anncur_score_topk(const void *X, int64_t ldx, const void *Et, int64_t lde,
                      int64_t Q, int64_t I, int32_t Kp, int32_t k,
                      float *out_val, int32_t *out_idx,
                      void *workspace, size_t workspace_bytes, void *stream);

/* The same with two hints an index builder can give (both leave the result the exact top-k of S_hat):
 *  ANNCUR_TOPK_LEADING_SAMPLE: the rows of Et are ordered so that the likeliest high scorers come first (e.g. by descending
 *    norm); the threshold sample then takes the leading item tiles instead of a strided sample (fewer survivors in the sweep);
 *  item_ids (int32[I], may be NULL): out_idx reports item_ids[row of Et] instead of the row (undoes such a reordering; score
 *    ties are then ordered by row, not by id).  Same workspace as anncur_score_topk. */
#define ANNCUR_TOPK_LEADING_SAMPLE 1
/*  The sweep has two bodies for Kp <= 256: v_mfma_f32_32x32x16_bf16 with per-lane candidate rings, and v_mfma_f32_16x16x32_bf16 with one
 *    candidate queue per wave (I < 2^26; the chip holds a higher clock on that shape).  Same products and fp32 sums: the result is the
 *    same bit for bit up to the order of exact score ties.  Default: the 16x16x32 body for k <= 384, the 32x32x16 body above.
 *  ANNCUR_TOPK_MFMA16 / ANNCUR_TOPK_MFMA32: force the 16x16x32 / the 32x32x16 body (A/B variants).  Kp = 512 has one MFMA shape and two
 *    candidate paths: one queue per wave with the dynamic tile schedule on 16x16x32 MFMAs (default, I < 2^26), per-lane rings with
 *    static shares on 32x32x16 MFMAs (ANNCUR_TOPK_MFMA32). */
#define ANNCUR_TOPK_MFMA16 2
/*  ANNCUR_TOPK_QT1 (Kp = 128 / 256): one 32-query sub-tile per wave and three workgroups per CU, with the cross-tile software
 *    pipeline of the Kp = 512 sweep, instead of two sub-tiles staggered inside a wave at two workgroups per CU (A/B variant). */
#define ANNCUR_TOPK_QT1 4
#define ANNCUR_TOPK_MFMA32 8
/*  ANNCUR_TOPK_RING (Kp = 128 / 256, k <= 128): the 16x16x32 body in 8-wave workgroups of 512 queries whose item tiles stream through a
 *    ring of four LDS slots synchronised by per-wave landed / done counters in LDS instead of a workgroup barrier per tile (round 4,
 *    csrc/score16r.hpp: half the L2 -> LDS traffic, half the DMA pieces per wave).  Same result bit for bit; measured slower than the
 *    default (4-wave workgroups, two tile buffers, one barrier per tile) at every size tried -- an A/B variant, not the default.
 *    Callers that set it MUST read workspace word 0 (uint32) after the call has completed: every spin of that body is bounded, and a
 *    wave whose spin ran out adds 2^30 to that word and stops waiting -- the call still returns ANNCUR_OK, its result is invalid
 *    (anncur_amd/ops.py::score_topk_fused(ring=True) raises on it). */
#define ANNCUR_TOPK_RING 16
/*  ANNCUR_TOPK_STAGED: the default 16x16x32 body (Kp <= 256, k <= 384) WITHOUT its threshold ladder -- the sweep in stages with a refinement
 *    launch between them, as rounds 1-4 ran it (A/B and parity reference; the default since round 5 is ONE sweep launch whose waves move
 *    their thresholds up a ladder of levels from device-wide counts of the candidates kept so far: csrc/score16.hpp).  Same result. */
#define ANNCUR_TOPK_STAGED 32
int anncur_score_topk_ex(const void *X, int64_t ldx, const void *Et, int64_t lde,
                         int64_t Q, int64_t I, int32_t Kp, int32_t k,
                         float *out_val, int32_t *out_idx,
                         void *workspace, size_t workspace_bytes,
                         int32_t flags, const int32_t *item_ids, void *stream);

/* a8 per-query evaluation loop: exact top-k of the stored scores AND the approximate retrieval, in one call ----------------------
 *   curr_ment_scores.topk(top_k) ; approx_curr_ment_scores.topk(top_k_retvr)          ...crossenc.py:97-106 ; ..._splits.py:80-89
 * = anncur_rowwise_topk(A, k_exact) + anncur_score_topk_ex(X, Et, k_retvr), same results, scheduled together: the retrieval is a chain
 * of MFMA-bound sweep launches with latency-bound launches between them (threshold, refinements, select); the HBM-bound exact scan is
 * cut into row chunks (whole rounds of the scan's rows in flight) and chunk i runs on `aux_stream` beside the i-th latency-bound
 * launch -- forked and joined with events, so the call is one unit of work on `stream` (and capturable into a graph).  aux_stream
 * NULL or equal to stream: the two parts run one after the other.  Workspace as anncur_score_topk.
 * On an error return the outputs are undefined (chunks of the scan already forked are joined back into `stream` first). */
int anncur_eval_topk(const void *A, int a_dtype, int64_t lda, int32_t k_exact, float *exact_val, int32_t *exact_idx,
                     const void *X, int64_t ldx, const void *Et, int64_t lde, int64_t Q, int64_t I, int32_t Kp, int32_t k_retvr,
                     float *approx_val, int32_t *approx_idx, void *workspace, size_t workspace_bytes,
                     int32_t flags, const int32_t *item_ids, void *stream, void *aux_stream);

/* Measurement only: same as anncur_score_topk but records HIP events on `stream` between the four
 * launches, synchronises, and returns their durations in stage_ms[9] (host floats, milliseconds):
 * {prepass, threshold, sweep stage (sweep launches + the threshold refinements between them), select,
 *  sum of the sweep-kernel launches alone, number of sweep launches, and the duration of each of the up to
 *  three sweep launches (0 where the plan has fewer stages; anncur_score_topk_plan_ex gives their tile ranges)}.
 *  bench.py's live roofline figure is (2*Q*Kp*I / launches) / (stage_ms[4] / launches). */
int anncur_score_topk_timed(const void *X, int64_t ldx, const void *Et, int64_t lde,
                            int64_t Q, int64_t I, int32_t Kp, int32_t k,
                            float *out_val, int32_t *out_idx,
                            void *workspace, size_t workspace_bytes,
                            int32_t flags, const int32_t *item_ids, void *stream, float *stage_ms);
/* Plan introspection: out5 = {sample tiles, item tiles, item splits S, segment capacity, group size}. */
int anncur_score_topk_plan(int64_t Q, int64_t I, int32_t Kp, int32_t k, int32_t *out5);
/* The plan a call with `flags` (ANNCUR_TOPK_*) would run: out[0 .. n_out), n_out <= 17 = {sample tiles, item tiles, item splits S,
 * segment capacity, group size, candidate segments per (query, item split) -- 2: 32x32x16 sweep, 1: 16x16x32 sweep, 4: wide
 * kernel --, 32-query sub-tiles per wave, number of sweep stages, stage_end[3] (tiles), body per stage[3] (0: 32x32x16 with the ballot
 * filter, 1: 32x32x16 with the exec-mask filter, 2: 16x16x32, 3 / 4: the Kp = 512 body with the wave-level queue on 32x32x16 / 16x16x32 MFMAs), ring
 * drain period per stage[3]}.  Lets a caller (and the parity tests) see that a variant flag was honoured for the shape. */
int anncur_score_topk_plan_ex(int64_t Q, int64_t I, int32_t Kp, int32_t k, int32_t flags, int32_t *out, int32_t n_out);
/* Diagnostics: mean number of candidates per query the sweep of the last call on `workspace` kept (reads the segment counts it left
 * behind; synchronises `stream`).  k ln(I / k) is what a sequential threshold can reach. */
int anncur_score_topk_survivors(const void *workspace, int64_t Q, int64_t I, int32_t Kp, int32_t k, int32_t flags, double *mean_per_query,
                                void *stream);

/* a8: exact re-rank of the approximately retrieved items + a10 overlap counts --------
 *   temp[approx_idx] = exact[approx_idx]; temp.topk(k)      ...crossenc.py:108-113 ; ..._splits.py:93-96
 *   compute_overlap(exact[:, :top_k], rerank[:, :top_k])    eval/eval_utils.py:115-150
 * A [Q x I] exact scores; approx_idx int32, row q at approx_idx + q*ld_idx, first k_retvr entries used
 * (distinct per row; ld_idx lets a prefix of a longer sorted list be re-ranked in place);
 * rerank_val float[Q x k_out], rerank_idx int32[Q x k_out]: the k_out best retrieved items
 * by exact score (k_out <= k_retvr <= ANNCUR_MAX_TOPK). */
int anncur_rerank(const void *A, int dtype, int64_t Q, int64_t I, int64_t lda,
                  const int32_t *approx_idx, int64_t ld_idx, int32_t k_retvr, int32_t k_out,
                  float *rerank_val, int32_t *rerank_idx, void *stream);
/* common[p*Q + q] = | a[q, :ka[p]] (set) intersect b[q, :kb[p]] | for each of n_pairs
 * (ka,kb) prefix-length pairs (host int32 arrays), a int32[Q x la], b int32[Q x lb]. */
int anncur_overlap_counts(const int32_t *a, int32_t la, const int32_t *b, int32_t lb, int64_t Q,
                          const int32_t *ka, const int32_t *kb, int32_t n_pairs,
                          int32_t *common, void *stream);

/* f3: IVF-flat inner-product index (the branch of build_flat_or_ivff_index above 11 000 vectors) -------------------------------
 *   faiss.IndexIVFFlat(IndexFlatIP(d), d, nlist, METRIC_INNER_PRODUCT).train / .add / .search     models/nearest_nbr.py:40-52
 * FAISS is not vendored nor pinned by the reference (parity unpinned): restated from the published algorithm, judged on recall
 * against the exact search.  The dense steps (points x centroids, queries x centroids -> top-nprobe) are anncur_gemm +
 * anncur_rowwise_topk; these three hold what is particular to the inverted file.
 *  anncur_ivf_build_lists: assign int32[n] (list of every point) -> counts int32[nlist], offsets int32[nlist+1] (exclusive prefix),
 *    ids int32[n] (points of list l at ids[offsets[l] .. offsets[l+1]) in ascending id order; deterministic, no atomics);
 *  anncur_ivf_list_means: centroids[l] = mean of rows offsets[l]..offsets[l+1] of Xs (vectors stored in list order), summed in
 *    list order; an empty list keeps its centroid (k-means update step);
 *  anncur_renorm_rows: every row of the fp32 matrix M rescaled to unit L2 norm in place (a zero row stays) -- FAISS' fvec_renorm_L2:
 *    IndexIVF trains its coarse quantiser with cp.spherical = true for METRIC_INNER_PRODUCT, i.e. the centroids are renormalised after
 *    every k-means update (faiss/IndexIVF.cpp Level1Quantizer::train_q1; the reference reaches it through models/nearest_nbr.py:46-49);
 *  anncur_ivf_scan: per query, exact inner products with every vector of its nprobe lists (probe int32[nq x nprobe], -1 = skip)
 *    and the k best: out_val float[nq x k] descending, out_idx int32[nq x k] (ids of the points; (-inf, -1) where the probed lists
 *    hold fewer than k vectors).  Xs / Q rows zero-padded to dp floats, dp a multiple of 16, 16-byte aligned. */
int anncur_ivf_build_lists(const int32_t *assign, int64_t n, int32_t nlist, int32_t *counts, int32_t *offsets, int32_t *ids, void *stream);
int anncur_ivf_list_means(const float *Xs, int64_t ldx, int32_t d, const int32_t *offsets, int32_t nlist, float *centroids, int64_t ldc, void *stream);
int anncur_renorm_rows(float *M, int64_t n_rows, int64_t n_cols, int64_t ld, void *stream);
int anncur_ivf_scan(const float *Xs, int64_t ldx, int32_t dp, const int32_t *offsets, const int32_t *ids, const float *Q, int64_t ldq, int64_t nq,
                    const int32_t *probe, int32_t nprobe, int32_t k, float *out_val, int32_t *out_idx, void *stream);

/* Batched IVF search (many queries, e.g. the reference's hard-negative mining, utils/data_process.py:343-365, where every mention
 * queries the index): the (query, probe slot) pairs are sorted by list -- anncur_ivf_build_lists on the flattened probe array gives
 * pair_offsets[nlist+1] and pair_ids (pair = q * nprobe + slot) -- and every list is scored against its pairs' queries as a small GEMM on
 * the fp32 matrix cores, so a list's vectors are read once per 64 queries instead of once per query.  tiles int32[n_tiles x 3] =
 * (list, 64-pair tile, 64-vector tile) worklist built by the host from the list / pair counts.  S float[nq * nprobe x lmax] (lmax >=
 * longest list), pre-filled with -inf by the caller: S[pair][position in its list] = <query, vector>.  Then anncur_rowwise_topk over
 * S viewed as [nq x nprobe * lmax] and anncur_ivf_map_ids (column -> id of the vector; -1 where the score is the -inf padding). */
int anncur_ivf_group_scores(const float *Xs, int64_t ldx, int32_t dp, const int32_t *offsets, const float *Q, int64_t ldq, int32_t nprobe,
                            const int32_t *pair_ids, const int32_t *pair_offsets, const int32_t *tiles, int32_t n_tiles, int64_t lmax, float *S,
                            void *stream);
/* The same on bf16 operands (Xs / Q: bf16 rows zero-padded to dp elements, dp a multiple of 16, 16-byte aligned; an index built with
 * dtype = "bf16" keeps such a copy of its lists): v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 scores. */
int anncur_ivf_group_scores_bf16(const void *Xs, int64_t ldx, int32_t dp, const int32_t *offsets, const void *Q, int64_t ldq, int32_t nprobe,
                                 const int32_t *pair_ids, const int32_t *pair_offsets, const int32_t *tiles, int32_t n_tiles, int64_t lmax, float *S,
                                 void *stream);
/* The same with the tile worklist built ON THE DEVICE (no host look at the pairs-per-list counts, no synchronisation inside a search):
 * tile_start int32[nlist + 1] is filled here (tile_start[l] = tiles of the lists before l, a tile = 64 pairs x 64 vectors), the GEMM is
 * launched with max_tiles workgroups -- any upper bound on sum_l ceil(pairs_l / 64) ceil(size_l / 64), e.g.
 * (n_pairs / 64) * max_l ceil(size_l / 64) + sum_l ceil(size_l / 64) -- and workgroup b finds its (list, pair tile, vector tile) by binary
 * search; workgroups past the last tile exit.  dtype ANNCUR_F32 (Xs / Q fp32) or ANNCUR_BF16 (bf16 rows as anncur_ivf_group_scores_bf16). */
int anncur_ivf_group_scores_dev(const void *Xs, int dtype, int64_t ldx, int32_t dp, const int32_t *offsets, int32_t nlist, const void *Q, int64_t ldq,
                                int32_t nprobe, const int32_t *pair_ids, const int32_t *pair_offsets, int32_t *tile_start, int32_t max_tiles, int64_t lmax,
                                float *S, void *stream);
int anncur_ivf_map_ids(const int32_t *col, const float *val, int64_t nq, int32_t k, int64_t lmax, const int32_t *probe, int32_t nprobe,
                       const int32_t *offsets, const int32_t *ids, int32_t *out_idx, void *stream);

/* The batched search as ONE stream-ordered call (round 5) -- faiss.IndexIVFFlat.search for many queries, models/nearest_nbr.py:50-52 as
 * the reference's hard-negative mining drives it (utils/data_process.py:343-365).  From the probed lists (probe int32[nq x nprobe], entries
 * outside 0..nlist-1 skipped) to the k best (out_val float[nq x k] descending, out_idx int32[nq x k] ids; (-inf, -1) where the probed lists
 * hold fewer than k vectors): pairs grouped by list on the device, one tile GEMM launch on the matrix cores (bf16 rows with dp a multiple
 * of 128: 128 x 128 tiles, v_mfma_f32_32x32x16_bf16; fp32 rows or other bf16 row lengths: the 64 x 64 tiles of anncur_ivf_group_scores),
 * scores written to PACKED rows -- S float[nq x pitch], 16-byte aligned, query q's probed lists back to back, nothing pre-filled -- and
 * scanned by anncur_rowwise_topk_ragged.  Xs / Q as for anncur_ivf_group_scores(_bf16) (dtype selects fp32 or bf16 rows for BOTH).
 * pitch >= max(k, longest packed row) -- nprobe x longest list always suffices -- a multiple of 4, nq x pitch < 2^32 (split the queries); a
 * probed list that no longer fits a query's row is left out of that query's search (a pitch below the contract loses candidates, never memory).
 * max_tiles: any upper bound on sum_l ceil(pairs_l / T) ceil(size_l / T), T = anncur_ivf_search_tile(...) (the launch's grid: workgroups
 * past the last tile exit), e.g. (nq nprobe / T) max_l ceil(size_l / T) + sum_l ceil(size_l / T).  k <= 128 and nlist <= 8192
 * (ANNCUR_E_UNSUPPORTED otherwise: the calls above).  Workspace: anncur_ivf_search_workspace_bytes, 256-byte aligned. */
int32_t anncur_ivf_search_tile(int dtype, int32_t dp, int64_t ldx, int64_t ldq, int64_t nq);
size_t anncur_ivf_search_workspace_bytes(int64_t nq, int32_t nprobe, int32_t nlist, int32_t k, int64_t max_tiles);
int anncur_ivf_search_grouped(const void *Xs, int dtype, int64_t ldx, int32_t dp, const int32_t *offsets, const int32_t *ids, int32_t nlist,
                              const void *Q, int64_t ldq, int64_t nq, const int32_t *probe, int32_t nprobe, int32_t k, int64_t max_tiles,
                              float *S, int64_t pitch, void *workspace, size_t workspace_bytes, float *out_val, int32_t *out_idx, void *stream);

/* Index-build hint of anncur_score_topk_ex: bucket[i] in 0..n_buckets-1 by the squared norm of row i of the fp32 matrix A, largest
 * norms first (linear between the matrix' largest and smallest row norm); norms float[n_rows] and minmax2 uint32[2] are scratch
 * outputs.  anncur_ivf_build_lists(bucket, n_rows, n_buckets, ...) then returns the rows in coarse descending-norm order (a stable
 * counting sort): the order in which CURApprox stores the hint copy of E^T (matrix_approx_zeshel.py:65 builds latent_cols; the
 * ordering is this build's own and only moves speed). */
int anncur_norm_buckets(const float *A, int64_t n_rows, int64_t n_cols, int64_t lda, int32_t n_buckets, float *norms, uint32_t *minmax2,
                        int32_t *bucket, void *stream);

/* dtype plumbing: fp32 <-> bf16 (round-to-nearest-even), strided 2-D ------------------ */
int anncur_convert(const void *src, int src_dtype, int64_t lds_, void *dst, int dst_dtype, int64_t ldd,
                   int64_t n_rows, int64_t n_cols, void *stream);

/* Device-side byte copy (16-byte vector stores); `dst` may be mapped pinned HOST memory, so small per-step results
 * (the 4*Q overlap counts the reference's statistics are computed from, eval/eval_utils.py:131-136) reach the host from inside a
 * captured graph without a copy-engine hop.  src / dst 16-byte aligned. */
int anncur_copy_bytes(const void *src, void *dst, size_t nbytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ANNCUR_HIP_H */
