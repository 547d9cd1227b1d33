/*
 * ifr.h -- C ABI of the MI355X-native scoring + ranking-evaluation library (libifr.so).
 *
 * The reference (Informfully/Recommenders, a Cornac fork) exposes this hot path as a
 * duck-typed Python class API, not as a C FFI.  Each entry point
 * below therefore cites the reference *Python* interface it replaces (file:line relative
 * to the reference root); INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *  - extern "C", opaque handle, int status return (0 = IFR_OK, <0 = error; the text of the
 *    last error is available from ifr_last_error).  No exceptions cross the boundary.
 *  - "dev" pointers must be device memory on the handle's GPU; "any" pointers may be host
 *    or device memory (copied with hipMemcpyDefault during the call; borrowed only for its
 *    duration).  `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *  - One host thread per handle; a handle is not re-entrant.  Calls on one handle that
 *    use the same stream are ordered by it.
 *  - Item ids are int32, user ids int32, CSR row pointers int64, CSR column indices int32
 *    sorted ascending within a row (what scipy.sparse.csr_matrix.sort_indices() yields;
 *    the reference's Dataset.csr_matrix is sorted, cornac/data/dataset.py:229-238).
 */
#ifndef IFR_H
#define IFR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ifr_handle_s *ifr_handle;

enum {
    IFR_OK = 0,
    IFR_ERR_ARG = -1,        /* bad argument (null pointer, size, unsupported k/K) */
    IFR_ERR_STATE = -2,      /* called before the data it needs was set            */
    IFR_ERR_HIP = -3,        /* a HIP runtime call failed                          */
    IFR_ERR_NOMEM = -4       /* device allocation failed                           */
};

#define IFR_MAX_FACTORS 256  /* latent dimension k supported by the fused kernel   */
#define IFR_MAX_TOPK 128     /* K supported by the fused score+select kernel (dot-product models) */
#define IFR_MAX_TOPK_ITEM 32 /* ... for item-item models with a dense item matrix                  */
#define IFR_MAX_TOPK_ITEM_CSR 128 /* ... with a CSR item matrix (re-ranker pools: rerankers/reranker.py:620-639) */

/* Library / handle life cycle.  (No reference counterpart: the reference keeps model
 * parameters as NumPy arrays on the Recommender object, cornac/models/recommender.py:160-168.) */
const char *ifr_version(void);
int ifr_create(ifr_handle *out, int device_id);
int ifr_destroy(ifr_handle h);
const char *ifr_last_error(ifr_handle h);

/* Launch tuning (no reference counterpart).  "chunks": item chunks per user tile in the fused
 * kernel (0 = heuristic); "target_blocks": the launch plan considers up to ceil(target_blocks / user tiles)
 * item chunks per user tile and picks the cheapest by its cost model (rounds x (tiles per chunk + a fixed cost)); "catalogue_items":
 * see ifr_pad_scores_min; "share_bounds": 0 switches the per-user bound words between item chunks off;
 * "tail_split": 0 = one grid region; "pair_topk": 0 = 16 < K <= 100 takes the candidate-buffer path like K > 100
 * ("pair_max": largest K of the pair path, default 100, at most 128; "pair_div": that path takes one item chunk per
 * pair_div wanted items, default 12, at most 8 chunks);
 * "wide_chunk_cap", "wide_stats": tuning / statistics of that path; "item_f64": see ifr_fit_ease (none of these
 * changes a result);
 * "csr_wave": CSR top-K kernel choice -- 0 = the workgroup-per-user kernel only; 1 (default) = one wave per user where
 * the history has at most 512 rows (192 / 128 rows with two / three words per segment; the rest handed to the workgroup
 * kernel; needs the kernel's side table of one to three 16-byte words -- three entries each -- per matrix row and
 * 1024-column window, built when it stays within twice the entries' bytes / 256 MB), IF at most 0.3 % of the matrix's
 * (row, 1024-column window) segments hold more entries than those words (the word count is chosen and the share
 * measured when the matrix is set, ifr_item_csr_layout: evenly spread rows of up to ~1000 entries per 125 000 columns;
 * on matrices with uniformly drawn columns the workgroup kernel is the faster one); 2 = force the wave kernel (tests).
 * Same results either way;
 * "csr_cand_cap": entries of the CSR top-K kernel's buffer of candidates that wait to be merged into a user's list
 * (K ... 256, 0 = default 256; a value outside is clamped; tests shrink it to force the overflow path; no result changes);
 * "debug_mode": development ablations (results may change; the fused kernel's modes, bits 0-2
 * and 7 of "debug_mode", exist only in a library built with `make ABLATE=1` / `STAMPS=1` and are refused with
 * IFR_ERR_ARG otherwise; bit 6 = one workgroup per CU works everywhere).  Options are per handle and
 * set only through this call: the library reads no environment variable. */
int ifr_set_option(ifr_handle h, const char *name, int64_t value);

/* Model parameters of a dot-product model: score(u,i) = (ibase[i] + ubase[u]) + <U[u], V[i]>.
 * Replaces the parameter reads of BPR.score (cornac/models/bpr/recom_bpr.pyx:290-293:
 * ibase = i_biases), MF.score (cornac/models/mf/recom_mf.py:275-280: ibase = fl(mu + Bi),
 * ubase = Bu) and WMF.score (cornac/models/wmf/recom_wmf.py:236-237: both NULL).
 * U: [n_users][k], V: [n_items][k] row-major fp32 ("any" pointers).  k <= IFR_MAX_FACTORS. */
int ifr_set_factors(ifr_handle h, const float *U, int64_t n_users, const float *V, int64_t n_items,
                    int k, const float *ubase, const float *ibase);

/* Replace only the user side (query vectors) of a dot-product model, keeping the item factors on
 * the device: the query of BaseANN.knn_query(query[B, d], k) (cornac/models/ann/recom_ann_base.py:
 * 91-98), whose rows need not be rows of the model's user matrix.  k must equal the item factors'. */
int ifr_set_user_factors(ifr_handle h, const float *U, int64_t n_users, int k, const float *ubase);

/* Item-item models: score(u, :) = (sum_{j in hist(u)} x_uj * M[j, :]) * col_scale, the other model
 * family of the path.  Replaces the parameter reads of EASE.score (cornac/models/ease/recom_ease.py:
 * 123-124: M = B, hist(u) = row u of train_set.matrix with its ratings, col_scale NULL) and of
 * RP3_Beta.score (cornac/models/rp3_beta/graph_recommender.py:97-117 and recom_rp3beta.py:122: in
 * item-item form M = W = P_iu.P_ui, x_uj = 1/deg(u), col_scale = 1/deg(item)^beta).
 * Setting an item matrix replaces any factors set before (and the other way round); ifr_rank_topk,
 * ifr_score_dense and everything downstream then serve the item-item model.
 *   dense:  M [n_items][ld] row-major fp32, ld >= n_items.  Per (user, item) one fp32 fma chain
 *           over the history row in its CSR order, then one fp32 multiply by col_scale.
 *   csr:    rows of M with strictly ascending columns.  Products are exact (fp64), scaled by
 *           2^shift (ifr_item_fixed_point_shift: every product and partial sum stays an integer
 *           below 2^51), rounded to nearest even and summed exactly, so the result does not depend
 *           on summation order.  RANKING uses those exact sums (sum descending, item ascending):
 *           ifr_rank_topk rounds only its K winners to fp32, ifr_score_dense_f64 returns the sums
 *           scaled back (exact in fp64) and the *_f64 row consumers rank them; ifr_score_dense
 *           rounds each once to fp32.  col_scale is folded into the stored values when the matrix
 *           is set (M'[j,c] = fl32(M[j,c] * col_scale[c])); a sum that is exactly 0 is still
 *           multiplied by col_scale[c], so that 0 * inf = NaN as in the reference.
 * col_scale: [n_items] or NULL.  "any" pointers. */
int ifr_set_item_matrix_dense(ifr_handle h, const float *M, int64_t n_items, int64_t ld, const float *col_scale);
int ifr_set_item_matrix_csr(ifr_handle h, const int64_t *indptr, const int32_t *indices, const float *values,
                            int64_t n_items, int64_t nnz, const float *col_scale);

/* The fp64 option of the dense item matrix (the reference scores EASE in float64: B from np.linalg.inv on float64,
 * recom_ease.py:78-95, score X[u, :].dot(B), :123-124, and orders those float64 values, recommender.py:643-652):
 * M [n_items][ld] row-major fp64 is kept in fp64 on the device and every (user, item) sum is one fp64 fma chain over
 * the history row in CSR order, times col_scale in fp64.  ifr_score_dense_f64 returns those sums.  ifr_rank_topk RANKS
 * the fp64 sums themselves (64-bit order-preserving keys through the selection and the merge; the scores it returns
 * are the winners' sums rounded to fp32 afterwards), and the *_f64 consumers of dense rows below (ifr_select_topk_f64,
 * ifr_rank_full_f64, ifr_rank_items_f64, ifr_auc_dense_f64, ifr_eval_dense_metrics_f64) take the fp64 rows
 * unrounded.  ifr_score_dense still writes fp32 rows (the sums rounded once).  Twice the bytes of the fp32 matrix per
 * user (the path is HBM-bound). */
int ifr_set_item_matrix_dense_f64(ifr_handle h, const double *M, int64_t n_items, int64_t ld, const float *col_scale);
/* out: dev [n][ld] fp64, ld >= n_items; IFR_ERR_STATE unless the item matrix is an fp64 one. */
int ifr_score_dense_f64(ifr_handle h, const int32_t *user_ids, int64_t n, double *out, int64_t ld, void *stream);

/* EASE model build on the device, cornac/models/ease/recom_ease.py:75-95:
 *     G = X^T X + lamb I,  P = G^-1,  B = P / (-diag P) column-wise,  diag B = 0,  posB: B[B < 0] = 0
 * X = train_set.matrix as CSR [n_users][n_items] (columns strictly ascending inside a row, finite
 * ratings; "any" pointers).  G and P are fp64 like the reference's (Gram by fp64 atomics: exact for
 * integer ratings, else order-dependent in the last bits; inverse by Cholesky, the reference uses LU:
 * equal to ~1e-10 relative, tests/test_gpu_fit.py).  B becomes the handle's dense item matrix exactly
 * as if it had been passed to ifr_set_item_matrix_dense (fp32, no col_scale); call ifr_set_history_csr
 * next.  B_out: host fp64 [n_items][n_items] row-major copy of B, or NULL.  Needs 16 * n_items^2 bytes
 * of device memory while it runs.  After ifr_set_option(h, "item_f64", 1) B is installed in fp64 instead, as if passed
 * to ifr_set_item_matrix_dense_f64. */
int ifr_fit_ease(ifr_handle h, const int64_t *x_indptr, const int32_t *x_indices, const float *x_values, int64_t n_users,
                 int64_t n_items, int64_t nnz, double lamb, int posB, double *B_out);

/* Per-user histories of an item-item model: row u = the items and weights x_uj that enter user u's
 * score (EASE: train_set.matrix, recom_ease.py:75; RP3-beta: the user's train items, each weighted
 * 1/deg(u), graph_recommender.py:53-55).  Users >= n_rows have an empty history (score 0).
 * Call after ifr_set_item_matrix_*: item ids are validated against its n_items.  "any" pointers. */
int ifr_set_history_csr(ifr_handle h, const int64_t *indptr, const int32_t *indices, const float *values,
                        int64_t n_rows, int64_t nnz);

/* RP3-beta model build on the device, cornac/models/rp3_beta/graph_recommender.py:24-55, 83-84, 104-115 in
 * item-item form: P_ui = A / deg(u), P_iu = A^T / deg(item), W = P_iu . P_ui (fp64 sparse product like the
 * reference's SciPy one, rounded to fp32), col_scale = 1 / deg(item)^beta from the integer degrees.
 * A = the binarised train matrix as a CSR pattern [n_users][n_items] (the interactions with rating > 0,
 * recom_rp3beta.py:76; columns strictly ascending; "any" pointers).  Installs W as the CSR item matrix
 * (as ifr_set_item_matrix_csr would) and P_ui as the histories (as ifr_set_history_csr would); the
 * reference's (U+I)^2 cube P^3 is never built.  w_nnz: entries of W, or NULL. */
int ifr_fit_rp3beta(ifr_handle h, const int64_t *a_indptr, const int32_t *a_indices, int64_t n_users, int64_t n_items, int64_t nnz,
                    double beta, int64_t *w_nnz);

/* The power-of-two scale of the fixed-point accumulation of a CSR item matrix, derived from
 * max|M| and the largest history L1 norm so that no sum can overflow; lets a CPU checker
 * reproduce the scores bit-for-bit (no reference counterpart).  Needs matrix and histories set. */
int ifr_item_fixed_point_shift(ifr_handle h, int *shift);

/* How a CSR item matrix was laid out for the wave-per-user top-K kernel (see "csr_wave"; no reference counterpart):
 * win_words = 16-byte words per matrix row and 1024-column window (1, 2 or 3: three entries each; 0 = that table was
 * not built), long_share = the share of (row, window) segments that hold more entries than those words (the kernel is
 * taken by default where it is at most 0.3 %).  Either pointer may be NULL. */
int ifr_item_csr_layout(ifr_handle h, int *win_words, double *long_share);

/* Per-user exclusion lists = items that must not be ranked: (train U validation positives)
 * minus test positives, already filtered by rating_threshold and clipped to [0, n_items)
 * (cornac/eval_methods/base_method.py:246-251, 265-283; SURVEY.md Appendix B rules 2-3).
 * n_rows may be smaller than n_users: users >= n_rows have no exclusions (:266-270).
 * "any" pointers.  Pass nnz = 0 / NULLs to clear. */
int ifr_set_exclusion_csr(ifr_handle h, const int64_t *indptr, const int32_t *indices,
                          int64_t n_rows, int64_t nnz);

/* Per-user ground truth = test positives (rating >= threshold, item < n_items), the gt_pos of
 * cornac/eval_methods/base_method.py:257-263, 282.  "any" pointers. */
int ifr_set_truth_csr(ifr_handle h, const int64_t *indptr, const int32_t *indices,
                      int64_t n_rows, int64_t nnz);

/* One split as the reference's Dataset.csr_matrix holds it (cornac/data/dataset.py:229-238), in DEVICE memory. */
typedef struct ifr_csr_split {
    const int64_t *indptr;   /* dev [n_rows + 1]; NULL = no such split                                   */
    const int32_t *indices;  /* dev [nnz]: columns strictly ascending inside a row (canonical CSR)        */
    const void *ratings;     /* dev [nnz]: float64 (ratings_f64 != 0, the reference's dtype) or float32   */
    int64_t n_rows;
} ifr_csr_split;

/* The candidate-set builder of ranking_eval (cornac/eval_methods/base_method.py:246-283, SURVEY.md Appendix B rules
 * 1-3) on the device: from the raw train / validation (may be NULL) / test rating matrices, for the users
 * [row0, row0 + n_rows) (output row i = user row0 + i; rows past a split's n_rows are empty, :266-270):
 *     truth[u]     = test items rated >= rating_threshold, item < n_items                       (gt_pos, :257-263)
 *     exclusion[u] = (train U validation items rated >= rating_threshold, < n_items) - truth[u] (:246-251, 265-283)
 *     has_pos[u]   = u has a test item rated >= rating_threshold BEFORE the clip to n_items     (:257-259: evaluated)
 * and installs both CSRs in the handle exactly as ifr_set_truth_csr / ifr_set_exclusion_csr would (rows ascending).
 * One wave per user row, two sweeps (count, prefix sums, fill) + a merge of the train / validation runs: index work,
 * bound by the HBM read of the raw rows.  has_pos: "any" pointer to n_rows bytes, or NULL; the three counts: host,
 * or NULL.  Needs the model set (the exclusion encoding depends on the model's item count). */
int ifr_build_eval_csrs(ifr_handle h, const ifr_csr_split *train, const ifr_csr_split *val, const ifr_csr_split *test,
                        int ratings_f64, double rating_threshold, int64_t n_items, int64_t row0, int64_t n_rows,
                        uint8_t *has_pos, int64_t *truth_nnz, int64_t *excl_nnz, int *max_truth_row);

/* Reads back the truth (which = 0) or exclusion (which = 1) CSR the handle holds -- however it was set -- into "any"
 * buffers of n_rows + 1 / nnz entries; with NULL buffers only the sizes are returned.  (Host consumers: the
 * reference's per-user caches, tests.)  No reference counterpart. */
int ifr_read_eval_csr(ifr_handle h, int which, int64_t *indptr, int32_t *indices, int64_t *n_rows, int64_t *nnz);

/* Fused scoring + exclusion + top-K selection for a batch of users.  Replaces the head of
 * Recommender.rank (cornac/models/recommender.py:624-652: score, candidate gather,
 * argsort/argpartition) for every listed user at once.  Order: score descending, item index
 * ascending on ties.  out_ids/out_scores: dev [n][K]; unfilled slots (fewer than K
 * candidates) hold id -1 / score -inf.  user_ids: dev [n].
 * 1 <= K <= IFR_MAX_TOPK for dot-product models, IFR_MAX_TOPK_ITEM for dense item matrices, IFR_MAX_TOPK_ITEM_CSR for
 * CSR item matrices (K <= 64: histories of at most 512 rows by the wave-per-user kernel where the matrix suits it, see
 * "csr_wave"; everything else by the workgroup-per-user kernel).  Dot-product models, K <= 16: lane-private
 * sorted lists, merged per user; the call is asynchronous on `stream`.  16 < K <= 100 (dot-product models): the same
 * kernel; a user's two 16-entry lists per item chunk (2 ... 8 chunks, about K / 12) hold the top K unless more than
 * 16 of them fall into one list -- the merge sees that (a full list whose last entry ranks before the user's K-th)
 * and such a user is ranked again from a dense score row; the call synchronises `stream` once to learn whether any
 * user needs it.  100 < K <= 128 (dot-product models), or any K > 16 of a model on which that path had to redo more
 * than 5 % of a batch: the lists
 * only supply a lower bound of the user's K-th best score, every score that reaches it goes to a per-user candidate
 * buffer (16 K entries, at least 512) and the exact top K is selected from it -- the score matrix still never reaches
 * HBM (Recall@50 / @100, re-ranker pools: recommender.py:644-649, rerankers/reranker.py:620-639).  A user whose buffer
 * overflows (scores that rise with the item index) is ranked from a dense score row instead; to find those users the
 * call synchronises `stream` once before it returns. */
int ifr_rank_topk(ifr_handle h, const int32_t *user_ids, int64_t n, int K,
                  int32_t *out_ids, float *out_scores, void *stream);

/* ifr_rank_topk that also writes every score it computes: dense_out[n][ld] receives the rows ifr_score_dense
 * would write (bit-identical: same fma chain), so an evaluation that needs both the top-K lists and the full
 * score vector -- ranking_eval with AUC or whole-list metrics beside NDCG@k, base_method.py:275-299, or
 * Recommender.rank, which returns the ranking AND the score vector (recommender.py:624-661) -- pays one GEMM
 * instead of two.  Dot-product models only (IFR_ERR_STATE otherwise: call ifr_rank_topk and ifr_score_dense),
 * 1 <= K <= IFR_MAX_TOPK, ld a multiple of 4 and >= n_items rounded up to 32 (columns n_items .. ld-1 of a row may be
 * overwritten with padding scores), dense_out 16-byte aligned.  Exclusions do not alter dense_out. */
int ifr_rank_topk_dense(ifr_handle h, const int32_t *user_ids, int64_t n, int K, int32_t *out_ids,
                        float *out_scores, float *dense_out, int64_t ld, void *stream);

/* ifr_rank_topk_dense for a float64 dense item matrix (ifr_set_item_matrix_dense_f64 / option "item_f64"): the lists
 * are ranked on the float64 sums like ifr_rank_topk does, and the same sums are written as float64 rows (what
 * ifr_score_dense_f64 returns) -- one pass over the matrix for the top-K metrics and for AUC (recom_ease.py:123-124 ->
 * recommender.py:643-661).  K <= IFR_MAX_TOPK_ITEM, ld >= n_items.  (ifr_rank_topk_dense itself serves a dense fp32
 * item matrix the same way, with fp32 rows.) */
int ifr_rank_topk_dense_f64(ifr_handle h, const int32_t *user_ids, int64_t n, int K, int32_t *out_ids, float *out_scores,
                            double *dense_out, int64_t ld, void *stream);

/* ifr_rank_topk for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  (No counterpart in the reference.)  Same arguments, same results bit for bit as ifr_rank_topk for
 * a dot-product model (IFR_ERR_STATE otherwise), but the call makes no hipMalloc / hipFree / allocating library call:
 * the partial top-K lists and per-user bound words live in a static __device__ array of IFR_NOALLOC_SCRATCH_BYTES that
 * every GPU the library runs on holds until the process ends (its per-device address is looked up once in ifr_create).
 * Limits: n <= IFR_NOALLOC_MAX_USERS users per call (IFR_ERR_ARG beyond); K <= IFR_NOALLOC_MAX_TOPK (longer lists need
 * candidate buffers of a size that depends on the batch, and one stream synchronisation); no profiling events are
 * recorded for it.  All calls on one GPU share that GPU's array -- whatever handle they use -- and the array keeps the
 * last call's contents (the call clears what it needs at zero itself): calls that overlap on one GPU, on different
 * streams or from different handles, race.  Calls on one stream are ordered by it. */
#define IFR_NOALLOC_SCRATCH_BYTES (64u << 20)
#define IFR_NOALLOC_MAX_USERS 65536
#define IFR_NOALLOC_MAX_TOPK 16
int ifr_rank_topk_noalloc(ifr_handle h, const int32_t *user_ids, int64_t n, int K,
                          int32_t *out_ids, float *out_scores, void *stream);

#define IFR_METRICS_PER_K 6  /* per cut-off: ndcg, recall, precision, hit ratio, f1, ncrr */

/* Ranking metrics from top-K lists against the truth CSR.  Replaces NDCG.compute, Recall.compute,
 * Precision.compute, HitRatio.compute, FMeasure.compute and NCRR.compute
 * (cornac/metrics/ranking.py:100-123, 366-386, 329-349, 291-312, 403-431, 140-180) and the
 * per-user loop + mean of ranking_eval (cornac/eval_methods/base_method.py:291-303).
 * ks: host array of n_k cut-offs, each 1 <= ks[j] <= K <= 32.  out_per_user: dev
 * [n][IFR_METRICS_PER_K*n_k] (NaN rows for users without a test positive, which the reference
 * skips, :258-259).  evaluated: dev [n] 0/1.  out_sums: dev [IFR_METRICS_PER_K*n_k + 1] = column
 * sums over evaluated users followed by their count (the operands of one all-reduce in the
 * multi-GPU path). */
int ifr_eval_topk_metrics(ifr_handle h, const int32_t *user_ids, int64_t n,
                          const int32_t *topk_ids, int K, const int32_t *ks, int n_k,
                          double *out_per_user, int32_t *evaluated, double *out_sums, void *stream);

/* The same metrics from full ranked lists, for cut-offs of any size and ks[j] = -1 (whole list),
 * plus two per-user columns after the per-cut-off block: reciprocal rank of the first hit (MRR,
 * cornac/metrics/ranking.py:194-223; NaN when nothing matches, where the reference raises) and
 * average precision (MAP, :500-527, on a tie-free ranking).  ranked: dev [n][ld] item ids in
 * rank order, n_ranked: dev [n] valid entries per row.  out_per_user: dev
 * [n][IFR_METRICS_PER_K*n_k + 2]; out_sums: dev [IFR_METRICS_PER_K*n_k + 3]. */
int ifr_eval_ranked_metrics(ifr_handle h, const int32_t *user_ids, int64_t n, const int32_t *ranked,
                            int64_t ld, const int32_t *n_ranked, const int32_t *ks, int n_k,
                            double *out_per_user, int32_t *evaluated, double *out_sums, void *stream);

/* The same columns as ifr_eval_ranked_metrics, computed straight from dense score rows (as written by
 * ifr_score_dense) without building the ranking: every list metric (cornac/metrics/ranking.py:67-123,
 * 194-223, 240-274, 500-527) depends only on the positions of the user's positives in the ranking
 * "score descending, item index ascending, exclusions left out", and position(p) = number of candidates
 * that beat p.  Rows with up to 4096 positives.  scores: dev [n][ld], ld >= catalogue items. */
int ifr_eval_dense_metrics(ifr_handle h, const int32_t *user_ids, int64_t n, const float *scores, int64_t ld, const int32_t *ks,
                           int n_k, double *out_per_user, int32_t *evaluated, double *out_sums, void *stream);

/* Full score rows for a batch of users (bit-identical to the scores the fused kernel ranks).
 * Replaces Recommender.score(user_idx) (cornac/models/recommender.py:480-498) /
 * item_scores of rank (:643).  out: dev [n][ld] fp32, ld >= n_items. */
int ifr_score_dense(ifr_handle h, const int32_t *user_ids, int64_t n, float *out, int64_t ld,
                    void *stream);

/* AUC per user from dense score rows.  Replaces AUC.compute (cornac/metrics/ranking.py:473-485)
 * with gt_neg = candidates - gt_pos, candidates = [0,n_items) - exclusions
 * (base_method.py:281-283).  scores: dev [n][ld]; out_auc: dev [n] (NaN without positives). */
int ifr_auc_dense(ifr_handle h, const int32_t *user_ids, int64_t n, const float *scores,
                  int64_t ld, double *out_auc, void *stream);

/* Full ranking of dense score rows: every candidate ([0,n_items) minus exclusions) ordered by
 * score descending, item index ascending.  Replaces the k == -1 branch of Recommender.rank
 * (cornac/models/recommender.py:651-652).  out_ids: dev [n][ld] int32 (first n_cand[i] entries
 * valid, rest -1); n_cand: dev [n]. */
int ifr_rank_full(ifr_handle h, const int32_t *user_ids, int64_t n, const float *scores,
                  int64_t ld, int32_t *out_ids, int32_t *n_cand, void *stream);

/* Order an arbitrary candidate list by one user's scores (score descending, item index
 * ascending).  Replaces `item_indices[item_scores.argsort()[::-1]]` of Recommender.rank for a
 * caller-supplied item_indices (cornac/models/recommender.py:638-652).  scores: dev [n_scores]
 * indexed by item id; items: dev [m]; out_ids: dev [m]. */
int ifr_rank_items(ifr_handle h, const float *scores, int64_t n_scores, const int32_t *items, int64_t m,
                   int32_t *out_ids, void *stream);

/* Partial ranking of dense score rows: the K best candidates of every row (K <= 1024) in the order of
 * ifr_rank_full -- score descending, item index ascending, excluded items skipped, NaN first like
 * argsort()[::-1] -- without sorting the row.  Recommender.rank(k != -1) does the same with argpartition +
 * a sort of the head (cornac/models/recommender.py:644-649).  The path for cut-offs beyond the fused
 * kernel's 32.  scores: dev [n][ld] as written by ifr_score_dense (ld >= catalogue items);
 * out_ids / out_scores: dev [n][K], rows with fewer than K candidates end in -1 / -inf. */
int ifr_select_topk(ifr_handle h, const int32_t *user_ids, int64_t n, const float *scores, int64_t ld, int K, int32_t *out_ids,
                    float *out_scores, void *stream);

/* The consumers of dense score rows for fp64 rows (as written by ifr_score_dense_f64): the reference ranks the float64
 * scores of EASE / RP3-beta directly (cornac/models/recommender.py:643-652 on recom_ease.py:123-124 /
 * graph_recommender.py:104-117 outputs), so these order 64-bit keys of the doubles -- score descending, item index
 * ascending, nothing rounded before it is ranked.  Arguments, outputs and limits as their fp32 namesakes
 * (ifr_select_topk: out_scores are the selected sums rounded to fp32; ifr_eval_dense_metrics_f64: rows with up to 2048
 * positives; ifr_auc_dense_f64: up to 4096), scores: dev [n][ld] fp64. */
int ifr_select_topk_f64(ifr_handle h, const int32_t *user_ids, int64_t n, const double *scores, int64_t ld, int K, int32_t *out_ids,
                        float *out_scores, void *stream);
int ifr_rank_full_f64(ifr_handle h, const int32_t *user_ids, int64_t n, const double *scores, int64_t ld, int32_t *out_ids,
                      int32_t *n_cand, void *stream);
int ifr_rank_items_f64(ifr_handle h, const double *scores, int64_t n_scores, const int32_t *items, int64_t m, int32_t *out_ids,
                       void *stream);
int ifr_auc_dense_f64(ifr_handle h, const int32_t *user_ids, int64_t n, const double *scores, int64_t ld, double *out_auc, void *stream);
int ifr_eval_dense_metrics_f64(ifr_handle h, const int32_t *user_ids, int64_t n, const double *scores, int64_t ld, const int32_t *ks,
                               int n_k, double *out_per_user, int32_t *evaluated, double *out_sums, void *stream);
int ifr_pad_scores_min_f64(ifr_handle h, double *scores, int64_t n, int64_t ld, void *stream);

/* Items unknown to the model (evaluation with exclude_unknowns = False, cornac/eval_methods/
 * base_method.py:262-279): after ifr_set_option(h, "catalogue_items", N) with N > n_items,
 * ifr_auc_dense and ifr_rank_full treat score rows as N items wide (ld >= N, exclusions < N
 * apply), and ifr_pad_scores_min fills columns [n_items, N) of each row with the row's minimum
 * over the known items -- the padding of Recommender.rank (cornac/models/recommender.py:631-635).
 * scores: dev [n][ld] whose first n_items columns were written by ifr_score_dense. */
int ifr_pad_scores_min(ifr_handle h, float *scores, int64_t n, int64_t ld, void *stream);

/* Kernel timing with HIP events on the launch stream (used by bench.py for the roofline
 * figure; no reference counterpart).  After ifr_profile_enable(h, 1) every ifr_rank_topk
 * records events around its fused score+select kernel; ifr_profile_read synchronises those
 * events and returns the summed kernel milliseconds, launch count and users*items scored
 * since the last read. */
int ifr_profile_enable(ifr_handle h, int on);
int ifr_profile_read(ifr_handle h, double *kernel_ms, int64_t *launches, double *pairs_scored);

/* Users that ifr_rank_topk / ifr_rank_topk_dense (16 < K <= 128: lost-item check or buffer overflow) ranked from a dense score row since the handle was
 * created, because their candidate buffer overflowed (no reference counterpart; a health counter: the results are
 * the same either way, the fallback is only slower). */
int ifr_wide_overflows(ifr_handle h, int64_t *count);
/* ... and the volume behind it: users that path ranked, candidates their buffers received, candidates that reached the
 * final bound (the ones the exact selection ranks); users whose buffer overflowed count in `users` only. */
int ifr_wide_stats(ifr_handle h, int64_t *users, double *candidates, double *kept);

/* Development only (no reference counterpart): per-phase cycle sums of the fused kernel's main loop, written when the
 * library is built with `make STAMPS=1` and the option "debug_mode" has bit 7 set (tools/stamps.py).  out: host
 * [cap] 64-bit words, *n = words written ([grid][4 waves][8]); 0 words in a normal build. */
int ifr_debug_read_stamps(ifr_handle h, unsigned long long *out, int64_t cap, int64_t *n);
/* Development only: the per-user candidate counts of the handle's last 16 < K <= 128 ranking call (host out[cap]). */
int ifr_debug_wide_counts(ifr_handle h, uint32_t *out, int64_t cap, int64_t *n);

/* The k-index order of the fp32 fma chain the kernels accumulate each dot product in
 * (perm[t] = factor index consumed at chain step t, for the padded dimension the handle
 * uses); lets a CPU checker reproduce scores bit-for-bit.  perm: host [*n], *n = padded k. */
int ifr_factor_chain_order(ifr_handle h, int32_t *perm, int *n);

#ifdef __cplusplus
}
#endif
#endif /* IFR_H */
