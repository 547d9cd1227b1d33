/*
 * ifr_oracle.c -- CPU restatement of the reference's scoring + ranking-eval path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in recommenders_amd/ may import, link or
 * call this file.  It is used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / timed CPU baseline.
 *
 * Each function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  Pinned by: tests/golden/ref_known_answers.json
 * (vectors transcribed from the reference's own tests) and tests/golden/*.npz
 * (outputs of the reference itself, made by tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC -o oracle/libifr_oracle.so oracle/ifr_oracle.c -lm
 * (no -ffast-math: fmaf chains must round exactly once per step).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* a4: cornac/utils/fast_dot.pyx:25-43  out[i] += dot(vec, mat[i,:])   */
/* BLAS sdot/ddot summation order is not pinned by the reference; the  */
/* restatement uses plain k-ascending accumulation.                    */
/* ------------------------------------------------------------------ */
void orc_fast_dot_f32(const float *vec, const float *mat, float *out, int d0, int d1)
{
    for (int i = 0; i < d0; ++i) {
        float acc = 0.0f;
        for (int f = 0; f < d1; ++f) acc += vec[f] * mat[(size_t)i * d1 + f];
        out[i] += acc;
    }
}

void orc_fast_dot_f64(const double *vec, const double *mat, double *out, int d0, int d1)
{
    for (int i = 0; i < d0; ++i) {
        double acc = 0.0;
        for (int f = 0; f < d1; ++f) acc += vec[f] * mat[(size_t)i * d1 + f];
        out[i] += acc;
    }
}

/* ------------------------------------------------------------------ */
/* a3/a5/a6: factor-model score vector for one user                    */
/*   BPR  recom_bpr.pyx:290-293   s = Bi + V.U[u]                       */
/*   MF   recom_mf.py:275-280     s = ((mu + Bi) + Bu[u]) + V.U[u]      */
/*   WMF  recom_wmf.py:236-237    s = V.U[u]                            */
/* ibase[i] = Bi (BPR), fl(mu+Bi) (MF) or NULL (WMF); ubase = Bu[u] or */
/* 0.  The dot is a single-rounded fmaf chain.  kperm == NULL gives    */
/* k-ascending order; a permutation reproduces a specific hardware     */
/* accumulation order (used to compare bit-for-bit with the MFMA       */
/* kernel, whose chain order is documented in DESIGN.md).              */
/* ------------------------------------------------------------------ */
void orc_score_factor(const float *u, const float *V, const float *ibase, float ubase,
                      int n_items, int k, const int32_t *kperm, float *out)
{
    for (int i = 0; i < n_items; ++i) {
        const float *v = V + (size_t)i * k;
        float acc = 0.0f;
        if (kperm) {
            for (int t = 0; t < k; ++t) acc = fmaf(v[kperm[t]], u[kperm[t]], acc);
        } else {
            for (int t = 0; t < k; ++t) acc = fmaf(v[t], u[t], acc);
        }
        if (ibase) {
            float base = ibase[i] + ubase;
            out[i] = base + acc;
        } else {
            out[i] = acc;
        }
    }
}

/* ------------------------------------------------------------------ */
/* canonical order: score descending, item index ascending on ties.    */
/* (reference: recommender.py:645-652 argsort()[::-1]; its tie order   */
/* is unspecified -- tests/cornac/models/test_recommender.py:91-93.)   */
/* ------------------------------------------------------------------ */
typedef struct { float s; int32_t id; } orc_pair;

static int orc_pair_cmp(const void *a, const void *b)
{
    const orc_pair *x = (const orc_pair *)a, *y = (const orc_pair *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->id > y->id) - (x->id < y->id);
}

static inline int orc_beats(float s, int32_t id, float s2, int32_t id2)
{
    return (s > s2) || (s == s2 && id < id2);
}

/* a1: Recommender.rank (recommender.py:637-661) restricted to the head:
 * candidates = items [0,n_items) minus the sorted exclusion list; returns the
 * top-K ids/scores in canonical order (k <= 0: the whole candidate list).
 * Unfilled slots get id -1 / score -inf.  Returns the candidate count. */
int orc_rank_topk(const float *scores, int n_items, const int32_t *excl, int n_excl,
                  int K, int32_t *out_ids, float *out_scores)
{
    int n_cand = 0;
    if (K <= 0) {
        orc_pair *p = (orc_pair *)malloc(sizeof(orc_pair) * (size_t)(n_items > 0 ? n_items : 1));
        int e = 0;
        for (int i = 0; i < n_items; ++i) {
            while (e < n_excl && excl[e] < i) ++e;
            if (e < n_excl && excl[e] == i) continue;
            p[n_cand].s = scores[i]; p[n_cand].id = i; ++n_cand;
        }
        qsort(p, (size_t)n_cand, sizeof(orc_pair), orc_pair_cmp);
        for (int i = 0; i < n_cand; ++i) { out_ids[i] = p[i].id; out_scores[i] = p[i].s; }
        free(p);
        return n_cand;
    }
    int filled = 0, e = 0;
    for (int i = 0; i < n_items; ++i) {
        while (e < n_excl && excl[e] < i) ++e;
        if (e < n_excl && excl[e] == i) continue;
        ++n_cand;
        float s = scores[i];
        if (s != s) continue;                       /* NaN never ranks */
        if (filled == K && !orc_beats(s, i, out_scores[K - 1], out_ids[K - 1])) continue;
        int pos = filled < K ? filled : K - 1;
        while (pos > 0 && orc_beats(s, i, out_scores[pos - 1], out_ids[pos - 1])) {
            out_scores[pos] = out_scores[pos - 1]; out_ids[pos] = out_ids[pos - 1]; --pos;
        }
        out_scores[pos] = s; out_ids[pos] = i;
        if (filled < K) ++filled;
    }
    for (int i = filled; i < K; ++i) { out_ids[i] = -1; out_scores[i] = -INFINITY; }
    return n_cand;
}

/* ------------------------------------------------------------------ */
/* a11/a12: NDCG@k, Recall@k, Precision@k  (metrics/ranking.py:67-123,  */
/* 240-274, 329-349, 366-386).  ranked = head of pd_rank (n_ranked      */
/* valid entries, n_cand = full candidate-list length for k = -1),      */
/* pos = sorted gt_pos.  out[0]=ndcg out[1]=recall out[2]=precision.    */
/* ------------------------------------------------------------------ */
static int orc_in_sorted(const int32_t *a, int n, int32_t x)
{
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo < n && a[lo] == x;
}

void orc_metrics_at_k(const int32_t *ranked, int n_ranked, int n_cand,
                      const int32_t *pos, int n_pos, int k, double *out)
{
    int trunc = (k > 0) ? (k < n_ranked ? k : n_ranked) : n_ranked;
    double dcg = 0.0; int tp = 0;
    for (int i = 0; i < trunc; ++i) {
        if (ranked[i] >= 0 && orc_in_sorted(pos, n_pos, ranked[i])) {
            dcg += 1.0 / log2((double)i + 2.0);
            ++tp;
        }
    }
    int ideal = (k > 0) ? (k < n_pos ? k : n_pos) : n_pos;
    double idcg = 0.0;
    for (int i = 0; i < ideal; ++i) idcg += 1.0 / log2((double)i + 2.0);
    out[0] = dcg / idcg;                              /* ranking.py:119-121 */
    out[1] = (double)tp / (double)n_pos;              /* ranking.py:385-386 */
    out[2] = (double)tp / (double)(k > 0 ? k : n_cand); /* ranking.py:272, 348-349 */
}

/* a13: AUC.compute (metrics/ranking.py:473-485): fraction of (pos,neg)
 * pairs with score_pos > score_neg, strict.  scores indexed by item id;
 * neg = candidates - pos, candidates = [0,n_items) - excl. */
double orc_auc(const float *scores, int n_items, const int32_t *excl, int n_excl,
               const int32_t *pos, int n_pos)
{
    if (n_pos == 0) return NAN;
    float *ps = (float *)malloc(sizeof(float) * (size_t)n_pos);
    for (int i = 0; i < n_pos; ++i) ps[i] = scores[pos[i]];
    /* insertion sort ascending (n_pos is small) */
    for (int i = 1; i < n_pos; ++i) {
        float x = ps[i]; int j = i;
        while (j > 0 && ps[j - 1] > x) { ps[j] = ps[j - 1]; --j; }
        ps[j] = x;
    }
    uint64_t wins = 0, n_neg = 0;
    int e = 0, p = 0;
    for (int i = 0; i < n_items; ++i) {
        while (e < n_excl && excl[e] < i) ++e;
        while (p < n_pos && pos[p] < i) ++p;
        if (p < n_pos && pos[p] == i) continue;       /* positive: never a negative */
        if (e < n_excl && excl[e] == i) continue;     /* excluded */
        ++n_neg;
        float s = scores[i];
        /* number of positives with ps > s */
        int lo = 0, hi = n_pos;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (ps[mid] > s) hi = mid; else lo = mid + 1; }
        wins += (uint64_t)(n_pos - lo);
    }
    free(ps);
    return (double)wins / ((double)n_pos * (double)n_neg);
}

/* The same two functions on float64 score rows: the reference ranks the float64 scores of its item-item models
 * directly (recommender.py:643-652 on recom_ease.py:123-124 / graph_recommender.py:104-117 outputs).  The library's CSR
 * path accumulates exact fixed-point sums; scaled back to float64 they are exact, and this is the order it ranks in.
 * out_scores: the winners' scores rounded once to fp32. */
static inline int orc_beats64(double s, int32_t id, double s2, int32_t id2)
{
    return (s > s2) || (s == s2 && id < id2);
}

int orc_rank_topk64(const double *scores, int n_items, const int32_t *excl, int n_excl,
                    int K, int32_t *out_ids, float *out_scores)
{
    int n_cand = 0, filled = 0, e = 0;
    double *best = (double *)malloc(sizeof(double) * (size_t)(K > 0 ? K : 1));
    for (int i = 0; i < n_items; ++i) {
        while (e < n_excl && excl[e] < i) ++e;
        if (e < n_excl && excl[e] == i) continue;
        ++n_cand;
        double s = scores[i];
        if (s != s) continue;                       /* NaN never ranks */
        if (filled == K && !orc_beats64(s, i, best[K - 1], out_ids[K - 1])) continue;
        int pos = filled < K ? filled : K - 1;
        while (pos > 0 && orc_beats64(s, i, best[pos - 1], out_ids[pos - 1])) {
            best[pos] = best[pos - 1]; out_ids[pos] = out_ids[pos - 1]; --pos;
        }
        best[pos] = s; out_ids[pos] = i;
        if (filled < K) ++filled;
    }
    for (int i = 0; i < filled; ++i) out_scores[i] = (float)best[i];
    for (int i = filled; i < K; ++i) { out_ids[i] = -1; out_scores[i] = -INFINITY; }
    free(best);
    return n_cand;
}

double orc_auc64(const double *scores, int n_items, const int32_t *excl, int n_excl,
                 const int32_t *pos, int n_pos)
{
    if (n_pos == 0) return NAN;
    double *ps = (double *)malloc(sizeof(double) * (size_t)n_pos);
    for (int i = 0; i < n_pos; ++i) ps[i] = scores[pos[i]];
    for (int i = 1; i < n_pos; ++i) {
        double x = ps[i]; int j = i;
        while (j > 0 && ps[j - 1] > x) { ps[j] = ps[j - 1]; --j; }
        ps[j] = x;
    }
    uint64_t wins = 0, n_neg = 0;
    int e = 0, p = 0;
    for (int i = 0; i < n_items; ++i) {
        while (e < n_excl && excl[e] < i) ++e;
        while (p < n_pos && pos[p] < i) ++p;
        if (p < n_pos && pos[p] == i) continue;
        if (e < n_excl && excl[e] == i) continue;
        ++n_neg;
        double s = scores[i];
        int lo = 0, hi = n_pos;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (ps[mid] > s) hi = mid; else lo = mid + 1; }
        wins += (uint64_t)(n_pos - lo);
    }
    free(ps);
    return (double)wins / ((double)n_pos * (double)n_neg);
}

/* ------------------------------------------------------------------ */
/* a9: ranking_eval hot loop (eval_methods/base_method.py:253-303) for  */
/* factor models, one iteration per listed user.  CSR rows are already  */
/* threshold-filtered and clipped to [0,n_items) (SURVEY Appendix B     */
/* rules 1-3): excl = (train U val positives) - test positives, truth = */
/* test positives.  Per-user outputs (row-major [n][n_k*3 + 1]):        */
/* ndcg,recall,precision for each k, then AUC (NaN unless want_auc).    */
/* top-K ids/scores for K = kmax are returned too.  Users without a     */
/* test positive are skipped (evaluated[i] = 0).  OpenMP over users is  */
/* only the CPU-baseline convenience: per-user results do not depend on */
/* the thread count.                                                    */
/* ------------------------------------------------------------------ */
int orc_ranking_eval_factor(
    const float *U, const float *V, const float *ibase, const float *ubase,
    int n_items, int k, const int32_t *kperm,
    const int32_t *users, int n,
    const int64_t *excl_ptr, const int32_t *excl_idx,
    const int64_t *truth_ptr, const int32_t *truth_idx,
    const int32_t *ks, int n_k, int want_auc, int kmax,
    int32_t *out_ids, float *out_scores, double *out_metrics, int32_t *evaluated,
    int n_threads)
{
    int stride = n_k * 3 + 1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel
#endif
    {
        float *scores = (float *)malloc(sizeof(float) * (size_t)n_items);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (int i = 0; i < n; ++i) {
            int u = users[i];
            const int32_t *ex = excl_idx + excl_ptr[u];
            int n_ex = (int)(excl_ptr[u + 1] - excl_ptr[u]);
            const int32_t *pos = truth_idx + truth_ptr[u];
            int n_pos = (int)(truth_ptr[u + 1] - truth_ptr[u]);
            int32_t *ids = out_ids + (size_t)i * kmax;
            float *sc = out_scores + (size_t)i * kmax;
            double *m = out_metrics + (size_t)i * stride;
            orc_score_factor(U + (size_t)u * k, V, ibase, ubase ? ubase[u] : 0.0f,
                             n_items, k, kperm, scores);
            int n_cand = orc_rank_topk(scores, n_items, ex, n_ex, kmax, ids, sc);
            evaluated[i] = n_pos > 0;
            if (n_pos == 0) {                          /* base_method.py:258-259 */
                for (int j = 0; j < stride; ++j) m[j] = NAN;
                continue;
            }
            int n_ranked = kmax < n_cand ? kmax : n_cand;
            for (int j = 0; j < n_k; ++j)
                orc_metrics_at_k(ids, n_ranked, n_cand, pos, n_pos, ks[j], m + 3 * j);
            m[n_k * 3] = want_auc ? orc_auc(scores, n_items, ex, n_ex, pos, n_pos) : NAN;
        }
        free(scores);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* a7/a8: item-item models, one user's score vector                    */
/*   EASE  recom_ease.py:123-124   s = X[u,:] . B                       */
/*   RP3   graph_recommender.py:97-117 in item-item form                */
/*         s = (P_ui[u,:] . W) * 1/deg^beta,  W = P_iu . P_ui           */
/* The reference computes both in fp64 (and rounds P, P2, P3 to fp32    */
/* for RP3); the north star allows 1e-5 relative on scores.  These two  */
/* restatements follow the arithmetic the HIP path documents            */
/* (include/ifr.h) so that it can be compared bit-for-bit:              */
/*  dense: per item one fp32 fma chain over the history in CSR order,   */
/*         then one fp32 multiply by col_scale;                         */
/*  csr:   col_scale folded into the values (fl32(M * col_scale)),      */
/*         exact fp64 products scaled by 2^shift, rounded to nearest    */
/*         even, summed as int64 (order-free), rounded to fp64, to      */
/*         fp32; an exact 0 is still multiplied by col_scale.           */
/* tests/test_oracle.py pins both against the reference's own EASE and  */
/* RP3 outputs (tests/golden/toy_all_models.npz) within 1e-5 relative.  */
/* ------------------------------------------------------------------ */
void orc_score_item_dense(const float *M, int64_t ld, int n_items,
                          const int32_t *h_idx, const float *h_val, int n_hist,
                          const float *col_scale, float *out)
{
    for (int c = 0; c < n_items; ++c) {
        float acc = 0.0f;
        for (int e = 0; e < n_hist; ++e)
            acc = fmaf(h_val[e], M[(size_t)h_idx[e] * (size_t)ld + (size_t)c], acc);
        out[c] = col_scale ? acc * col_scale[c] : acc;
    }
}

/* The fp64 option of the dense item matrix: one fp64 fma chain per column in history order, times col_scale in fp64
 * (the reference's X[u, :].dot(B) in float64, cornac/models/ease/recom_ease.py:123-124, is a float64 sum over the same
 * terms; its order and roundings are SciPy's, so the two agree to a few ulp, not bit for bit).  out32 (may be NULL)
 * receives the sums rounded once to fp32: the keys the ranking uses. */
void orc_score_item_dense_f64(const double *M, int64_t ld, int n_items,
                              const int32_t *h_idx, const float *h_val, int n_hist,
                              const float *col_scale, double *out, float *out32)
{
    for (int c = 0; c < n_items; ++c) {
        double acc = 0.0;
        for (int e = 0; e < n_hist; ++e)
            acc = fma((double)h_val[e], M[(size_t)h_idx[e] * (size_t)ld + (size_t)c], acc);
        if (col_scale) acc *= (double)col_scale[c];
        out[c] = acc;
        if (out32) out32[c] = (float)acc;
    }
}

/* Fixed-point accumulation of a CSR item matrix (include/ifr.h, ifr_set_item_matrix_csr): every product x * M' is
 * exact in float64, scaled by 2^shift, rounded to the nearest-even integer and summed as an integer -- order-free.  The
 * shift keeps every partial sum below 2^52, so the sum scaled back to float64 (out64) is exact; out = that rounded once
 * to fp32.  A zero sum still meets the column scale (0 * inf = NaN like the reference).
 * returns 0, or -1 if a sum left the 52-bit range the shift must guarantee */
int orc_score_item_csr(const int64_t *m_ptr, const int32_t *m_idx, const float *m_val, int n_items,
                       const int32_t *h_idx, const float *h_val, int n_hist,
                       int shift, const float *col_scale, float *out);

int orc_score_item_csr64(const int64_t *m_ptr, const int32_t *m_idx, const float *m_val, int n_items,
                         const int32_t *h_idx, const float *h_val, int n_hist,
                         int shift, const float *col_scale, double *out64, float *out)
{
    int64_t *acc = (int64_t *)calloc((size_t)(n_items > 0 ? n_items : 1), sizeof(int64_t));
    int rc = 0;
    for (int e = 0; e < n_hist; ++e) {
        const double x = (double)h_val[e];
        const int j = h_idx[e];
        for (int64_t q = m_ptr[j]; q < m_ptr[j + 1]; ++q) {
            const float w = col_scale ? m_val[q] * col_scale[m_idx[q]] : m_val[q];   /* folded, one fp32 rounding */
            const double pr = x * (double)w;                     /* exact */
            acc[m_idx[q]] += llrint(ldexp(pr, shift));           /* nearest even */
            if (acc[m_idx[q]] > ((int64_t)1 << 52) || acc[m_idx[q]] < -((int64_t)1 << 52)) rc = -1;
        }
    }
    for (int c = 0; c < n_items; ++c) {
        double s = ldexp((double)acc[c], -shift);                /* exact: |acc| < 2^52 */
        if (s == 0.0) s = col_scale ? 0.0 * (double)col_scale[c] : 0.0;
        if (out64) out64[c] = s;
        if (out) out[c] = (float)s;
    }
    free(acc);
    return rc;
}

int orc_score_item_csr(const int64_t *m_ptr, const int32_t *m_idx, const float *m_val, int n_items,
                       const int32_t *h_idx, const float *h_val, int n_hist,
                       int shift, const float *col_scale, float *out)
{
    return orc_score_item_csr64(m_ptr, m_idx, m_val, n_items, h_idx, h_val, n_hist, shift, col_scale, NULL, out);
}

/* a9 for item-item models: the per-user loop of ranking_eval (base_method.py:253-303) with the   */
/* scores above; same outputs as orc_ranking_eval_factor.  Exactly one of M / m_ptr is given.     */
int orc_ranking_eval_item(
    const float *M, int64_t ld,
    const int64_t *m_ptr, const int32_t *m_idx, const float *m_val, int shift,
    int n_items, const float *col_scale,
    const int64_t *h_ptr, const int32_t *h_idx, const float *h_val, int64_t h_rows,
    const int32_t *users, int n,
    const int64_t *excl_ptr, const int32_t *excl_idx, int64_t excl_rows,
    const int64_t *truth_ptr, const int32_t *truth_idx, int64_t truth_rows,
    const int32_t *ks, int n_k, int want_auc, int kmax,
    int32_t *out_ids, float *out_scores, double *out_metrics, int32_t *evaluated,
    int n_threads)
{
    int stride = n_k * 3 + 1;
    int bad = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel
#endif
    {
        float *scores = (float *)malloc(sizeof(float) * (size_t)n_items);
        double *scores64 = M ? NULL : (double *)malloc(sizeof(double) * (size_t)n_items);   /* CSR: ranked on the exact sums */
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int i = 0; i < n; ++i) {
            int u = users[i];
            const int32_t *ex = NULL, *pos = NULL, *hi = NULL;
            const float *hv = NULL;
            int n_ex = 0, n_pos = 0, n_h = 0;
            if (excl_ptr && u < excl_rows) { ex = excl_idx + excl_ptr[u]; n_ex = (int)(excl_ptr[u + 1] - excl_ptr[u]); }
            if (truth_ptr && u < truth_rows) { pos = truth_idx + truth_ptr[u]; n_pos = (int)(truth_ptr[u + 1] - truth_ptr[u]); }
            if (u < h_rows) { hi = h_idx + h_ptr[u]; hv = h_val + h_ptr[u]; n_h = (int)(h_ptr[u + 1] - h_ptr[u]); }
            int32_t *ids = out_ids + (size_t)i * kmax;
            float *sc = out_scores + (size_t)i * kmax;
            double *m = out_metrics + (size_t)i * stride;
            int n_cand;
            if (M) {
                orc_score_item_dense(M, ld, n_items, hi, hv, n_h, col_scale, scores);
                n_cand = orc_rank_topk(scores, n_items, ex, n_ex, kmax, ids, sc);
            } else {
                if (orc_score_item_csr64(m_ptr, m_idx, m_val, n_items, hi, hv, n_h, shift, col_scale, scores64, scores)) bad = 1;
                n_cand = orc_rank_topk64(scores64, n_items, ex, n_ex, kmax, ids, sc);
            }
            evaluated[i] = n_pos > 0;
            if (n_pos == 0) {
                for (int j = 0; j < stride; ++j) m[j] = NAN;
                continue;
            }
            int n_ranked = kmax < n_cand ? kmax : n_cand;
            for (int j = 0; j < n_k; ++j)
                orc_metrics_at_k(ids, n_ranked, n_cand, pos, n_pos, ks[j], m + 3 * j);
            m[n_k * 3] = !want_auc ? NAN : (M ? orc_auc(scores, n_items, ex, n_ex, pos, n_pos)
                                              : orc_auc64(scores64, n_items, ex, n_ex, pos, n_pos));
        }
        free(scores);
        free(scores64);
    }
    return bad ? -1 : 0;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
