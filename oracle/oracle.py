"""CPU oracle: NumPy restatement of the reference's scoring + ranking-eval path.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
``recommenders_amd`` never does.

Every function cites the reference file:line (relative to /root/reference) it
restates.  Pinning: ``tests/golden/ref_known_answers.json`` holds the known-answer
vectors of the reference's own unit tests (tests/cornac/metrics/test_ranking.py,
tests/cornac/utils/test_fastdot.py); ``tests/golden/*.npz`` hold outputs of the
reference itself (made by ``tests/golden/make_golden.py`` in the build container).
``tests/test_oracle.py`` checks this module and the C twin (ifr_oracle.c) against both.

The one place the restatement is *stricter* than the reference is tie order: the
reference's ``argsort()[::-1]`` leaves the order of equal scores unspecified
(tests/cornac/models/test_recommender.py:91-93 tolerates any), the oracle uses
score descending, item index ascending.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- #
# metrics (cornac/metrics/ranking.py)                                          #
# --------------------------------------------------------------------------- #
def dcg_score(gt_pos, pd_rank, k=-1):
    """ranking.py:67-98."""
    trunc = pd_rank[:k] if k > 0 else pd_rank
    rel = np.isin(trunc, gt_pos).astype(int)
    gain = 2 ** rel - 1
    discounts = np.log2(np.arange(len(rel)) + 2)
    return np.sum(gain / discounts)


def ndcg(gt_pos, pd_rank, k=-1):
    """ranking.py:100-123."""
    return dcg_score(gt_pos, pd_rank, k) / dcg_score(gt_pos, gt_pos, k)


def measure_at_k(gt_pos, pd_rank, k=-1):
    """ranking.py:240-274 -> (tp, tp_fn, tp_fp)."""
    trunc = pd_rank[:k] if k > 0 else pd_rank
    tp = int(np.sum(np.isin(trunc, gt_pos)))
    return tp, len(gt_pos), (k if k > 0 else len(trunc))


def precision(gt_pos, pd_rank, k=-1):
    """ranking.py:329-349."""
    tp, _, tp_fp = measure_at_k(gt_pos, pd_rank, k)
    return tp / tp_fp


def recall(gt_pos, pd_rank, k=-1):
    """ranking.py:366-386."""
    tp, tp_fn, _ = measure_at_k(gt_pos, pd_rank, k)
    return tp / tp_fn


def hit_ratio(gt_pos, pd_rank, k=-1):
    """ranking.py:291-312."""
    tp, _, _ = measure_at_k(gt_pos, pd_rank, k)
    return 1.0 if tp > 0 else 0.0


def f_measure(gt_pos, pd_rank, k=-1):
    """ranking.py:403-431."""
    tp, tp_fn, tp_fp = measure_at_k(gt_pos, pd_rank, k)
    prec, rec = tp / tp_fp, tp / tp_fn
    return 2 * (prec * rec) / (prec + rec) if (prec + rec) > 0 else 0


def ncrr(gt_pos, pd_rank, k=-1):
    """ranking.py:140-180."""
    trunc = pd_rank[:k] if k > 0 else pd_rank
    rec_rank = np.where(np.isin(trunc, gt_pos))[0]
    if len(rec_rank) == 0:
        return 0.0
    crr = np.sum(1.0 / (rec_rank + 1))
    max_nb_pos = min(len(gt_pos), len(trunc))
    icrr = np.sum(1.0 / (np.arange(max_nb_pos) + 1))
    return crr / icrr


def mrr(gt_pos, pd_rank):
    """ranking.py:194-223."""
    matched = np.nonzero(np.isin(pd_rank, gt_pos))[0]
    if len(matched) == 0:
        raise ValueError("No matched between ground-truth items and recommendations")
    return np.divide(1, (matched[0] + 1))


def auc(item_indices, pd_scores, gt_pos, gt_neg=None):
    """ranking.py:446-485 (pairwise form, O(|pos|*|neg|) memory like the reference)."""
    pos_mask = np.isin(item_indices, gt_pos)
    neg_mask = np.logical_not(pos_mask) if gt_neg is None else np.isin(item_indices, gt_neg)
    pos_scores = pd_scores[pos_mask]
    neg_scores = pd_scores[neg_mask]
    ui = np.repeat(pos_scores, len(neg_scores))
    uj = np.tile(neg_scores, len(pos_scores))
    return (ui > uj).sum() / len(uj)


def average_precision(item_indices, pd_scores, gt_pos):
    """ranking.py:500-527 (MAP.compute; rankdata 'max' restated without scipy)."""
    relevant = np.isin(item_indices, gt_pos)

    def rank_max(x):  # scipy.stats.rankdata(x, 'max')
        return np.array([(x <= v).sum() for v in x], dtype=float)

    rank = rank_max(-pd_scores)[relevant]
    L = rank_max(-pd_scores[relevant])
    return (L / rank).mean()


# --------------------------------------------------------------------------- #
# model scores                                                                 #
# --------------------------------------------------------------------------- #
def fast_dot(vec, mat, out):
    """cornac/utils/fast_dot.pyx:40-43: out[i] += dot(vec, mat[i, :]) in place."""
    for i in range(mat.shape[0]):
        out[i] += np.dot(vec, mat[i])
    return out


def score_bpr(U, V, Bi, u):
    """recom_bpr.pyx:290-293."""
    s = np.copy(Bi)
    s += (V @ U[u]).astype(s.dtype)
    return s


def score_mf(U, V, Bu, Bi, mu, u, knows_user=True):
    """recom_mf.py:275-280."""
    s = np.float32(mu) + Bi
    if knows_user:
        s += Bu[u]
        s += (V @ U[u]).astype(s.dtype)
    return s


def score_wmf(U, V, u):
    """recom_wmf.py:236-237."""
    return V.dot(U[u, :])


def score_ease(X_csr, B, u):
    """recom_ease.py:123-124 (fp64; the reference returns shape (1, I), flattened
    by the broadcast at recommender.py:634-635)."""
    return np.asarray(X_csr[u, :].dot(B)).ravel()


def ease_closed_form(X_csr, lamb, posB):
    """EASE.fit, cornac/models/ease/recom_ease.py:75-95: G = X^T X + lamb I, P = G^-1,
    B = P / (-diag P) column-wise, diag B = 0, posB: B[B < 0] = 0.  fp64, dense [I, I]."""
    import scipy.sparse as sp
    X = sp.csr_matrix(X_csr, dtype=np.float64)
    G = X.T.dot(X).toarray()
    di = np.diag_indices(G.shape[0])
    G[di] = G.diagonal() + lamb
    P = np.linalg.inv(G)
    B = P / (-np.diag(P))
    B[di] = 0.0
    if posB:
        B[B < 0] = 0
    return B


def rp3_item_item(A_csr):
    """graph_recommender.py:24-55, 83-84 written in item-item form.

    P = D^-1 A on the bipartite graph; P3[u, items] = P_ui[u,:] @ (P_iu @ P_ui), so
    W = P_iu @ P_ui (I x I).  The reference rounds P, P2, P3 to fp32
    (graph_recommender.py:36-38, 83-84); this restatement keeps fp64 and is
    therefore ~1e-6 rel away from it ('Precision mix').
    Returns (P_ui csr fp64, W csr fp64, item_degree int64)."""
    import scipy.sparse as sp
    A = sp.csr_matrix(A_csr, dtype=np.float64)
    A.data[:] = 1.0
    du = np.asarray(A.sum(axis=1)).ravel()
    di = np.asarray(A.sum(axis=0)).ravel()
    du_ = np.where(du == 0, 0.0001, du)              # graph_recommender.py:54
    di_ = np.where(di == 0, 0.0001, di)
    P_ui = sp.diags(1.0 / du_) @ A
    P_iu = sp.diags(1.0 / di_) @ A.T
    W = (P_iu @ P_ui).tocsr()
    return P_ui.tocsr(), W, A.getnnz(axis=0).astype(np.int64)


def score_rp3(P_ui, W, item_deg, beta, u):
    """graph_recommender.py:87-117.  item_deg is an *int* array in the reference, so
    the '0.0001' guard stores 0 and zero-degree items get 1/0**beta = inf and a
    score of 0*inf = nan; restated as such."""
    recs = np.asarray((P_ui[u, :] @ W).todense()).ravel()
    deg = item_deg.copy()
    deg[deg == 0] = 0.0001                           # int array: stays 0
    with np.errstate(divide="ignore", invalid="ignore"):
        pen = 1 / (deg ** beta)
        return recs * pen


# --------------------------------------------------------------------------- #
# rank + ranking_eval                                                          #
# --------------------------------------------------------------------------- #
def canonical_rank(item_indices, item_scores, k=-1):
    """recommender.py:637-661: order candidates by score descending; ties by
    ascending item index (the oracle's canonical rule).  Returns the full
    permutation (k = -1) or its head."""
    order = np.lexsort((item_indices, -item_scores.astype(np.float64)))
    ranked = item_indices[order]
    return ranked if k == -1 else ranked[:k]


def pos_items(csr, u, thr):
    """base_method.py:246-251."""
    if u >= csr.shape[0]:
        return np.zeros(0, dtype=np.int64)
    lo, hi = csr.indptr[u], csr.indptr[u + 1]
    idx, val = csr.indices[lo:hi], csr.data[lo:hi]
    return idx[val >= thr].astype(np.int64)


def eval_sets(train_csr, test_csr, val_csr, u, thr, n_train_items, n_test_items,
              exclude_unknowns=True):
    """base_method.py:257-283 -> (item_indices, gt_pos, gt_neg) or None if skipped."""
    test_pos = pos_items(test_csr, u, thr)
    if len(test_pos) == 0:
        return None
    pos_mask = np.zeros(n_test_items, dtype="int")
    pos_mask[test_pos] = 1
    val_pos = pos_items(val_csr, u, thr) if val_csr is not None else np.zeros(0, np.int64)
    train_pos = pos_items(train_csr, u, thr)
    neg_mask = np.ones(n_test_items, dtype="int")
    neg_mask[np.concatenate([test_pos, val_pos, train_pos]).astype(np.int64)] = 0
    if exclude_unknowns:
        pos_mask = pos_mask[:n_train_items]
        neg_mask = neg_mask[:n_train_items]
    return (np.nonzero(pos_mask + neg_mask)[0], np.nonzero(pos_mask)[0],
            np.nonzero(neg_mask)[0])


def ranking_eval(score_fn, metrics, train_csr, test_csr, val_csr=None,
                 rating_threshold=1.0, exclude_unknowns=True,
                 n_train_items=None, n_test_items=None, test_users=None, topk=None):
    """base_method.py:185-305 for a model given as ``score_fn(u) -> scores[total_items]``.

    metrics: list of (name, k) with name in {'ndcg','recall','precision','auc',
    'hit','f1','ncrr','mrr','map'}.  Returns (avg, per_user list of dicts, tops)
    where tops[u] = (top-`topk` ids, their scores) when topk is given."""
    n_train_items = train_csr.shape[1] if n_train_items is None else n_train_items
    n_test_items = test_csr.shape[1] if n_test_items is None else n_test_items
    if test_users is None:
        test_users = np.unique(test_csr.tocoo().row)
    user_results = [dict() for _ in metrics]
    tops = {}
    for u in test_users:
        u = int(u)
        sets = eval_sets(train_csr, test_csr, val_csr, u, rating_threshold,
                         n_train_items, n_test_items, exclude_unknowns)
        if sets is None:
            continue
        item_indices, gt_pos, gt_neg = sets
        all_scores = np.asarray(score_fn(u))
        item_scores = all_scores[item_indices]
        item_rank = canonical_rank(item_indices, item_scores)
        if topk is not None:
            head = item_rank[:topk]
            tops[u] = (head, all_scores[head])
        for i, (name, k) in enumerate(metrics):
            if name == "ndcg":
                v = ndcg(gt_pos, item_rank, k)
            elif name == "recall":
                v = recall(gt_pos, item_rank, k)
            elif name == "precision":
                v = precision(gt_pos, item_rank, k)
            elif name == "hit":
                v = hit_ratio(gt_pos, item_rank, k)
            elif name == "f1":
                v = f_measure(gt_pos, item_rank, k)
            elif name == "ncrr":
                v = ncrr(gt_pos, item_rank, k)
            elif name == "mrr":
                v = mrr(gt_pos, item_rank)
            elif name == "auc":
                v = auc(item_indices, item_scores, gt_pos, gt_neg)
            elif name == "map":
                v = average_precision(item_indices, item_scores, gt_pos)
            else:
                raise ValueError(name)
            user_results[i][u] = v
    avg = [sum(r.values()) / len(r) for r in user_results]   # base_method.py:302-303
    return avg, user_results, tops


# --------------------------------------------------------------------------- #
# C twin (ifr_oracle.c) via ctypes                                             #
# --------------------------------------------------------------------------- #
_LIB = None


def build_c(force=False):
    """Compile oracle/ifr_oracle.c -> oracle/libifr_oracle.so with gcc."""
    src = os.path.join(_HERE, "ifr_oracle.c")
    out = os.path.join(_HERE, "libifr_oracle.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", out, src, "-lm"])
    return out


def c_lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libifr_oracle.so")
        if not os.path.exists(path):
            path = build_c()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_auc.restype = ctypes.c_double
    return _LIB


def _p(a, t):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(t))


def c_score_factor(u_vec, V, ibase=None, ubase=0.0, kperm=None):
    V = np.ascontiguousarray(V, np.float32)
    u_vec = np.ascontiguousarray(u_vec, np.float32)
    out = np.empty(V.shape[0], np.float32)
    ib = None if ibase is None else np.ascontiguousarray(ibase, np.float32)
    kp = None if kperm is None else np.ascontiguousarray(kperm, np.int32)
    c_lib().orc_score_factor(_p(u_vec, ctypes.c_float), _p(V, ctypes.c_float), _p(ib, ctypes.c_float),
                             ctypes.c_float(ubase), V.shape[0], V.shape[1], _p(kp, ctypes.c_int32),
                             _p(out, ctypes.c_float))
    return out


def c_ranking_eval_factor(U, V, ibase, ubase, users, excl_ptr, excl_idx, truth_ptr, truth_idx,
                          ks, want_auc, kmax, kperm=None, n_threads=1):
    """Thin wrapper over orc_ranking_eval_factor; see ifr_oracle.c for the contract."""
    U = np.ascontiguousarray(U, np.float32)
    V = np.ascontiguousarray(V, np.float32)
    ib = None if ibase is None else np.ascontiguousarray(ibase, np.float32)
    ub = None if ubase is None else np.ascontiguousarray(ubase, np.float32)
    users = np.ascontiguousarray(users, np.int32)
    excl_ptr = np.ascontiguousarray(excl_ptr, np.int64)
    excl_idx = np.ascontiguousarray(excl_idx, np.int32)
    truth_ptr = np.ascontiguousarray(truth_ptr, np.int64)
    truth_idx = np.ascontiguousarray(truth_idx, np.int32)
    ks = np.ascontiguousarray(ks, np.int32)
    kp = None if kperm is None else np.ascontiguousarray(kperm, np.int32)
    n = len(users)
    ids = np.empty((n, kmax), np.int32)
    sc = np.empty((n, kmax), np.float32)
    met = np.empty((n, 3 * len(ks) + 1), np.float64)
    ev = np.empty(n, np.int32)
    c_lib().orc_ranking_eval_factor(
        _p(U, ctypes.c_float), _p(V, ctypes.c_float), _p(ib, ctypes.c_float), _p(ub, ctypes.c_float),
        V.shape[0], V.shape[1], _p(kp, ctypes.c_int32),
        _p(users, ctypes.c_int32), n,
        _p(excl_ptr, ctypes.c_int64), _p(excl_idx, ctypes.c_int32),
        _p(truth_ptr, ctypes.c_int64), _p(truth_idx, ctypes.c_int32),
        _p(ks, ctypes.c_int32), len(ks), int(want_auc), kmax,
        _p(ids, ctypes.c_int32), _p(sc, ctypes.c_float), _p(met, ctypes.c_double), _p(ev, ctypes.c_int32),
        int(n_threads))
    return ids, sc, met, ev


def _i64(a):
    return np.ascontiguousarray(a, np.int64)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, np.float32)


def c_rank_topk(scores, excl, K):
    """orc_rank_topk on one fp32 score row: (ids [K] padded with -1, scores [K] padded with -inf, candidate count)."""
    scores, excl = _f32(scores), np.sort(_i32(excl))
    ids = np.empty(K, np.int32)
    sc = np.empty(K, np.float32)
    n = c_lib().orc_rank_topk(_p(scores, ctypes.c_float), len(scores), _p(excl, ctypes.c_int32), len(excl), int(K),
                              _p(ids, ctypes.c_int32), _p(sc, ctypes.c_float))
    return ids, sc, n


def c_score_item_dense(M, h_idx, h_val, col_scale=None):
    """orc_score_item_dense: fp32 fma chain over the history in order, then * col_scale."""
    M = np.ascontiguousarray(M, np.float32)
    h_idx, h_val, cs = _i32(h_idx), _f32(h_val), _f32(col_scale)
    out = np.empty(M.shape[1], np.float32)          # (M may hold only the rows the history indexes: [R, n_items])
    c_lib().orc_score_item_dense(_p(M, ctypes.c_float), ctypes.c_int64(M.shape[1]), M.shape[1],
                                 _p(h_idx, ctypes.c_int32), _p(h_val, ctypes.c_float), len(h_idx),
                                 _p(cs, ctypes.c_float), _p(out, ctypes.c_float))
    return out


def c_score_item_dense_f64(M, h_idx, h_val, col_scale=None):
    """orc_score_item_dense_f64: fp64 fma chain over the history in order, then * col_scale in fp64.
    Returns (float64 scores, the same rounded once to fp32)."""
    M = np.ascontiguousarray(M, np.float64)                 # [rows, n_items]: square, or only the rows h_idx names
    h_idx, h_val, cs = _i32(h_idx), _f32(h_val), _f32(col_scale)
    out = np.empty(M.shape[1], np.float64)
    out32 = np.empty(M.shape[1], np.float32)
    c_lib().orc_score_item_dense_f64(_p(M, ctypes.c_double), ctypes.c_int64(M.shape[1]), M.shape[1],
                                     _p(h_idx, ctypes.c_int32), _p(h_val, ctypes.c_float), len(h_idx),
                                     _p(cs, ctypes.c_float), _p(out, ctypes.c_double), _p(out32, ctypes.c_float))
    return out, out32


def c_score_item_csr(m_ptr, m_idx, m_val, n_items, h_idx, h_val, shift, col_scale=None):
    """orc_score_item_csr: 2^shift fixed-point accumulation (order-free), then * col_scale."""
    m_ptr, m_idx, m_val = _i64(m_ptr), _i32(m_idx), _f32(m_val)
    h_idx, h_val, cs = _i32(h_idx), _f32(h_val), _f32(col_scale)
    out = np.empty(n_items, np.float32)
    rc = c_lib().orc_score_item_csr(_p(m_ptr, ctypes.c_int64), _p(m_idx, ctypes.c_int32), _p(m_val, ctypes.c_float),
                                    int(n_items), _p(h_idx, ctypes.c_int32), _p(h_val, ctypes.c_float), len(h_idx),
                                    int(shift), _p(cs, ctypes.c_float), _p(out, ctypes.c_float))
    if rc != 0:
        raise OverflowError("fixed-point sum left the 52-bit range: shift too large for these inputs")
    return out


def c_score_item_csr64(m_ptr, m_idx, m_val, n_items, h_idx, h_val, shift, col_scale=None):
    """orc_score_item_csr64: the exact fixed-point sums scaled back to float64 (what the library's CSR path ranks)."""
    m_ptr, m_idx, m_val = _i64(m_ptr), _i32(m_idx), _f32(m_val)
    h_idx, h_val, cs = _i32(h_idx), _f32(h_val), _f32(col_scale)
    out = np.empty(n_items, np.float64)
    rc = c_lib().orc_score_item_csr64(_p(m_ptr, ctypes.c_int64), _p(m_idx, ctypes.c_int32), _p(m_val, ctypes.c_float),
                                      int(n_items), _p(h_idx, ctypes.c_int32), _p(h_val, ctypes.c_float), len(h_idx),
                                      int(shift), _p(cs, ctypes.c_float), _p(out, ctypes.c_double), None)
    if rc != 0:
        raise OverflowError("fixed-point sum left the 52-bit range: shift too large for these inputs")
    return out


def c_ranking_eval_item(M, m_csr, shift, n_items, col_scale, h_ptr, h_idx, h_val, users, excl_ptr, excl_idx,
                        truth_ptr, truth_idx, ks, want_auc, kmax, n_threads=1):
    """orc_ranking_eval_item.  M: dense [I, ld] fp32 or None; m_csr: (ptr, idx, val) or None."""
    users = _i32(users)
    n = len(users)
    ks = _i32(ks)
    h_ptr, h_idx, h_val = _i64(h_ptr), _i32(h_idx), _f32(h_val)
    excl_ptr, excl_idx = _i64(excl_ptr), _i32(excl_idx)
    truth_ptr, truth_idx = _i64(truth_ptr), _i32(truth_idx)
    cs = _f32(col_scale)
    if M is not None:
        M = np.ascontiguousarray(M, np.float32)
        mp = mi = mv = None
        ld = M.shape[1]
    else:
        mp, mi, mv = _i64(m_csr[0]), _i32(m_csr[1]), _f32(m_csr[2])
        ld = 0
    ids = np.empty((n, kmax), np.int32)
    sc = np.empty((n, kmax), np.float32)
    met = np.empty((n, 3 * len(ks) + 1), np.float64)
    ev = np.empty(n, np.int32)
    rc = c_lib().orc_ranking_eval_item(
        _p(M, ctypes.c_float), ctypes.c_int64(ld),
        _p(mp, ctypes.c_int64), _p(mi, ctypes.c_int32), _p(mv, ctypes.c_float), int(shift),
        int(n_items), _p(cs, ctypes.c_float),
        _p(h_ptr, ctypes.c_int64), _p(h_idx, ctypes.c_int32), _p(h_val, ctypes.c_float), ctypes.c_int64(len(h_ptr) - 1),
        _p(users, ctypes.c_int32), n,
        _p(excl_ptr, ctypes.c_int64), _p(excl_idx, ctypes.c_int32), ctypes.c_int64(len(excl_ptr) - 1),
        _p(truth_ptr, ctypes.c_int64), _p(truth_idx, ctypes.c_int32), ctypes.c_int64(len(truth_ptr) - 1),
        _p(ks, ctypes.c_int32), len(ks), int(want_auc), int(kmax),
        _p(ids, ctypes.c_int32), _p(sc, ctypes.c_float), _p(met, ctypes.c_double), _p(ev, ctypes.c_int32),
        int(n_threads))
    if rc != 0:
        raise OverflowError("fixed-point sum left the 52-bit range")
    return ids, sc, met, ev


def item_fixed_point_shift(m_val, h_ptr, h_val, m_idx=None, col_scale=None):
    """The shift rule documented in include/ifr.h (ifr_item_fixed_point_shift): the largest power of
    two such that max|M'| * max_u sum_j |x_uj| * 1.001 * 2^shift <= 2^51, M' = the values with the
    column scale folded in (every product and partial sum then is an integer below 2^51: exact in int64 and in float64)."""
    import math
    m_val = np.asarray(m_val, np.float32)
    if col_scale is not None:
        m_val = (m_val * np.asarray(col_scale, np.float32)[np.asarray(m_idx)]).astype(np.float32)
    maxabs = float(np.max(np.abs(m_val))) if len(m_val) else 0.0
    h_ptr = np.asarray(h_ptr, np.int64)
    l1 = np.add.reduceat(np.abs(np.asarray(h_val, np.float64)), h_ptr[:-1][np.diff(h_ptr) > 0]) if len(h_val) else np.zeros(0)
    l1max = float(np.nextafter(np.float32(l1.max()), np.float32(np.inf))) if len(l1) else 0.0
    bound = maxabs * l1max * 1.001
    if not bound > 0.0:
        return 0
    _, e = math.frexp(bound)
    return max(-960, min(960, 51 - e))


def rp3_model_arrays(train_csr, beta):
    """Host arrays of the item-item form of RP3-beta from a (binarised) train matrix:
    (P_ui csr fp32 = histories, W csr fp32, col_scale fp32 = 1/deg^beta), following
    graph_recommender.py:24-55, 83-84, 104-115 (zero-degree items: 1/0**beta = inf)."""
    P_ui, W, deg = rp3_item_item(train_csr)
    with np.errstate(divide="ignore"):
        pen = (1.0 / (deg.astype(np.float64) ** beta)).astype(np.float32)
    P32 = P_ui.astype(np.float32)
    W32 = W.astype(np.float32)
    W32.sort_indices()
    P32.sort_indices()
    return P32, W32, pen
