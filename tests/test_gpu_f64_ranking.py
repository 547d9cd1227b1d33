"""GPU tests of ranking on the reference's own precision for float64-scored item-item models (missing #1).

The reference orders the float64 scores of EASE / RP3-beta directly (cornac/models/recommender.py:643-652 on the outputs of
recom_ease.py:123-124 / graph_recommender.py:104-117).  With precision="f64" every consumer here ranks 64-bit keys of the
float64 sums: the fused top-K (item_item_kernel epilogue + merge), ifr_select_topk_f64, ifr_rank_full_f64,
ifr_rank_items_f64, ifr_auc_dense_f64 and ifr_eval_dense_metrics_f64.  The workloads below hold column pairs whose sums
differ by ~1e-12 relative -- equal after rounding to fp32, where the tie rule (ascending item index) would order them the
other way round -- and exact duplicates (true ties: ascending item index).  Bar: ids and integer pair counts bit-exact
against NumPy on the float64 rows (oracle.canonical_rank / auc / the list-metric functions).
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _workload(seed, n_users, n_items, mean_hist, pen):
    from recommenders_amd import synth
    rng = np.random.default_rng(seed)
    tr_ptr, tr_idx, te_ptr, te_idx = synth.interactions(rng, n_users, n_items, mean_hist, lo=1)
    h_val = rng.integers(1, 6, len(tr_idx)).astype(np.float32)
    M = np.maximum(0, rng.normal(0, 1e-2, (n_items, n_items)))
    np.fill_diagonal(M, 0.0)
    # near-ties that only float64 resolves: column a (the LARGER index) is column b times (1 + 1e-12), so a beats b in
    # float64 while the fp32-rounded sums tie and the tie rule would put b first; and exact duplicates (true ties)
    pairs = rng.permutation(n_items)[: 2 * (n_items // 8)].reshape(-1, 2)
    for a, b in pairs:
        a, b = max(a, b), min(a, b)
        M[:, a] = M[:, b] * (1.0 + 1e-12)
    dup = rng.permutation(n_items)[: 2 * (n_items // 16)].reshape(-1, 2)
    for a, b in dup:
        M[:, a] = M[:, b]
    col_scale = (1.0 / rng.integers(1, 500, n_items) ** 0.7).astype(np.float32) if pen else None
    if pen:                      # pairs keep one column scale, else the scale decides them
        for a, b in np.concatenate([pairs, dup]):
            col_scale[a] = col_scale[b]
    return dict(M64=M, col_scale=col_scale, h_ptr=tr_ptr, h_idx=tr_idx, h_val=h_val, excl_ptr=tr_ptr, excl_idx=tr_idx,
                truth_ptr=te_ptr, truth_idx=te_idx, n_users=n_users, n_items=n_items, pairs=pairs)


def _scorer(w):
    from recommenders_amd import HipScorer
    sc = HipScorer(0)
    sc.set_item_matrix_dense(w["M64"], w["col_scale"], precision="f64")
    sc.set_history_csr(w["h_ptr"], w["h_idx"], w["h_val"])
    sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
    sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
    return sc


def _ap(pos, ranking):
    """Average precision on the canonical (tie-free) ranking: the exact duplicates of these workloads are true ties,
    where the reference's MAP (ranking.py:500-527, rankdata 'max') is tie-aware and the kernels' is defined on the
    canonical order (include/ifr.h, ifr_eval_ranked_metrics)."""
    where = np.sort(np.flatnonzero(np.isin(ranking, pos)))
    return float(np.mean((np.arange(len(where)) + 1.0) / (where + 1.0)))


def _cand(w, u):
    ex = w["excl_idx"][w["excl_ptr"][u]: w["excl_ptr"][u + 1]]
    return np.setdiff1d(np.arange(w["n_items"]), ex)


@pytest.mark.parametrize("n_items,pen", [(3001, False), (9001, True)])
def test_every_consumer_ranks_the_float64_sums(torch, n_items, pen):
    w = _workload(5 + n_items, 150, n_items, 30, pen)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    rows = sc.score_rows(users)
    assert rows.dtype == torch.float64
    r64 = rows.cpu().numpy()
    r32 = r64.astype(np.float32)
    full, n_cand = (t.cpu().numpy() for t in sc.rank_full(users, rows))
    top10, s10 = (t.cpu().numpy() for t in sc.rank_topk(users, 10))
    top32, s32 = (t.cpu().numpy() for t in sc.rank_topk(users, 32))
    top100, s100 = (t.cpu().numpy() for t in sc.rank_topk(users, 100))            # dense rows + ifr_select_topk_f64
    top1000, _ = (t.cpu().numpy() for t in sc.rank_topk(users[:20], 1000))
    auc = sc.auc_dense(users, rows).cpu().numpy()
    ks = [10, 50, -1]
    per, ev, _ = sc.eval_dense_metrics(users, rows, ks)
    per, ev = per.cpu().numpy(), ev.cpu().numpy()
    differs = 0
    for i, u in enumerate(users):
        cand = _cand(w, u)
        want = orc.canonical_rank(cand, r64[i][cand])
        differs += int(not np.array_equal(want[:100], orc.canonical_rank(cand, r32[i][cand].astype(np.float64))[:100]))
        assert n_cand[i] == len(cand)
        np.testing.assert_array_equal(full[i][: len(cand)], want)
        np.testing.assert_array_equal(top10[i], want[:10])
        np.testing.assert_array_equal(top32[i], want[:32])
        np.testing.assert_array_equal(top100[i], want[:100])
        np.testing.assert_array_equal(s32[i], r32[i][want[:32]])
        np.testing.assert_array_equal(s100[i], r32[i][want[:100]])
        if i < 20:
            np.testing.assert_array_equal(top1000[i], want[:1000])
        pos = w["truth_idx"][w["truth_ptr"][u]: w["truth_ptr"][u + 1]]
        pos = pos[np.isin(pos, cand)]
        if len(w["truth_idx"][w["truth_ptr"][u]: w["truth_ptr"][u + 1]]) == 0:
            assert ev[i] == 0
            continue
        # AUC: exact integer pair counts on the float64 scores
        if len(pos):
            assert auc[i] == orc.auc(cand, r64[i][cand], pos)
        exp = []
        for k in ks:
            exp += [orc.ndcg(pos, want, k), orc.recall(pos, want, k), orc.precision(pos, want, k), orc.hit_ratio(pos, want, k),
                    orc.f_measure(pos, want, k), orc.ncrr(pos, want, k)]
        if len(pos):
            exp += [orc.mrr(pos, want), _ap(pos, want)]
            np.testing.assert_allclose(per[i], exp, rtol=0, atol=1e-12)
    # the workload does what it is for: ranking the fp32-rounded sums would order many heads differently
    assert differs > len(users) // 2
    # rank_items on a caller's candidate list, in any order
    rng = np.random.default_rng(0)
    items = rng.permutation(n_items)[: n_items // 2].astype(np.int32)
    got = sc.rank_items(rows[3].contiguous(), torch.as_tensor(items, device=rows.device)).cpu().numpy()
    np.testing.assert_array_equal(got, orc.canonical_rank(items.astype(np.int64), r64[3][items]))


def test_facade_ranks_float64_like_the_reference(torch):
    """EASE(precision="f64") through score / rank / ranking_eval / the lazy caches: every ordering is the float64 one."""
    from recommenders_amd import ranking_eval
    from recommenders_amd import metrics as M
    from recommenders_amd.data import Dataset
    from recommenders_amd.models import EASE
    w = _workload(77, 120, 2000, 25, False)
    X = sp.csr_matrix((w["h_val"].astype(np.float64), w["h_idx"], w["h_ptr"]), shape=(w["n_users"], w["n_items"]))
    T = sp.csr_matrix((np.ones(len(w["truth_idx"])), w["truth_idx"], w["truth_ptr"]), shape=X.shape)
    train_set, test_set = Dataset.from_csr(X), Dataset.from_csr(T)
    model = EASE(trainable=False, init_params={"B": w["M64"]}, precision="f64").fit(train_set)
    mets = [M.NDCG(k=10), M.Recall(k=20), M.AUC(), M.NDCG(k=-1), M.MRR(), M.MAP(), M.Recall(k=100)]
    avg, per = ranking_eval(model, mets, train_set, test_set, rating_threshold=1.0)
    ref = X.dot(w["M64"])                                   # the reference's float64 scores up to summation order ...
    for u in list(per[0].keys())[::5]:
        s = model.score(u).ravel()
        np.testing.assert_allclose(s, ref[u], rtol=1e-13, atol=1e-300)
        # ... so expectations are built on the scores the model itself returns: the ORDER is what is under test
        seen = X[u].indices[X[u].data >= 1.0]
        pos = T[u].indices
        cand = np.setdiff1d(np.arange(w["n_items"]), np.setdiff1d(seen, pos))
        want = orc.canonical_rank(cand, s[cand])
        ranked = model.ranked_items[u]
        np.testing.assert_array_equal(ranked, want)
        np.testing.assert_array_equal(model.item_scores[u], s[model.item_scores_mapped_indices[u]])
        exp = [orc.ndcg(pos, want, 10), orc.recall(pos, want, 20), orc.auc(cand, s[cand], pos), orc.ndcg(pos, want, -1),
               orc.mrr(pos, want), _ap(pos, want), orc.recall(pos, want, 100)]
        np.testing.assert_allclose([per[j][u] for j in range(len(mets))], exp, rtol=0, atol=1e-12)
        r2, sc2 = model.rank(u, item_indices=cand[::3], k=5)
        np.testing.assert_array_equal(r2, orc.canonical_rank(cand[::3], s[cand[::3]]))
        np.testing.assert_array_equal(sc2, s[cand[::3]])
