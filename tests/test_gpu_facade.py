"""GPU tests of the reference-shaped facade: Recommender.score / rank / recommend and
ranking_eval with the reference's signatures, checked against outputs of the reference itself
(tests/golden/*.npz) -- these read like the reference's own tests
(tests/cornac/eval_methods/test_base_method.py:256-283, tests/cornac/models/test_recommender.py)."""
import numpy as np
import pytest

from recommenders_amd import ScoreException, cache_rankings, ranking_eval
from recommenders_amd import metrics as M
from recommenders_amd.data import Dataset
from recommenders_amd.models import BPR, MF, WMF
from test_oracle import _csr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _sets(d):
    mk = lambda m: Dataset.from_csr(m)
    return mk(_csr(d, "train")), mk(_csr(d, "test")), (mk(_csr(d, "val")) if "val_indptr" in d.files else None)


def _model(d, name, train_set):
    if name == "bpr":
        m = BPR(k=d["bpr_U"].shape[1], init_params={"U": d["bpr_U"], "V": d["bpr_V"], "Bi": d["bpr_Bi"]})
    elif name == "mf":
        m = MF(k=d["mf_U"].shape[1], init_params={"U": d["mf_U"], "V": d["mf_V"], "Bu": d["mf_Bu"], "Bi": d["mf_Bi"],
                                                 "mu": float(d["mf_mu"])})
    else:
        m = WMF(k=d["wmf_U"].shape[1], init_params={"U": d["wmf_U"], "V": d["wmf_V"]})
    return m.fit(train_set)


def _metrics(d):
    out = []
    for n in d["metric_names"]:
        name, k = str(n).split("@")
        out.append(M.AUC() if name == "auc" else {"ndcg": M.NDCG, "recall": M.Recall, "precision": M.Precision}[name](k=int(k)))
    return out


def _tie_free(d, model):
    hs = d[f"{model}__ranked_head_scores"]
    gaps = np.abs(np.diff(hs, axis=1))
    return (gaps > 1e-5 * np.maximum(1e-3, np.abs(hs[:, :-1]))).all(axis=1)


def _gap_sets(d, name):
    mk = lambda m: Dataset.from_csr(m)
    return mk(_csr(d, "train")), mk(_csr(d, f"{name}__test")), mk(_csr(d, "val"))


@pytest.mark.parametrize("name", ["bpr", "mf", "wmf"])
def test_ranking_eval_matches_reference(gap, name):
    """The north-star bar against the reference itself, on EVERY evaluated user (no tie filter): the fixture's test split
    only holds users whose decisive reference scores are separated by > 1e-4 relative (make_golden.py::make_gap), so
    top-20 ids must be equal, per-user NDCG / Recall / Precision / AUC equal, and every mean within 1e-6."""
    d = gap
    train_set, test_set, val_set = _gap_sets(d, name)
    model = _model(d, name, train_set)
    mets = _metrics(d)
    avg, per = ranking_eval(model, mets, train_set, test_set, val_set=val_set,
                            rating_threshold=float(d["rating_threshold"]), exclude_unknowns=True)
    users = d[f"{name}__users"]
    assert sorted(per[0].keys()) == users.tolist() and len(users) > 300
    ref = d[f"{name}__per_user"]
    for j, mt in enumerate(mets):
        got = np.asarray([per[j][int(u)] for u in users])
        np.testing.assert_allclose(got, ref[:, j], atol=1e-6, rtol=0, err_msg=mt.name)
    np.testing.assert_allclose(avg, d[f"{name}__avg"], atol=1e-6, rtol=0)
    # top-20 ids of every user, through the batched entry point and through the lazy caches
    ids, scores = model.rank_batch(users, 20, exclude_csr=model._hip_cache_owner.excl_global())
    np.testing.assert_array_equal(ids, d[f"{name}__ranked_head"][:, :20])
    np.testing.assert_allclose(scores, d[f"{name}__ranked_head_scores"][:, :20], rtol=1e-5, atol=1e-6)
    ranking_eval(model, mets[:1], train_set, test_set, val_set=val_set, rating_threshold=float(d["rating_threshold"]))
    for r in range(0, len(users), 7):
        u = int(users[r])
        assert u in model.ranked_items and u in model.item_scores
        ranked = model.ranked_items[u]
        np.testing.assert_array_equal(ranked[:20], d[f"{name}__ranked_head"][r][:20])
        assert len(ranked) == d[f"{name}__n_cand"][r] == len(model.item_scores[u]) == len(model.item_scores_mapped_indices[u])
        lut = dict(zip(model.item_scores_mapped_indices[u].tolist(), model.item_scores[u].tolist()))
        np.testing.assert_allclose([lut[int(i)] for i in ranked[:20]], d[f"{name}__ranked_head_scores"][r][:20],
                                   rtol=1e-5, atol=1e-6)
    u = int(users[0])
    rk, sc = cache_rankings(model, u, None)
    assert rk is model.ranked_items[u]


@pytest.mark.parametrize("name", ["bpr", "mf", "wmf"])
def test_score_and_rank_match_reference(toy, name):
    d = toy
    train_set, _, _ = _sets(d)
    model = _model(d, name, train_set)
    for u, ref in zip(d[f"{name}__score_users"], d[f"{name}__score_vecs"]):
        s = model.score(int(u))
        assert s.dtype == np.float32 and s.shape == ref.shape
        np.testing.assert_allclose(s, ref, rtol=1e-5, atol=1e-6)
        assert float(model.score(int(u), 3)) == pytest.approx(ref[3], rel=1e-5, abs=1e-6)
    u = int(d[f"{name}__score_users"][0])
    ranked, scores = model.rank(u)                                 # all known items
    assert len(ranked) == model.num_items and len(scores) == model.num_items
    s = model.score(u)
    np.testing.assert_array_equal(ranked, np.lexsort((np.arange(len(s)), -s.astype(np.float64))))
    cand = np.array([5, 3, 99, 42, 7])
    ranked, scores = model.rank(u, item_indices=cand, k=2)
    np.testing.assert_array_equal(scores, s[cand])                 # aligned with item_indices, not with the ranking
    assert s[ranked[0]] >= s[ranked[1]] >= s[ranked[2:]].max()
    with pytest.raises(ValueError):
        model.rank(u, item_indices=cand, k=9)


def test_unknown_users_follow_the_reference(toy):
    d = toy
    train_set, _, _ = _sets(d)
    wmf = _model(d, "wmf", train_set)
    with pytest.raises(ScoreException):
        wmf.score(wmf.num_users + 5)
    wmf.global_mean = 3.5
    ranked, scores = wmf.rank(wmf.num_users + 5, item_indices=np.array([9, 2, 4]))
    np.testing.assert_array_equal(ranked, [2, 4, 9])
    np.testing.assert_array_equal(scores, [3.5, 3.5, 3.5])         # ones * default_score (recommender.py:626-627)
    mf = _model(d, "mf", train_set)
    s = mf.score(mf.num_users + 1)                                  # unknown user: mu + Bi only (recom_mf.py:276-280)
    np.testing.assert_array_equal(s, (np.float32(d["mf_mu"]) + d["mf_Bi"]).astype(np.float32))


def test_recommend_and_rank_batch(toy):
    d = toy
    train_set, _, _ = _sets(d)
    model = _model(d, "bpr", train_set)
    uid = model.user_ids[4]
    rec = model.recommend(uid, k=10)
    s = model.score(4)
    np.testing.assert_array_equal(rec, np.lexsort((np.arange(len(s)), -s.astype(np.float64)))[:10])
    rec2 = model.recommend(uid, k=10, remove_seen=True, train_set=train_set)
    seen = set(train_set.csr_matrix.getrow(4).indices.tolist())
    assert not (set(rec2) & seen) and rec != rec2 or not (set(rec) & seen)
    with pytest.raises(ValueError):
        model.recommend("nobody")
    ids, scores = model.rank_batch(np.arange(50), 10, exclude_csr=train_set.csr_matrix)
    for u in (0, 17, 49):
        sv = model.score(u).astype(np.float64)
        sv[train_set.csr_matrix.getrow(u).indices] = -np.inf
        np.testing.assert_array_equal(ids[u], np.lexsort((np.arange(len(sv)), -sv))[:10])


def test_c1_config_bpr_k50(c1):
    """BASELINE.json configs[0]: BPR k=50, ML-100K shape, NDCG@10 / Recall@10 vs the reference run."""
    d = c1
    train_set, test_set, _ = _sets(d)
    model = _model(d, "bpr", train_set)
    avg, per = ranking_eval(model, [M.NDCG(k=10), M.Recall(k=10)], train_set, test_set,
                            rating_threshold=float(d["rating_threshold"]))
    users, ref = d["bpr__users"], d["bpr__per_user"]
    assert len(per[0]) == len(users) > 900
    for j in range(2):                               # gap-guaranteed fixture: every user, no filter
        got = np.asarray([per[j][int(u)] for u in users])
        np.testing.assert_allclose(got, ref[:, j], atol=1e-6, rtol=0)
    np.testing.assert_allclose(avg, d["bpr__avg"], atol=1e-6, rtol=0)
    ids, _ = model.rank_batch(users, 10, exclude_csr=model._hip_cache_owner.excl_global())
    np.testing.assert_array_equal(ids, d["bpr__ranked_head"][:, :10])


@pytest.mark.parametrize("name", ["mf", "bpr"])
def test_unknown_items_and_users_match_reference(unknowns, name):
    """exclude_unknowns=False: the min-score padding of recommender.py:631-635 and unknown users, batched."""
    from collections import OrderedDict
    d = unknowns
    train_set, test_set, _ = _sets(d)
    model = _model(d, name, train_set)
    # the reference's id maps are global (they also hold the ids only seen in the test split)
    model.uid_map = OrderedDict((i, i) for i in range(int(d["total_users"])))
    model.iid_map = OrderedDict((i, i) for i in range(int(d["total_items"])))
    mets = _metrics(d)
    avg, per = ranking_eval(model, mets, train_set, test_set, rating_threshold=float(d["rating_threshold"]),
                            exclude_unknowns=False)
    users = d[f"{name}__users"]
    assert sorted(per[0].keys()) == users.tolist()
    ref = d[f"{name}__per_user"]
    assert len(users) > 80
    for j, mt in enumerate(mets):                    # gap-guaranteed fixture: every user, no filter
        got = np.asarray([per[j][int(u)] for u in users])
        np.testing.assert_allclose(got, ref[:, j], atol=1e-6, rtol=0, err_msg=mt.name)
    np.testing.assert_allclose(avg, d[f"{name}__avg"], atol=1e-6, rtol=0)
    # a lazily materialised full ranking covers the unknown items too, at the bottom
    u = int(users[0])
    r = 0
    ranked = model.ranked_items[u]
    assert len(ranked) == d[f"{name}__n_cand"][r]
    np.testing.assert_array_equal(ranked[:19], d[f"{name}__ranked_head"][r][:19])


def test_metric_objects_compute_like_the_reference():
    """Known answers of tests/cornac/metrics/test_ranking.py through the HIP metric kernels."""
    import json
    import os
    from conftest import GOLDEN
    known = json.load(open(os.path.join(GOLDEN, "ref_known_answers.json")))
    A = np.asarray
    for c in known["ndcg"]:
        v = M.NDCG(k=c["k"]).compute(gt_pos=A(c["gt_pos"]), pd_rank=A(c["pd_rank"]))
        assert (float("{:.2f}".format(v)) == c["expect_round2"]) if "expect_round2" in c else (v == c["expect"])
    for key, cls in (("ncrr", M.NCRR), ("hit_ratio", M.HitRatio), ("precision", M.Precision), ("recall", M.Recall),
                     ("f_measure", M.FMeasure)):
        for c in known[key]:
            exp = c["expect"] if "expect" in c else eval(c["expect_expr"])
            assert cls(k=c["k"]).compute(A(c["gt_pos"]), A(c["pd_rank"])) == pytest.approx(exp, abs=1e-15)
    for c in known["mrr"]:
        if "raises" in c:
            with pytest.raises(ValueError):
                M.MRR().compute(A(c["gt_pos"]), A(c["pd_rank"]))
        else:
            assert M.MRR().compute(A(c["gt_pos"]), A(c["pd_rank"])) == c["expect"]
    for c in known["auc"]:
        gt_neg = None if c["gt_neg"] is None else np.flatnonzero(np.isin(c["item_indices"], c["gt_neg"]))
        assert M.AUC().compute(A(c["item_indices"]), A(c["pd_scores"]), A(c["gt_pos"]), gt_neg) == c["expect"]
    for c in known["map"]:
        s = A(c["pd_scores"]) if "pd_scores" in c else np.linspace(*c["pd_scores_linspace_desc"][:2],
                                                                   int(c["pd_scores_linspace_desc"][2]))[::-1]
        exp = c["expect"] if "expect" in c else eval(c["expect_expr"])
        assert M.MAP().compute(A(c["item_indices"]), s, A(c["gt_pos"])) == pytest.approx(exp, abs=1e-15)


def test_exact_ann_backend_matches_base_model(toy):
    """HipExactANN behind the reference's BaseANN interface (knn_query / rank / recommend_batch,
    cornac/models/ann/recom_ann_base.py:91-235): exact search returns what the base model's rank does."""
    from recommenders_amd.ann import HipExactANN
    d = toy
    train_set, _, _ = _sets(d)
    bpr = _model(d, "bpr", train_set)
    bpr.train_set = train_set
    ann = HipExactANN(bpr).fit(train_set)
    assert ann.user_vectors.shape[1] == d["bpr_U"].shape[1] + 1            # bias folded in (recom_bpr.pyx:309-333)
    users = np.array([0, 5, 17, 123])
    for k in (10, 50):
        ids, dist = ann.knn_query(ann.user_vectors[users], k)
        assert ids.shape == dist.shape == (len(users), k) and (np.diff(dist, axis=1) >= 0).all()
        S = ann.user_vectors[users].astype(np.float64) @ ann.item_vectors.astype(np.float64).T
        for r in range(len(users)):
            expect = np.lexsort((np.arange(S.shape[1]), -S[r]))[:k]
            sep = np.abs(np.diff(S[r][expect])) > 1e-6
            np.testing.assert_array_equal(ids[r][:-1][sep[:k - 1]], expect[:-1][sep[:k - 1]])
            np.testing.assert_allclose(-dist[r], S[r][ids[r]], rtol=1e-5, atol=1e-6)
    # arbitrary query vectors, not rows of the user matrix
    q = np.random.default_rng(0).normal(size=(3, ann.user_vectors.shape[1])).astype(np.float32)
    ids, dist = ann.knn_query(q, 5)
    np.testing.assert_array_equal(ids, np.argsort(-(q.astype(np.float64) @ ann.item_vectors.T.astype(np.float64)), axis=1)[:, :5])
    # rank(): neighbours first, the rest in index order with -inf (recom_ann_base.py:122-137)
    ranked, scores = ann.rank(5, k=10)
    base_ranked, base_scores = bpr.rank(5)
    np.testing.assert_array_equal(ranked[:10], base_ranked[:10])
    assert len(ranked) == ann.num_items and np.isneginf(scores).sum() == ann.num_items - 10
    np.testing.assert_allclose(scores[ranked[:10]], base_scores[base_ranked[:10]], rtol=1e-5, atol=1e-6)
    rec = ann.recommend_batch([ann.user_ids[5], ann.user_ids[0]], k=10)
    assert rec[0] == [ann.item_ids[i] for i in base_ranked[:10]]
    seen = set(train_set.csr_matrix.getrow(5).indices.tolist())
    rec2 = ann.recommend_batch([ann.user_ids[5]], k=10, remove_seen=True, train_set=train_set)[0]
    assert not (set(rec2) & seen) and set(rec2) <= set(rec[0])
    with pytest.raises(ValueError):
        ann.recommend_batch(["nobody"], k=3)


def test_cutoffs_beyond_32_match_the_sorted_list_path(toy):
    """NDCG@50 / Recall@100 / Precision@50 / MRR / MAP through dense rows + ifr_eval_dense_metrics equal the values
    computed from explicitly sorted lists (ifr_rank_full + ifr_eval_ranked_metrics)."""
    d = toy
    train_set, test_set, val_set = _sets(d)
    model = _model(d, "bpr", train_set)
    mets = [M.NDCG(k=50), M.Recall(k=100), M.Precision(k=50), M.NDCG(k=-1), M.MRR(), M.MAP()]
    avg, per = ranking_eval(model, mets, train_set, test_set, val_set, rating_threshold=float(d["rating_threshold"]))
    eng = model._hip_engine()
    users = np.asarray(sorted(per[0].keys()), np.int32)
    dense = eng.score_dense(users)
    ranked, n_cand = eng.rank_full(users, dense)
    pl, _, _ = eng.eval_ranked_metrics(users, ranked, n_cand, [50, 100, -1])
    pl = pl.cpu().numpy()
    want = np.stack([pl[:, 0], pl[:, 6 + 1], pl[:, 2], pl[:, 12], pl[:, -2], pl[:, -1]], axis=1)
    got = np.asarray([[per[j][int(u)] for j in range(len(mets))] for u in users])
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    np.testing.assert_allclose(avg, want.mean(axis=0), rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", ["bpr", "ease"])
def test_list_metrics_match_reference(list_metrics, name):
    """Cut-offs beyond the fused top-K (50 / 100 / 200), k = -1, MRR, MAP, NCRR, HitRatio, FMeasure -- served by
    ifr_eval_dense_metrics -- against the per-user values of the reference's own ranking_eval."""
    from conftest import metric_objects
    from recommenders_amd.models import EASE
    d = list_metrics
    train_set, test_set, _ = _sets(d)
    if name == "bpr":
        model = _model(d, "bpr", train_set)
    else:
        model = EASE(lamb=30, trainable=False, init_params={"B": d["ease_B"]}).fit(train_set)
    mets = metric_objects(d["metric_names"])
    avg, per = ranking_eval(model, mets, train_set, test_set, rating_threshold=float(d["rating_threshold"]),
                            exclude_unknowns=True)
    users, ref = d[f"{name}__users"], d[f"{name}__per_user"]
    assert sorted(per[0].keys()) == users.tolist() and len(users) > 150
    # gap-guaranteed fixture (every positive's position is decided by gaps > 1e-4 relative): every user, no filter
    got = np.asarray([[per[j][int(u)] for j in range(len(mets))] for u in users])
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)
    np.testing.assert_allclose(avg, d[f"{name}__avg"], atol=1e-6, rtol=0)


def test_save_recommendations_full_and_pooled(toy, tmp_path):
    """Recommender.save_recommendations (recommender.py:244-274): the reference's three pickle files; with
    pool_size the heads only (what the re-rankers read), batched on the GPU, equal to the heads of the full files
    and to the reference's own ranked heads."""
    import pickle
    d = toy
    train_set, test_set, val_set = _sets(d)
    model = _model(d, "bpr", train_set)
    ranking_eval(model, [M.NDCG(k=10)], train_set, test_set, val_set=val_set,
                 rating_threshold=float(d["rating_threshold"]), exclude_unknowns=True)
    users = d["bpr__users"]
    load = lambda path: pickle.load(open(path, "rb"))
    pooled = [load(f) for f in model.save_recommendations(str(tmp_path / "pool"), pool_size=19)]
    assert [type(x) for x in pooled] == [dict, dict, dict]
    assert sorted(pooled[0]) == users.tolist()
    tf = _tie_free(d, "bpr")
    for r, u in enumerate(users.tolist()):
        assert len(pooled[0][u]) == 19 == len(pooled[1][u]) == len(pooled[2][u])
        np.testing.assert_array_equal(pooled[0][u], pooled[2][u])
        if tf[r]:
            np.testing.assert_array_equal(pooled[0][u], d["bpr__ranked_head"][r][:19])
            np.testing.assert_allclose(pooled[1][u], d["bpr__ranked_head_scores"][r][:19], rtol=1e-5, atol=1e-6)
    full = [load(f) for f in model.save_recommendations(str(tmp_path / "full"))]
    assert [type(x) for x in full] == [dict, dict, dict] and sorted(full[0]) == users.tolist()
    for r, u in enumerate(users.tolist()):
        assert len(full[0][u]) == d["bpr__n_cand"][r] == len(full[1][u]) == len(full[2][u])
        np.testing.assert_array_equal(full[0][u][:19], pooled[0][u])
        lut = dict(zip(full[2][u].tolist(), full[1][u].tolist()))
        assert [lut[i] for i in pooled[0][u].tolist()] == pooled[1][u].tolist()


def test_topk_metrics_with_auc_share_one_scoring_pass(toy):
    """NDCG@10 + AUC: ranking_eval lets the fused kernel write the score rows AUC reads (ifr_rank_topk_dense).
    Same per-user values as the separate routes (cut-off 50 keeps AUC on ifr_score_dense), also when batched."""
    d = toy
    train_set, test_set, val_set = _sets(d)
    kw = dict(val_set=val_set, rating_threshold=float(d["rating_threshold"]), exclude_unknowns=True)
    for name in ("bpr", "wmf"):
        model = _model(d, name, train_set)
        assert model._hip_engine().can_rank_topk_dense(10)
        _, fused = ranking_eval(model, [M.NDCG(k=10), M.AUC(), M.Recall(k=5)], train_set, test_set, **kw)
        _, batched = ranking_eval(_model(d, name, train_set), [M.NDCG(k=10), M.AUC(), M.Recall(k=5)], train_set, test_set,
                                  batch_pairs=37 * 512, **kw)
        _, apart = ranking_eval(_model(d, name, train_set), [M.NDCG(k=10), M.AUC(), M.Recall(k=5), M.NDCG(k=50)], train_set,
                                test_set, **kw)
        for j in range(3):
            assert fused[j] == batched[j]
            assert sorted(fused[j]) == sorted(apart[j])
            # the list route sums the DCG terms in another order: last-bit differences for NDCG, none for AUC
            np.testing.assert_allclose([fused[j][u] for u in sorted(fused[j])], [apart[j][u] for u in sorted(apart[j])],
                                       rtol=0, atol=0 if j == 1 else 1e-12)
        users = d[f"{name}__users"]
        ref_auc = d[f"{name}__per_user"][:, list(d["metric_names"]).index("auc@-1")]
        got = np.asarray([fused[1][int(u)] for u in users])
        assert np.abs(got - ref_auc).max() < 1e-4 and abs(got.mean() - ref_auc.mean()) < 1e-6


def test_user_indices_outside_the_model_raise_like_the_reference(toy):
    """a user row outside the factor arrays must never reach the kernels as an address."""
    d = toy
    train_set, _, _ = _sets(d)
    bpr = _model(d, "bpr", train_set)
    n = len(bpr.u_factors)
    for bad in (10 ** 7, n, -n - 1):
        with pytest.raises(IndexError):
            bpr.score(bad)
        with pytest.raises(IndexError):
            bpr.rank(bad)
        with pytest.raises(IndexError):
            bpr.rank_batch([0, bad], 5)
    np.testing.assert_array_equal(bpr.score(-1), bpr.score(n - 1))          # NumPy's wrap-around, as u_factors[-1]
    eng = bpr._hip_engine()
    import torch
    for ids in (np.array([0, eng.n_users], np.int64), torch.tensor([3, -1], device=eng.device)):
        with pytest.raises(IndexError):
            eng.rank_topk(ids, 5)
        with pytest.raises(IndexError):
            eng.score_dense(ids)
    # a checked tensor is freed, the allocator hands its block to a new tensor with a bad id -- the range
    # check must run again (it is keyed on the tensor object, not on the storage address)
    good = torch.arange(64, dtype=torch.int32, device=eng.device)
    eng.rank_topk(good, 5)
    del good
    bad_ids = torch.full((64,), eng.n_users + 3, dtype=torch.int32, device=eng.device)
    with pytest.raises(IndexError):
        eng.rank_topk(bad_ids, 5)
    mf = _model(d, "mf", train_set)          # MF: an unknown user keeps mu + Bi whatever the index (recom_mf.py:277-280)
    np.testing.assert_array_equal(mf.score(10 ** 7), mf.score(mf.num_users + 1))


def test_rank_batch_after_ranking_eval_excludes_nothing(toy):
    """ranking_eval leaves its exclusion CSR on the model's engine; rank_batch without exclude_csr must not
    inherit it."""
    d = toy
    train_set, test_set, val_set = _sets(d)
    model = _model(d, "bpr", train_set)
    users = np.arange(20)
    before, _ = model.rank_batch(users, 10)
    ranking_eval(model, [M.NDCG(k=10)], train_set, test_set, val_set=val_set, rating_threshold=float(d["rating_threshold"]))
    after, _ = model.rank_batch(users, 10)
    np.testing.assert_array_equal(before, after)
    seen = train_set.csr_matrix
    assert any(np.isin(after[r], seen[int(u)].indices).any() for r, u in enumerate(users)), "fixture has no seen item in any head"
