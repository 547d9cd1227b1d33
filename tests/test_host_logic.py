"""CPU tests of the host side: C-ABI surface, CSR preparation, metric planning, result assembly
and caches of recommenders_amd.eval, run against the oracle-backed fake engine and pinned by
the reference-generated fixtures."""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT
from fake_engine import OracleEngine
from oracle import oracle as orc
from recommenders_amd import _lib, dist as ifr_dist
from recommenders_amd import eval as ifr_eval
from recommenders_amd import metrics as M
from recommenders_amd.data import Dataset
from test_oracle import _csr


# ------------------------------------------------------------------ C ABI surface
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ifr.h")).read()
    declared = sorted(set(re.findall(r"\b(ifr_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 15
    assert os.path.exists(_lib.LIB_PATH), "build libifr.so first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ifr.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in recommenders_amd/_lib.py"
    lib.ifr_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.ifr_version()


def test_no_oracle_import_in_product_package():
    pkg = os.path.join(ROOT, "recommenders_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("ifr_oracle", "").lower() or f == "synth.py", (dirpath, f)


# ------------------------------------------------------------------ fixtures -> datasets
def _datasets(d):
    train, test = _csr(d, "train"), _csr(d, "test")
    val = _csr(d, "val") if "val_indptr" in d.files else None
    mk = lambda m: None if m is None else Dataset.from_csr(m)
    return mk(train), mk(test), mk(val)


def _spec_objects(d):
    out = []
    for n in d["metric_names"]:
        name, k = str(n).split("@")
        k = int(k)
        out.append({"ndcg": M.NDCG, "recall": M.Recall, "precision": M.Precision}[name](k=k) if name != "auc" else M.AUC())
    return out


def _engine(d, model, n_user_rows):
    if model == "bpr":
        U, V, ub, ib = d["bpr_U"], d["bpr_V"], None, d["bpr_Bi"]
    elif model == "mf":
        U, V, ub, ib = d["mf_U"], d["mf_V"], d["mf_Bu"], (np.float32(d["mf_mu"]) + d["mf_Bi"]).astype(np.float32)
    else:
        U, V, ub, ib = d["wmf_U"], d["wmf_V"], None, None
    Ue = np.zeros((max(n_user_rows, len(U)) + 1, U.shape[1]), np.float32)
    Ue[: len(U)] = U
    if ub is not None:
        ube = np.zeros(len(Ue), np.float32)
        ube[: len(ub)] = ub
        ub = ube
    return OracleEngine(Ue, V, ub, ib)


def test_build_eval_csrs_matches_reference_rule(toy):
    d = toy
    train, test, val = _csr(d, "train"), _csr(d, "test"), _csr(d, "val")
    thr, ni = float(d["rating_threshold"]), int(d["num_items"])
    truth, excl, has_pos = ifr_eval.build_eval_csrs(train, test, val, thr, ni, test.shape[0])
    for u in range(test.shape[0]):
        sets = orc.eval_sets(train, test, val, u, thr, ni, test.shape[1], True)
        if sets is None:
            assert not has_pos[u]
            continue
        item_indices, gt_pos, gt_neg = sets
        assert has_pos[u]
        np.testing.assert_array_equal(truth[u].indices, gt_pos)
        cand = np.setdiff1d(np.arange(ni), excl[u].indices)
        np.testing.assert_array_equal(cand, item_indices)


@pytest.mark.parametrize("model", ["bpr", "mf", "wmf"])
def test_ranking_eval_host_logic_against_reference(gap, model):
    """The facade's host logic (candidate sets, metric plan, aggregation) on the CPU stand-in engine against the
    reference's ranking_eval on the gap-guaranteed fixture: EVERY evaluated user, per-user values to 1e-9, means to
    1e-6 (the north-star bar), no tie filter."""
    d = gap
    train_set, _, val_set = _datasets(d)
    test_set = Dataset.from_csr(_csr(d, f"{model}__test"))
    fake_model = types.SimpleNamespace(is_fitted=True)
    eng = _engine(d, model, test_set.num_users)
    mets = _spec_objects(d)
    avg, per = ifr_eval.ranking_eval(fake_model, mets, train_set, test_set, val_set,
                                     rating_threshold=float(d["rating_threshold"]), exclude_unknowns=True, _engine=eng)
    users = d[f"{model}__users"]
    assert sorted(per[0].keys()) == users.tolist() and len(users) > 300
    ref = d[f"{model}__per_user"]
    for j in range(len(mets)):
        got = np.asarray([per[j][int(u)] for u in users])
        np.testing.assert_allclose(got, ref[:, j], atol=1e-9, rtol=0)
        assert avg[j] == pytest.approx(got.mean(), abs=1e-12)
    np.testing.assert_allclose(avg, d[f"{model}__avg"], atol=1e-6, rtol=0)


def test_ranking_eval_full_list_metrics_and_errors(toy):
    d = toy
    train_set, test_set, val_set = _datasets(d)
    eng = _engine(d, "bpr", test_set.num_users)
    fake = types.SimpleNamespace(is_fitted=True)
    mets = [M.NDCG(k=-1), M.Recall(k=50), M.MRR(), M.MAP(), M.HitRatio(k=10), M.FMeasure(k=10), M.NCRR(k=10)]
    avg, per = ifr_eval.ranking_eval(fake, mets, train_set, test_set, val_set,
                                     rating_threshold=float(d["rating_threshold"]), _engine=eng)
    # cross-check three users against the NumPy restatement of the reference classes
    train, test, val = _csr(d, "train"), _csr(d, "test"), _csr(d, "val")
    for u in list(per[0].keys())[:3]:
        item_indices, gt_pos, _ = orc.eval_sets(train, test, val, u, float(d["rating_threshold"]),
                                                int(d["num_items"]), test.shape[1], True)
        s = orc.score_bpr(d["bpr_U"], d["bpr_V"], d["bpr_Bi"], u)
        rank = orc.canonical_rank(item_indices, s[item_indices])
        exp = [orc.ndcg(gt_pos, rank, -1), orc.recall(gt_pos, rank, 50), orc.mrr(gt_pos, rank),
               orc.average_precision(item_indices, s[item_indices].astype(np.float64), gt_pos),
               orc.hit_ratio(gt_pos, rank, 10), orc.f_measure(gt_pos, rank, 10), orc.ncrr(gt_pos, rank, 10)]
        for j in range(len(mets)):
            assert per[j][u] == pytest.approx(exp[j], abs=1e-9)
    assert ifr_eval.ranking_eval(fake, [], train_set, test_set) == ([], [])
    with pytest.raises(RuntimeError):
        ifr_eval.ranking_eval(types.SimpleNamespace(is_fitted=False), mets, train_set, test_set, _engine=eng)
    with pytest.raises(TypeError):
        ifr_eval.ranking_eval(fake, mets, train_set, test_set)          # no HIP path, no fallback
    empty = Dataset.from_csr(sp.csr_matrix(test_set.csr_matrix.shape))
    with pytest.raises(ZeroDivisionError):
        ifr_eval.ranking_eval(fake, [M.NDCG(k=10)], train_set, empty, _engine=eng)


def test_list_metrics_host_logic_against_reference(list_metrics):
    """k = -1 / 50 / 100 / 200 cut-offs, MRR, MAP, NCRR, HitRatio, FMeasure through ranking_eval's list route:
    the reference's own per-user values (fixture made by the reference's ranking_eval)."""
    from conftest import metric_objects
    d = list_metrics
    train_set, test_set, _ = _datasets(d)
    eng = _engine(d, "bpr", test_set.num_users)
    mets = metric_objects(d["metric_names"])
    thr, ni = float(d["rating_threshold"]), int(d["num_items"])
    avg, per = ifr_eval.ranking_eval(types.SimpleNamespace(is_fitted=True), mets, train_set, test_set,
                                     rating_threshold=thr, exclude_unknowns=True, _engine=eng)
    users, ref = d["bpr__users"], d["bpr__per_user"]
    assert sorted(per[0].keys()) == users.tolist() and len(users) > 150
    # gap-guaranteed fixture: every user, no filter
    got = np.asarray([[per[j][int(u)] for j in range(len(mets))] for u in users])
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)
    np.testing.assert_allclose(avg, d["bpr__avg"], atol=1e-6, rtol=0)


def test_candidate_pools_are_the_heads_of_the_cached_rankings(toy):
    """export_candidate_pools (the re-ranker hand-off, reranker.py:620-639): pool_size ids + aligned scores per
    cached user, equal to the head of the full ranking the lazy caches would materialise."""
    d = toy
    train_set, test_set, val_set = _datasets(d)
    eng = _engine(d, "bpr", test_set.num_users)

    class FakeModel:
        is_fitted = True

        def __init__(self):
            self.ranked_items, self.item_scores, self.item_scores_mapped_indices = {}, {}, {}

        def rank(self, user_idx, item_indices=None, k=-1):
            s = orc.score_bpr(d["bpr_U"], d["bpr_V"], d["bpr_Bi"], user_idx).astype(np.float32)
            cand = np.asarray(item_indices)
            ranked = orc.canonical_rank(cand, s[cand])
            self.ranked_items[user_idx], self.item_scores[user_idx] = ranked, s[cand]
            self.item_scores_mapped_indices[user_idx] = cand
            return ranked, s[cand]

    model = FakeModel()
    with pytest.raises(RuntimeError):
        ifr_eval.export_candidate_pools(model, 10, _engine=eng)
    ifr_eval.ranking_eval(model, [M.NDCG(k=10)], train_set, test_set, val_set,
                          rating_threshold=float(d["rating_threshold"]), _engine=eng)
    users = d["bpr__users"]
    first = int(users[0])
    full_first = model.ranked_items[first]                      # one user materialised before the export
    for pool in (5, 40):
        ranked, scores, mapped = ifr_eval.export_candidate_pools(model, pool, batch_users=64, _engine=eng)
        assert sorted(ranked) == users.tolist() == sorted(scores) == sorted(mapped)
        np.testing.assert_array_equal(ranked[first], full_first[:pool])
        for u in users[:: max(1, len(users) // 12)]:
            u = int(u)
            s = orc.score_bpr(d["bpr_U"], d["bpr_V"], d["bpr_Bi"], u).astype(np.float32)
            assert len(ranked[u]) == pool == len(scores[u])
            np.testing.assert_array_equal(mapped[u], ranked[u])
            np.testing.assert_allclose(scores[u], s[ranked[u]], rtol=1e-6, atol=1e-7)
            assert (np.diff(scores[u]) <= 0).all()
    for bad in (0, -3, 5000):
        with pytest.raises(ValueError):
            ifr_eval.export_candidate_pools(model, bad, _engine=eng)


def test_plan_metrics_and_names():
    specs, ks, need_auc, need_list = ifr_eval.plan_metrics([M.NDCG(k=10), M.Recall(k=10), M.Precision(k=5)])
    assert specs == [("ndcg", 10), ("recall", 10), ("precision", 5)] and ks == [5, 10] and not need_auc and not need_list
    assert ifr_eval.plan_metrics([M.AUC()])[2] is True
    assert ifr_eval.plan_metrics([M.NDCG()])[3] is True and ifr_eval.plan_metrics([M.Recall(k=200)])[3] is True
    assert ifr_eval.plan_metrics([M.Recall(k=100)])[3] is False and ifr_eval.plan_metrics([M.Recall(k=100)], 32)[3] is True
    assert M.NDCG(k=10).name == "NDCG@10" and M.NDCG().name == "NDCG@-1" and M.FMeasure(k=2).name == "F1@2"
    assert M.AUC().name == "AUC" and M.MRR().name == "MRR" and M.MAP().name == "MAP" and M.NDCG().type == "ranking"
    with pytest.raises(TypeError):
        ifr_eval.plan_metrics([types.SimpleNamespace(k=3, name="Gini")])


def test_shard_range_partitions():
    for n in (0, 1, 7, 138493):
        for world in (1, 2, 3, 8):
            parts = [ifr_dist.shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_lazy_cache_and_cache_rankings():
    calls = []

    class Model:
        is_fitted = True

        def __init__(self):
            self.ranked_items, self.item_scores, self.item_scores_mapped_indices = {}, {}, {}

        def rank(self, user_idx, item_indices=None, k=-1):
            calls.append(user_idx)
            r = np.asarray(item_indices)[::-1].copy()
            self.ranked_items[user_idx] = r
            self.item_scores[user_idx] = np.zeros(len(r))
            self.item_scores_mapped_indices[user_idx] = np.asarray(item_indices)
            return r, self.item_scores[user_idx]

    m = Model()
    excl = sp.csr_matrix(np.array([[1, 0, 0, 0], [0, 0, 1, 1]], dtype=np.int8))
    owner = ifr_eval._CacheOwner(m, excl, 4, np.array([0, 1]))
    for f in ("ranked_items", "item_scores", "item_scores_mapped_indices"):
        setattr(m, f, ifr_eval.LazyRankCache(owner, f, owner.users, {}))
    assert 1 in m.ranked_items and 5 not in m.ranked_items and len(m.ranked_items) == 2 and not calls
    np.testing.assert_array_equal(m.ranked_items[1], [1, 0])          # candidates {0, 1}, reversed by the stub
    assert calls == [1]
    np.testing.assert_array_equal(m.item_scores_mapped_indices[1], [0, 1])
    r, s = ifr_eval.cache_rankings(m, 1, np.arange(4))                  # cache hit: item_indices ignored (:166-167)
    np.testing.assert_array_equal(r, [1, 0])
    assert calls == [1]
    ifr_eval.cache_rankings(m, 0, np.array([1, 2, 3]))
    assert calls == [1, 0]
    with pytest.raises(RuntimeError):
        ifr_eval.cache_rankings(types.SimpleNamespace(is_fitted=False), 0, None)
    import pickle
    assert isinstance(pickle.loads(pickle.dumps(m.ranked_items)), dict)


def test_rp3_facade_hands_prebuilt_arrays_to_the_engine_unchanged(toy):
    """RP3_Beta(init_params=...): arrays built elsewhere (here: the oracle's restatement of graph_recommender.py:24-55,
    pinned by the reference's scores) reach the engine as fp32 W / P_ui / 1/deg^beta, and the train set is not
    modified.  (The package itself has no host build: `fit` without init_params runs ifr_fit_rp3beta on the GPU.)"""
    from recommenders_amd.models import RP3_Beta
    d = toy
    train_set, _, _ = _datasets(d)
    before = train_set.csr_matrix.data.copy()
    A_bin = sp.csr_matrix((_csr(d, "train") > 0).astype(np.float64))
    P_ui, W, deg = orc.rp3_item_item(A_bin)
    m = RP3_Beta(beta=float(d["rp3_beta"]), init_params={"P_ui": P_ui, "W": W, "item_degrees": deg}).fit(train_set)
    np.testing.assert_array_equal(train_set.csr_matrix.data, before)
    P32, W32, pen = orc.rp3_model_arrays(A_bin, float(d["rp3_beta"]))
    M, hist, col_scale = m._hip_item_model()
    np.testing.assert_array_equal(hist.toarray(), P32.toarray())
    np.testing.assert_array_equal(np.asarray(M), W32.toarray())
    np.testing.assert_array_equal(col_scale, pen)
    with pytest.raises(ValueError):
        RP3_Beta(init_params={"W": W})


@pytest.mark.parametrize("model", ["mf", "bpr"])
def test_ranking_eval_with_unknown_items_and_users(unknowns, model):
    """exclude_unknowns=False (fixture made by BaseMethod.from_splits): items the model has never seen
    score the user's minimum and rank last (recommender.py:631-635), unknown users keep only the
    item-side terms; per-user values equal the reference's."""
    d = unknowns
    train_set, test_set, _ = _datasets(d)
    eng = _engine(d, model, test_set.num_users)
    assert (eng.n_items < test_set.num_items) == (model == "mf")
    mets = _spec_objects(d)
    avg, per = ifr_eval.ranking_eval(types.SimpleNamespace(is_fitted=True), mets, train_set, test_set,
                                     rating_threshold=float(d["rating_threshold"]), exclude_unknowns=False, _engine=eng)
    users = d[f"{model}__users"]
    assert sorted(per[0].keys()) == users.tolist()
    ref = d[f"{model}__per_user"]
    assert len(users) > 80
    for j in range(len(mets)):                      # gap-guaranteed fixture: every user, no filter
        got = np.asarray([per[j][int(u)] for u in users])
        np.testing.assert_allclose(got, ref[:, j], atol=1e-9, rtol=0)
    np.testing.assert_allclose(avg, d[f"{model}__avg"], atol=1e-6, rtol=0)


def test_append_unknown_items_fills_short_lists():
    import torch
    ids = torch.tensor([[5, 2, -1, -1], [1, 2, 3, 4], [-1, -1, -1, -1]], dtype=torch.int32)
    excl = sp.csr_matrix(np.array([[0, 0, 0, 0, 0, 0, 1, 0, 0], [0] * 9, [0, 0, 0, 0, 0, 0, 1, 1, 1]], dtype=np.int8))
    out = ifr_eval._append_unknown_items(ids, np.array([0, 1, 2]), excl, 6, 9).numpy()
    np.testing.assert_array_equal(out, [[5, 2, 7, 8], [1, 2, 3, 4], [-1, -1, -1, -1]])


def test_eval_csrs_are_memoised_per_split_and_notice_edits():
    """cached_eval_csrs: the same split objects give the same CSRs back; an in-place rating edit or another
    threshold rebuilds them."""
    from recommenders_amd import eval as ev
    rng = np.random.default_rng(3)
    tr = sp.random(50, 40, density=0.2, format="csr", random_state=np.random.RandomState(1), data_rvs=lambda n: rng.integers(1, 6, n).astype(float))
    te = sp.random(50, 40, density=0.1, format="csr", random_state=np.random.RandomState(2), data_rvs=lambda n: rng.integers(1, 6, n).astype(float))
    a = ev.cached_eval_csrs(tr, te, None, 3.0, 40, 50)
    b = ev.cached_eval_csrs(tr, te, None, 3.0, 40, 50)
    assert a[0] is b[0] and a[1] is b[1]
    ref = ev.build_eval_csrs(tr, te, None, 3.0, 40, 50)
    assert (a[0] != ref[0]).nnz == 0 and (a[1] != ref[1]).nnz == 0 and (a[2] == ref[2]).all()
    c = ev.cached_eval_csrs(tr, te, None, 4.0, 40, 50)
    assert c[0] is not a[0]
    te.data[0] = 5.0 if te.data[0] != 5.0 else 1.0                      # in-place edit of one rating
    d = ev.cached_eval_csrs(tr, te, None, 3.0, 40, 50)
    assert d[0] is not a[0]
    ref2 = ev.build_eval_csrs(tr, te, None, 3.0, 40, 50)
    assert (d[0] != ref2[0]).nnz == 0 and (d[1] != ref2[1]).nnz == 0


def test_memoised_csrs_are_uploaded_once_per_engine(toy):
    """A repeated ranking_eval on the same splits finds its exclusion / truth CSRs on the device; any other upload
    (rank_batch, a metric object, another evaluation) makes the next call send them again."""
    d = toy
    train_set, test_set, val_set = _datasets(d)
    eng = _engine(d, "bpr", test_set.num_users)
    fake = types.SimpleNamespace(is_fitted=True)
    kw = dict(rating_threshold=float(d["rating_threshold"]), _engine=eng)
    first, _ = ifr_eval.ranking_eval(fake, [M.NDCG(k=10)], train_set, test_set, val_set, **kw)
    assert eng.uploads == 2
    again, _ = ifr_eval.ranking_eval(fake, [M.NDCG(k=10)], train_set, test_set, val_set, **kw)
    assert eng.uploads == 2 and again == first
    eng.set_exclusion_csr(np.zeros(1, np.int64), np.zeros(0, np.int32))        # somebody else's exclusions
    third, _ = ifr_eval.ranking_eval(fake, [M.NDCG(k=10)], train_set, test_set, val_set, **kw)
    assert eng.uploads == 4 and third == first


def test_list_metrics_fall_back_to_sorted_lists_for_very_long_truth_rows(toy, monkeypatch):
    """Rows with more positives than ifr_eval_dense_metrics holds in LDS take ifr_rank_full + ifr_eval_ranked_metrics;
    both routes give the same values."""
    d = toy
    train_set, test_set, val_set = _datasets(d)
    mets = [M.NDCG(k=-1), M.Recall(k=50), M.MRR(), M.MAP()]
    kw = dict(rating_threshold=float(d["rating_threshold"]))
    a_avg, a_usr = ifr_eval.ranking_eval(types.SimpleNamespace(is_fitted=True), mets, train_set, test_set, val_set,
                                         _engine=_engine(d, "bpr", test_set.num_users), **kw)
    monkeypatch.setattr(ifr_eval, "MAX_DENSE_METRIC_POS", 0)
    eng = _engine(d, "bpr", test_set.num_users)
    eng.eval_dense_metrics = None                      # must not be called on this route
    b_avg, b_usr = ifr_eval.ranking_eval(types.SimpleNamespace(is_fitted=True), mets, train_set, test_set, val_set, _engine=eng, **kw)
    np.testing.assert_allclose(a_avg, b_avg, rtol=0, atol=1e-12)
    assert a_usr[3] == pytest.approx(b_usr[3], abs=1e-12)


def test_dense_row_batches_reuse_one_buffer_when_memory_is_short(toy):
    """the dense-row loop (AUC, k = -1, MRR / MAP) must hold ONE row buffer however many batches it runs,
    sized from the bytes a row of the chosen branch really keeps on the device."""
    d = toy
    train_set, test_set, val_set = _datasets(d)
    fake = types.SimpleNamespace(is_fitted=True)
    mets = [M.NDCG(k=-1), M.AUC(), M.MRR()]
    ref_eng = _engine(d, "bpr", test_set.num_users)
    ref_avg, ref_per = ifr_eval.ranking_eval(fake, mets, train_set, test_set, val_set,
                                             rating_threshold=float(d["rating_threshold"]), _engine=ref_eng)
    n_items = int(d["num_items"])
    eng = _engine(d, "bpr", test_set.num_users)
    eng.free_bytes = lambda: 2 * 7 * 4 * n_items            # half of it = seven fp32 rows
    avg, per = ifr_eval.ranking_eval(types.SimpleNamespace(is_fitted=True), mets, train_set, test_set, val_set,
                                     rating_threshold=float(d["rating_threshold"]), _engine=eng)
    assert eng.row_buffers == 1 and eng.largest_row_buffer == 7 and getattr(eng, "fresh_dense", 0) == 0
    assert avg == pytest.approx(ref_avg, abs=1e-12)
    assert per[0] == ref_per[0] and per[1] == ref_per[1]
    # the sorted-list branch keeps 28 bytes per (row, item): the same memory holds a quarter... one row here
    assert ifr_eval._rows_that_fit(eng, 1000, 28 * n_items) == 1
    assert ifr_eval._rows_that_fit(eng, 3, 4 * n_items) == 3


def test_user_results_reads_and_pickles_like_the_reference_dict():
    import pickle
    from recommenders_amd import UserResults
    r = UserResults([7, 2, 5], [0.7, 0.2, 0.5])                     # any order in, ascending inside
    assert list(r) == [2, 5, 7] and len(r) == 3 and 5 in r and 4 not in r
    assert r[7] == 0.7 and r == {2: 0.2, 5: 0.5, 7: 0.7} and r.keys() == [2, 5, 7]
    assert sum(r.values()) / len(r) == pytest.approx(1.4 / 3)       # the reference's aggregation (base_method.py:302-303)
    assert dict(r.items()) == r.to_dict()
    with pytest.raises(KeyError):
        r[3]
    back = pickle.loads(pickle.dumps(r))
    assert type(back) is dict and back == r.to_dict()
    assert r == pytest.approx({2: 0.2, 5: 0.5, 7: 0.7 + 1e-15})


def test_user_results_over_values_that_are_still_elsewhere():
    """ranking_eval hands `user_results` a reader for per-user values that are still on the device: nothing is copied
    until a value is read, then once; unsorted users materialise at once (the order has to be fixed)."""
    from recommenders_amd import UserResults
    calls = []

    def read():
        calls.append(1)
        return np.array([0.25, 0.5, 0.75])
    r = UserResults([3, 8, 9], read)
    assert len(r) == 3 and 8 in r and list(r) == [3, 8, 9] and not calls          # keys need no values
    assert r[8] == 0.5 and r.values() == [0.25, 0.5, 0.75] and r == {3: 0.25, 8: 0.5, 9: 0.75}
    assert len(calls) == 1
    r2 = UserResults([9, 3, 8], lambda: np.array([0.75, 0.25, 0.5]))
    assert r2 == r and r2.keys() == [3, 8, 9]


def test_device_eval_csrs_fall_back_to_the_host_build_without_an_engine(toy):
    """The object that stands for device-built candidate sets: its SciPy forms come from the engine that holds them or,
    when that engine is gone and the splits are SciPy matrices, from the host build -- the same rows either way."""
    d = toy
    train, test, val = _csr(d, "train"), _csr(d, "test"), _csr(d, "val")
    n_items, thr = int(d["num_items"]), float(d["rating_threshold"])
    n_rows = int(np.flatnonzero(np.diff(test.indptr)).max()) + 1
    obj = ifr_eval.DeviceEvalCsrs((train, test, val), thr, n_items, n_rows, 0)
    truth, excl, _ = ifr_eval.build_eval_csrs(train, test, val, thr, n_items, n_rows)
    for which, ref in (("truth", truth), ("exclusion", excl)):
        got = obj.host(which)
        assert (got != ref).nnz == 0 and got.shape == ref.shape
    # memoised per splits like the host CSRs, and keyed on content
    ifr_eval.clear_eval_cache()
    a = ifr_eval.cached_device_eval_csrs(train, test, val, thr, n_items, n_rows)
    assert ifr_eval.cached_device_eval_csrs(train, test, val, thr, n_items, n_rows) is a
    assert ifr_eval.cached_device_eval_csrs(train, test, val, thr + 1, n_items, n_rows) is not a
    ifr_eval.clear_eval_cache()
