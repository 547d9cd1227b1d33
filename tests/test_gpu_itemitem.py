"""GPU parity tests of the item-item path (EASE, RP3-beta)
through the C ABI, against the CPU oracle and against outputs of the reference itself.

Bars: top-K ids and fp32 scores bit-exact against the oracle's restatement of the documented
arithmetic (include/ifr.h: dense = fp32 fma chain in history order; CSR = 2^shift fixed point),
scores within 1e-5 relative of the reference's fp64 scores, per-user metrics within 1e-6 of the
reference's on users whose reference scores are separated.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as orc
from test_oracle import _csr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _workload(seed, n_users, n_items, mean_hist, density=None, row_nnz=None, weights="ratings", pen=False):
    """Synthetic item-item model: dense M (density None) or CSR M with ~row_nnz entries per row."""
    from recommenders_amd import synth
    rng = np.random.default_rng(seed)
    tr_ptr, tr_idx, te_ptr, te_idx = synth.interactions(rng, n_users, n_items, mean_hist, lo=0)
    if weights == "ratings":
        h_val = rng.integers(1, 6, len(tr_idx)).astype(np.float32)
    else:
        cnt = np.diff(tr_ptr)
        h_val = np.repeat(1.0 / np.maximum(cnt, 1), cnt).astype(np.float32)
    if row_nnz is None:
        M = np.maximum(0, rng.normal(0, 1e-2, (n_items, n_items))).astype(np.float32)
        M -= np.float32(0.002) * (rng.random((n_items, n_items)) < 0.1)      # a few negatives too
        np.fill_diagonal(M, 0.0)
        csr = None
    else:
        csr = sp.random(n_items, n_items, density=row_nnz / n_items, format="csr", dtype=np.float32,
                        random_state=np.random.RandomState(seed), data_rvs=lambda n: rng.random(n).astype(np.float32) * 1e-2)
        csr.sort_indices()
        M = None
    col_scale = (1.0 / rng.integers(1, 500, n_items) ** 0.7).astype(np.float32) if pen else None
    return dict(M=M, csr=csr, col_scale=col_scale, h_ptr=tr_ptr, h_idx=tr_idx, h_val=h_val, excl_ptr=tr_ptr,
                excl_idx=tr_idx, truth_ptr=te_ptr, truth_idx=te_idx, n_users=n_users, n_items=n_items)


def _scorer(w):
    from recommenders_amd import HipScorer
    sc = HipScorer(0)
    if w["M"] is not None:
        sc.set_item_matrix_dense(w["M"], w["col_scale"])
    else:
        c = w["csr"]
        sc.set_item_matrix_csr(c.indptr, c.indices, c.data, w["n_items"], w["col_scale"])
    sc.set_history_csr(w["h_ptr"], w["h_idx"], w["h_val"])
    sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
    sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
    return sc


def _oracle(w, sc, users, ks, kmax, want_auc=False):
    shift = sc.item_fixed_point_shift() if w["M"] is None else 0
    m_csr = None if w["csr"] is None else (w["csr"].indptr, w["csr"].indices, w["csr"].data)
    return orc.c_ranking_eval_item(w["M"], m_csr, shift, w["n_items"], w["col_scale"], w["h_ptr"], w["h_idx"], w["h_val"],
                                   users, w["excl_ptr"], w["excl_idx"], w["truth_ptr"], w["truth_idx"], ks, want_auc, kmax,
                                   n_threads=8)


@pytest.mark.parametrize("n_items,K,pen", [(3001, 10, False), (9000, 20, True), (17000, 10, False)])
def test_dense_topk_bit_exact_vs_oracle(torch, n_items, K, pen):
    w = _workload(40 + K, 333, n_items, 25, pen=pen)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    ids, scores = sc.rank_topk(users, K)
    o_ids, o_sc, _, _ = _oracle(w, sc, users, [K], K)
    np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
    np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)


@pytest.mark.parametrize("n_items,row_nnz,K,pen", [(3001, 40, 10, True), (10000, 120, 20, False), (20000, 300, 10, True), (9000, 12, 32, False),
                                                   # 32 < K <= 128: the workgroup kernel's long-list instantiation (K > 64: the short-list
                                                   # bound is split over the four waves); 12 entries per row: lists that end short
                                                   (10000, 120, 100, True), (3001, 40, 128, False), (20000, 300, 64, True), (9000, 12, 65, False),
                                                   (9000, 12, 33, True)])
def test_csr_topk_bit_exact_vs_oracle(torch, n_items, row_nnz, K, pen):
    w = _workload(50 + K, 301, n_items, 30, row_nnz=row_nnz, weights="prob", pen=pen)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    o_ids, o_sc, _, _ = _oracle(w, sc, users, [K], K)
    for wave in (2, 0):                         # the wave-per-user kernel forced (+ hand-over), and the workgroup-per-user kernel alone
        sc.set_option("csr_wave", wave)
        ids, scores = sc.rank_topk(users, K)
        np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
        np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
    # the library's shift follows the documented rule
    assert sc.item_fixed_point_shift() == orc.item_fixed_point_shift(w["csr"].data, w["h_ptr"], w["h_val"], w["csr"].indices,
                                                                   w["col_scale"])


@pytest.mark.parametrize("pass_entries,wave", [(0, 2), (0, 0), (32, 0)])
def test_csr_topk_pass_shapes(torch, pass_entries, wave):
    """The CSR top-K kernel over its paths: candidates ranked by counting and -- candidate buffer cut to max(32, K) entries,
    which the first windows of these users overflow -- by the exact block-wide argmax rounds; histories longer than a
    staging round (128 rows), exclusion rows beyond the LDS copy (256 entries), negative values (zero-sum columns
    enter the list) and users without history."""
    w = _workload(77, 160, 9000, 120, row_nnz=60, weights="ratings", pen=True)
    assert np.diff(w["h_ptr"]).max() > 256                                  # staging rounds and a global exclusion row
    w["csr"].data[::3] *= -1.0
    sc = _scorer(w)
    sc.set_option("csr_wave", wave)            # 2: one wave per user (forced), handing long histories / zero-sum lists to the workgroup kernel
    if pass_entries:
        sc.set_option("csr_cand_cap", pass_entries)
    users = np.r_[np.arange(w["n_users"]), [w["n_users"] + 5]].astype(np.int32)
    for K in (10, 32, 48, 100, 128):
        ids, scores = sc.rank_topk(users, K)
        o_ids, o_sc, _, _ = _oracle(w, sc, users, [K], K)
        np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
        np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
    # all-negative model: every touched column loses against the untouched (zero-score) ones
    w["csr"].data[:] = -np.abs(w["csr"].data)
    sc = _scorer(w)
    sc.set_option("csr_wave", wave)
    if pass_entries:
        sc.set_option("csr_cand_cap", pass_entries)
    for K in (10, 100):
        ids, scores = sc.rank_topk(users, K)
        o_ids, o_sc, _, _ = _oracle(w, sc, users, [K], K)
        np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
        np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)


def test_csr_equals_dense_form_within_fp32_rounding(torch):
    w = _workload(61, 120, 5000, 30, row_nnz=200, weights="prob", pen=True)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    sparse = sc.score_dense(users).cpu().numpy()
    wd = dict(w, M=w["csr"].toarray(), csr=None)
    dense = _scorer(wd).score_dense(users).cpu().numpy()
    np.testing.assert_allclose(sparse, dense, rtol=3e-6, atol=1e-12)


def test_dense_rows_auc_and_long_histories(torch):
    # histories longer than one staging round (512) and users without any history
    w = _workload(62, 90, 2500, 700, pen=True)
    assert np.diff(w["h_ptr"]).max() > 512
    w["h_ptr"] = w["h_ptr"].copy()
    sc = _scorer(w)
    users = np.r_[np.arange(w["n_users"]), [w["n_users"] + 3]].astype(np.int32)      # last one: beyond the rows
    rows = sc.score_dense(users)
    ids, scores = sc.rank_topk(users, 10)
    gathered = torch.gather(rows, 1, ids.long().clamp(min=0))
    assert torch.equal(gathered, scores)
    for u in (0, 7, 89):
        hi = w["h_idx"][w["h_ptr"][u]: w["h_ptr"][u + 1]]
        hv = w["h_val"][w["h_ptr"][u]: w["h_ptr"][u + 1]]
        np.testing.assert_array_equal(rows[u].cpu().numpy(), orc.c_score_item_dense(w["M"], hi, hv, w["col_scale"]))
    # a user without history scores 0 everywhere: all ties, ascending item index
    assert (rows[-1] == 0).all()
    np.testing.assert_array_equal(ids[-1].cpu().numpy(), np.arange(10))
    auc = sc.auc_dense(users[:-1], rows[:-1]).cpu().numpy()
    _, _, o_met, o_ev = _oracle(w, sc, users[:-1], [10], 10, want_auc=True)
    ev = o_ev.astype(bool)
    np.testing.assert_allclose(auc[ev], o_met[ev][:, 3], rtol=0, atol=1e-12)


def test_fewer_candidates_than_k_and_ties(torch):
    n_items = 40
    M = np.ones((n_items, n_items), np.float32)
    excl = [np.arange(40), np.arange(35), np.zeros(0, int), np.arange(5, 40), np.arange(1, 40)]
    ptr = np.cumsum([0] + [len(e) for e in excl]).astype(np.int64)
    w = dict(M=M, csr=None, col_scale=None, h_ptr=np.arange(6, dtype=np.int64) * 2,
             h_idx=np.tile(np.array([1, 2], np.int32), 5), h_val=np.ones(10, np.float32),
             excl_ptr=ptr, excl_idx=np.concatenate(excl).astype(np.int32),
             truth_ptr=np.zeros(6, np.int64), truth_idx=np.zeros(0, np.int32), n_users=5, n_items=n_items)
    for form in ("dense", "csr"):
        ww = w if form == "dense" else dict(w, M=None, csr=sp.csr_matrix(M))
        sc = _scorer(ww)
        ids, scores = sc.rank_topk(np.arange(5, dtype=np.int32), 10)
        ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
        assert (ids[0] == -1).all() and (ids[4] == np.r_[0, [-1] * 9]).all()
        np.testing.assert_array_equal(ids[1], np.r_[35:40, [-1] * 5])
        np.testing.assert_array_equal(ids[2], np.arange(10))                  # all scores equal 2.0
        assert (scores[2] == 2.0).all() and np.isneginf(scores[0]).all()
        if form == "csr":                        # a list longer than the catalogue: 40 columns, K = 70
            ids, scores = sc.rank_topk(np.arange(5, dtype=np.int32), 70)
            ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
            np.testing.assert_array_equal(ids[2], np.r_[0:40, [-1] * 30])
            np.testing.assert_array_equal(ids[1], np.r_[35:40, [-1] * 65])
            assert (ids[0] == -1).all() and (scores[2][:40] == 2.0).all() and np.isneginf(scores[2][40:]).all()
            # the C entry point itself refuses K > 128 (the facade serves those from dense rows)
            u1 = torch.zeros(1, dtype=torch.int32, device="cuda")
            oi, os_ = torch.empty((1, 129), dtype=torch.int32, device="cuda"), torch.empty((1, 129), dtype=torch.float32, device="cuda")
            assert sc._lib.ifr_rank_topk(sc._h, u1.data_ptr(), 1, 129, oi.data_ptr(), os_.data_ptr(), None) != 0
            big, _ = sc.rank_topk(np.arange(5, dtype=np.int32), 129)
            np.testing.assert_array_equal(big.cpu().numpy()[2], np.r_[0:40, [-1] * 89])


def test_bad_inputs_are_rejected(torch):
    from recommenders_amd import HipScorer, IfrError
    sc = HipScorer(0)
    with pytest.raises(IfrError):
        sc.set_history_csr(np.zeros(2, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32))   # no matrix yet
    with pytest.raises(IfrError):                                                                   # unsorted columns
        sc.set_item_matrix_csr(np.array([0, 2, 2, 2], np.int64), np.array([2, 1], np.int32), np.ones(2, np.float32), 3)
    with pytest.raises(IfrError):                                                                   # column out of range
        sc.set_item_matrix_csr(np.array([0, 1, 1, 1], np.int64), np.array([3], np.int32), np.ones(1, np.float32), 3)
    sc.set_item_matrix_dense(np.eye(3, dtype=np.float32))
    with pytest.raises(IfrError):                                                                   # history item out of range
        sc.set_history_csr(np.array([0, 1], np.int64), np.array([7], np.int32), np.ones(1, np.float32))
    with pytest.raises(IfrError):                                                                   # histories missing
        sc.rank_topk(np.zeros(1, np.int32), 2)


# ------------------------------------------------------------------ the facade against the reference's own outputs
def _sets(d):
    from recommenders_amd.data import Dataset
    mk = lambda m: Dataset.from_csr(m)
    return mk(_csr(d, "train")), mk(_csr(d, "test")), mk(_csr(d, "val"))


def _metrics(d):
    from recommenders_amd import metrics as M
    out = []
    for n in d["metric_names"]:
        name, k = str(n).split("@")
        out.append(M.AUC() if name == "auc" else {"ndcg": M.NDCG, "recall": M.Recall, "precision": M.Precision}[name](k=int(k)))
    return out


def _facade(d, name, train_set, form):
    from recommenders_amd.models import EASE, RP3_Beta
    if name == "ease":
        return EASE(lamb=50, trainable=False, init_params={"B": d["ease_B"]},
                    precision="f64" if form == "dense_f64" else "f32").fit(train_set)
    # the item-item arrays of the oracle's restatement of graph_recommender.py:24-55 (pinned by the reference's scores)
    P_ui, W, deg = orc.rp3_item_item(sp.csr_matrix((_csr(d, "train") > 0).astype(np.float64)))
    return RP3_Beta(beta=float(d["rp3_beta"]), item_matrix="dense" if form == "dense_f64" else form,
                    precision="f64" if form == "dense_f64" else "f32",
                    init_params={"P_ui": P_ui, "W": W, "item_degrees": deg}).fit(train_set)


@pytest.mark.parametrize("name,form", [("ease", "dense"), ("ease", "dense_f64"), ("rp3", "dense"), ("rp3", "dense_f64"), ("rp3", "csr"),
                                       ("rp3", "device")])
def test_facade_matches_reference(torch, gap, name, form):
    """EASE / RP3-beta against the reference itself on EVERY evaluated user of the gap-guaranteed fixture (no tie
    filter): top-20 ids equal, per-user NDCG / Recall / Precision / AUC equal, every mean within 1e-6 -- the
    north-star bar.  "device" = RP3_Beta.fit on the GPU (ifr_fit_rp3beta) instead of uploaded host arrays."""
    from recommenders_amd import ScoreException, ranking_eval
    from recommenders_amd.data import Dataset
    from recommenders_amd.models import RP3_Beta
    d = gap
    train_set, test_set, val_set = (Dataset.from_csr(_csr(d, "train")), Dataset.from_csr(_csr(d, f"{name}__test")),
                                    Dataset.from_csr(_csr(d, "val")))
    model = RP3_Beta(beta=float(d["rp3_beta"])).fit(train_set) if form == "device" else _facade(d, name, train_set, form)
    # Recommender.score: fp64 like the reference, EASE with its (1, I) shape
    for u, ref in zip(d[f"{name}__score_users"], d[f"{name}__score_vecs"]):
        s = model.score(int(u))
        assert s.dtype == np.float64 and s.shape == ((1, len(ref)) if name == "ease" else ref.shape)
        s = s.ravel()
        ok = np.isfinite(ref)
        scale = np.maximum(np.abs(ref), np.abs(ref[ok]).max() * 1e-3)
        # fp32 sums: 1e-5 of the score; the fp64 option: the reference's float64 values to 1e-13 (a different
        # summation order of the same float64 terms)
        assert np.all(np.abs(s[ok] - ref[ok]) <= (1e-13 if (form, name) == ("dense_f64", "ease") else 1e-5) * scale[ok])
    with pytest.raises(ScoreException):
        model.score(model.num_users + 3)
    # ranking_eval
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
    # top-20 ids of every user (K = 20 is within the fused item-item kernel's 32)
    ids, scores = model.rank_batch(users, 20, exclude_csr=model._hip_cache_owner.excl_global())
    np.testing.assert_array_equal(ids, d[f"{name}__ranked_head"][:, :20])
    np.testing.assert_allclose(scores, d[f"{name}__ranked_head_scores"][:, :20], rtol=1e-5, atol=1e-12)
    # rank(): the reference's head, scores aligned with the candidates
    for r in range(0, len(users), 11):
        u = int(users[r])
        ranked = model.ranked_items[u]
        np.testing.assert_array_equal(ranked[:20], d[f"{name}__ranked_head"][r][:20])
        assert len(ranked) == d[f"{name}__n_cand"][r]


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_dense_item_matrix_lists_and_rows_in_one_launch(torch, precision):
    """ifr_rank_topk_dense / ifr_rank_topk_dense_f64 for a dense item matrix: the lists of ifr_rank_topk and the rows of
    ifr_score_dense / ifr_score_dense_f64 from ONE pass over M (what ranking_eval uses for top-K cut-offs + AUC)."""
    from recommenders_amd import HipScorer
    w = _workload(91, 257, 9001, 22, pen=True)
    sc = HipScorer(0)
    sc.set_item_matrix_dense(w["M"].astype(np.float64) if precision == "f64" else w["M"], w["col_scale"], precision=precision)
    sc.set_history_csr(w["h_ptr"], w["h_idx"], w["h_val"])
    sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
    sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
    users = np.r_[np.arange(w["n_users"]), [w["n_users"] + 2]].astype(np.int32)[::-1].copy()
    assert sc.can_rank_topk_dense(32) and not sc.can_rank_topk_dense(33)
    for K in (10, 32):
        ids, scores = sc.rank_topk(users, K)
        ids2, scores2, dense = sc.rank_topk_dense(users, K)
        np.testing.assert_array_equal(ids2.cpu().numpy(), ids.cpu().numpy())
        np.testing.assert_array_equal(scores2.cpu().numpy(), scores.cpu().numpy())
        rows = sc.score_rows(users)
        assert dense.dtype == rows.dtype == (torch.float64 if precision == "f64" else torch.float32)
        np.testing.assert_array_equal(dense.cpu().numpy(), rows[:, : w["n_items"]].cpu().numpy())
        np.testing.assert_array_equal(sc.auc_dense(users, dense).cpu().numpy(), sc.auc_dense(users, rows).cpu().numpy())
    sc.close()


@pytest.mark.parametrize("form", ["dense", "csr"])
def test_item_item_cutoff_beyond_32_through_partial_ranking(torch, form):
    """rank_topk(K = 50) of an item-item model: dense rows (ifr_score_dense) + ifr_select_topk, against the oracle."""
    if form == "dense":
        w = _workload(91, 150, 3001, 25, pen=True)
    else:
        w = _workload(92, 150, 6000, 30, row_nnz=80, weights="prob", pen=True)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    ids, scores = sc.rank_topk(users, 50)
    o_ids, o_sc, _, _ = _oracle(w, sc, users, [50], 50)
    np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
    np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)


# ------------------------------------------------------------------ the fp64 option of the dense item matrix
def _f64_workload(seed, n_users, n_items, mean_hist, pen):
    w = _workload(seed, n_users, n_items, mean_hist, pen=pen)
    rng = np.random.default_rng(seed + 1000)
    w["M64"] = w["M"].astype(np.float64) * (1.0 + 1e-9 * rng.standard_normal(w["M"].shape))   # not fp32-representable
    return w


def _f64_scorer(w):
    from recommenders_amd import HipScorer
    sc = HipScorer(0)
    sc.set_item_matrix_dense(w["M64"], w["col_scale"], precision="f64")
    sc.set_history_csr(w["h_ptr"], w["h_idx"], w["h_val"])
    sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
    sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
    return sc


@pytest.mark.parametrize("n_items,K,pen", [(3001, 10, True), (9000, 20, False)])
def test_dense_f64_bit_exact_vs_oracle(torch, n_items, K, pen):
    """ifr_set_item_matrix_dense_f64: float64 rows bit-exact against the oracle's fp64 fma chain, within 1e-14
    (relative to sum |x m|) of NumPy's float64 product; fp32 rows = those sums rounded once; top-K lists = the
    canonical ranking of the float64 sums."""
    w = _f64_workload(70 + K, 200, n_items, 30, pen)
    sc = _f64_scorer(w)
    users = np.r_[np.arange(w["n_users"]), [w["n_users"] + 2]].astype(np.int32)
    rows64 = sc.score_dense_f64(users).cpu().numpy()
    rows32 = sc.score_dense(users).cpu().numpy()
    assert rows64.dtype == np.float64 and (rows64[-1] == 0).all()
    np.testing.assert_array_equal(rows32, rows64.astype(np.float32))
    X = sp.csr_matrix((w["h_val"].astype(np.float64), w["h_idx"], w["h_ptr"]), shape=(w["n_users"], n_items))
    ref = X.dot(w["M64"]) * (1.0 if w["col_scale"] is None else w["col_scale"].astype(np.float64))
    mag = abs(X).dot(np.abs(w["M64"])) * (1.0 if w["col_scale"] is None else w["col_scale"].astype(np.float64))
    assert np.all(np.abs(rows64[:-1] - ref) <= 1e-14 * mag + 1e-300)
    for u in (0, 57, 199):
        hi = w["h_idx"][w["h_ptr"][u]: w["h_ptr"][u + 1]]
        hv = w["h_val"][w["h_ptr"][u]: w["h_ptr"][u + 1]]
        o64, o32 = orc.c_score_item_dense_f64(w["M64"], hi, hv, w["col_scale"])
        np.testing.assert_array_equal(rows64[u], o64)
        np.testing.assert_array_equal(rows32[u], o32)
    # the fused top-K and the partial ranking beyond it order the FLOAT64 sums (tests/test_gpu_f64_ranking.py holds the
    # cases where that differs from ranking the fp32-rounded sums); the scores they return are those sums rounded once
    ids, scores = sc.rank_topk(users, K)
    ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
    ids50, sc50 = (t.cpu().numpy() for t in sc.rank_topk(users[:40], 50))
    for i, u in enumerate(users[:-1]):
        ex = w["excl_idx"][w["excl_ptr"][u]: w["excl_ptr"][u + 1]]
        cand = np.setdiff1d(np.arange(n_items), ex)
        want = orc.canonical_rank(cand, rows64[i][cand], max(K, 50))
        np.testing.assert_array_equal(ids[i], want[:K])
        np.testing.assert_array_equal(scores[i], rows64[i][want[:K]].astype(np.float32))
        if i < 40:
            np.testing.assert_array_equal(ids50[i], want[:50])
            np.testing.assert_array_equal(sc50[i], rows64[i][want[:50]].astype(np.float32))
    from recommenders_amd import IfrError
    with pytest.raises(IfrError):
        _scorer(w).score_dense_f64(users)                     # an fp32 matrix has no float64 rows


def test_fit_ease_f64_option(torch, toy):
    """ifr_fit_ease with option item_f64: B stays float64 on the device; scores agree with the host float64 product
    of the B it returns to 1e-13, and with the default build's B."""
    from recommenders_amd import HipScorer
    X = _csr(toy, "train").tocsr().astype(np.float32)
    X.sort_indices()
    n_items = X.shape[1]
    sc = HipScorer(0)
    B = sc.fit_ease(X.indptr.astype(np.int64), X.indices.astype(np.int32), X.data, n_items, 50.0, True, want_B=True,
                    precision="f64")
    sc.set_history_csr(X.indptr.astype(np.int64), X.indices.astype(np.int32), X.data)
    users = np.arange(64, dtype=np.int32)
    rows = sc.score_dense_f64(users).cpu().numpy()
    ref = X[:64].astype(np.float64).dot(B)
    mag = abs(X[:64]).astype(np.float64).dot(np.abs(B))
    assert np.all(np.abs(rows - ref) <= 1e-13 * mag + 1e-300)
    sc32 = HipScorer(0)
    B32 = sc32.fit_ease(X.indptr.astype(np.int64), X.indices.astype(np.int32), X.data, n_items, 50.0, True, want_B=True)
    np.testing.assert_array_equal(B, B32)                     # the build itself does not depend on the option
