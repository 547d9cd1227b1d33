"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bars (BASELINE.json north_star): top-K ids bit-exact, fp32 scores within 1e-5 relative
(here: bit-for-bit against the oracle run in the kernel's documented fma-chain order, and
within 1e-5 of the k-ascending / BLAS-order scores), metrics within 1e-6.
"""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _workload(seed, n_users, n_items, k, bias, mean=40):
    from recommenders_amd import synth
    rng = np.random.default_rng(seed)
    U, V, ub, ib = synth.factor_model(rng, n_users, n_items, k, bias)
    tr_ptr, tr_idx, te_ptr, te_idx = synth.interactions(rng, n_users, n_items, mean, lo=0)
    return dict(U=U, V=V, ubase=ub, ibase=ib, excl_ptr=tr_ptr, excl_idx=tr_idx, truth_ptr=te_ptr, truth_idx=te_idx,
                n_users=n_users, n_items=n_items, k=k)


def _scorer(w, excl=True, truth=True):
    from recommenders_amd import HipScorer
    sc = HipScorer(0)
    sc.set_factors(w["U"], w["V"], w["ubase"], w["ibase"])
    if excl:
        sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
    if truth:
        sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
    return sc


def _oracle(w, users, ks, kmax, kperm, want_auc=False):
    kp = None if kperm is None else kperm[kperm < w["k"]]
    return orc.c_ranking_eval_factor(w["U"], w["V"], w["ibase"], w["ubase"], users, w["excl_ptr"], w["excl_idx"],
                                     w["truth_ptr"], w["truth_idx"], ks, want_auc, kmax, kperm=kp, n_threads=8)


@pytest.mark.parametrize("k,bias,K", [(128, False, 10), (50, True, 10), (10, True, 5), (128, True, 20), (64, False, 32),
                                      (200, False, 10), (256, True, 20),    # 200 = the reference's WMF default k
                                      (150, True, 10), (180, False, 20),    # the 160 / 192 / 224-wide instantiations
                                      (80, True, 10), (96, False, 32)])     # 96-wide
def test_topk_bit_exact_vs_oracle(torch, k, bias, K):
    w = _workload(1000 + k + K, 777, 3001, k, bias)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    ids, scores = sc.rank_topk(users, K)
    ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
    o_ids, o_sc, _, _ = _oracle(w, users, [K], K, sc.chain_order())
    np.testing.assert_array_equal(ids, o_ids)
    np.testing.assert_array_equal(scores, o_sc)            # bit-for-bit in the documented chain order
    # and within 1e-5 (relative to |u|.|v|) of the k-ascending order the reference's BLAS may use
    r_ids, r_sc, _, _ = _oracle(w, users, [K], K, None)
    scale = np.linalg.norm(w["U"], axis=1).max() * np.linalg.norm(w["V"], axis=1).max()
    same = ids == r_ids
    assert same.mean() > 0.999
    assert np.abs(scores[same] - r_sc[same]).max() <= 1e-5 * scale


def test_chunking_does_not_change_results(torch):
    w = _workload(5, 300, 5000, 128, True)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    ref = None
    for chunks in (1, 3, 9):
        sc.set_option("chunks", chunks)
        ids, scores = sc.rank_topk(users, 10)
        got = (ids.cpu().numpy(), scores.cpu().numpy())
        if ref is None:
            ref = got
        np.testing.assert_array_equal(got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])


def test_two_region_grid_does_not_change_results(torch):
    """More user tiles than resident workgroup slots: the last, partial round (region B of the grid) sweeps shorter
    item chunks.  Same lists with the split off, and bit-exact against the oracle on both sides of the border."""
    n_users = 128 * 552 + 17                                     # 553 user tiles on 512 slots; the last one ragged
    w = _workload(6, n_users, 2048, 32, True, mean=12)
    sc = _scorer(w)
    sc.set_option("target_blocks", 1)                            # one item chunk in region A, so that region B's 4 differ
    users = np.arange(n_users, dtype=np.int32)
    ids, scores = sc.rank_topk(users, 10)
    sc.set_option("tail_split", 0)
    ids1, scores1 = sc.rank_topk(users, 10)
    assert torch.equal(ids, ids1) and torch.equal(scores, scores1)
    sample = np.r_[0:300, 128 * 511 - 100:128 * 513 + 100, n_users - 300:n_users].astype(np.int32)
    o_ids, o_sc, _, _ = _oracle(w, sample, [10], 10, sc.chain_order())
    np.testing.assert_array_equal(ids.cpu().numpy()[sample], o_ids)
    np.testing.assert_array_equal(scores.cpu().numpy()[sample], o_sc)


def test_user_subset_order_and_ragged_batch(torch):
    w = _workload(6, 500, 2000, 64, False)
    sc = _scorer(w)
    users = np.random.default_rng(0).permutation(w["n_users"])[:131].astype(np.int32)   # not a multiple of 128
    ids, scores = sc.rank_topk(users, 10)
    o_ids, o_sc, _, _ = _oracle(w, users, [10], 10, sc.chain_order())
    np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
    np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
    one = sc.rank_topk(users[:1], 10)[0].cpu().numpy()
    np.testing.assert_array_equal(one, o_ids[:1])


def test_ties_break_by_item_index(torch):
    # constant factors: every score equal -> the canonical order is ascending item index
    n_users, n_items, k = 130, 1000, 32
    w = dict(U=np.ones((n_users, k), np.float32), V=np.ones((n_items, k), np.float32), ubase=None, ibase=None,
             excl_ptr=np.arange(n_users + 1, dtype=np.int64) * 2,
             excl_idx=np.tile(np.array([0, 3], np.int32), n_users),
             truth_ptr=np.zeros(n_users + 1, np.int64), truth_idx=np.zeros(0, np.int32), n_users=n_users,
             n_items=n_items, k=k)
    sc = _scorer(w, truth=False)
    sc.set_option("chunks", 4)
    ids, scores = sc.rank_topk(np.arange(n_users, dtype=np.int32), 10)
    expect = np.array([1, 2, 4, 5, 6, 7, 8, 9, 10, 11], np.int32)
    np.testing.assert_array_equal(ids.cpu().numpy(), np.tile(expect, (n_users, 1)))
    assert (scores.cpu().numpy() == 32.0).all()


def test_fewer_candidates_than_k(torch):
    n_users, n_items, k = 5, 40, 16
    rng = np.random.default_rng(3)
    excl = [np.arange(40), np.arange(35), np.zeros(0, int), np.arange(5, 40), np.arange(1, 40)]
    ptr = np.cumsum([0] + [len(e) for e in excl]).astype(np.int64)
    w = dict(U=rng.normal(size=(n_users, k)).astype(np.float32), V=rng.normal(size=(n_items, k)).astype(np.float32),
             ubase=None, ibase=None, excl_ptr=ptr, excl_idx=np.concatenate(excl).astype(np.int32),
             truth_ptr=np.zeros(n_users + 1, np.int64), truth_idx=np.zeros(0, np.int32), n_users=n_users,
             n_items=n_items, k=k)
    sc = _scorer(w, truth=False)
    users = np.arange(n_users, dtype=np.int32)
    ids, scores = sc.rank_topk(users, 10)
    o_ids, o_sc, _, _ = _oracle(w, users, [10], 10, sc.chain_order())
    np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
    np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
    assert (ids.cpu().numpy()[0] == -1).all() and (ids.cpu().numpy()[4] == np.r_[0, [-1] * 9]).all()


def test_metrics_vs_oracle(torch):
    w = _workload(8, 1000, 2500, 128, False)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    ks = [10, 5, 20]
    ids, _ = sc.rank_topk(users, 20)
    per_user, evaluated, sums = sc.eval_topk_metrics(users, ids, ks)
    per_user, evaluated, sums = per_user.cpu().numpy(), evaluated.cpu().numpy(), sums.cpu().numpy()
    _, _, o_met, o_ev = _oracle(w, users, ks, 20, sc.chain_order())
    np.testing.assert_array_equal(evaluated, o_ev)
    assert 0 < evaluated.sum() < len(users) or evaluated.all()
    ev = evaluated.astype(bool)
    cols = [6 * q + c for q in range(3) for c in range(3)]            # ndcg, recall, precision of each cut-off
    np.testing.assert_allclose(per_user[ev][:, cols], o_met[ev][:, :9], rtol=0, atol=1e-12)
    assert np.isnan(per_user[~ev]).all()
    np.testing.assert_allclose(sums[cols], o_met[ev][:, :9].sum(0), rtol=1e-12)
    assert sums[-1] == ev.sum()
    # aggregate (mean over evaluated users, base_method.py:302-303) within 1e-6
    np.testing.assert_allclose(sums[cols] / sums[-1], o_met[ev][:, :9].mean(0), atol=1e-6, rtol=0)
    # hit ratio / f1 / ncrr against the NumPy restatement of the reference classes
    ids_h = ids.cpu().numpy()
    for u in np.flatnonzero(ev)[:200]:
        pos = w["truth_idx"][w["truth_ptr"][u]: w["truth_ptr"][u + 1]]
        for q, k in enumerate(ks):
            assert per_user[u, 6 * q + 3] == orc.hit_ratio(pos, ids_h[u], k)
            assert per_user[u, 6 * q + 4] == pytest.approx(orc.f_measure(pos, ids_h[u], k), abs=1e-12)
            assert per_user[u, 6 * q + 5] == pytest.approx(orc.ncrr(pos, ids_h[u], k), abs=1e-12)


@pytest.mark.parametrize("k", [50, 200])
def test_dense_scores_match_fused_and_auc(torch, k):
    w = _workload(9, 260, 1500, k, True)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    dense = sc.score_dense(users)
    ids, scores = sc.rank_topk(users, 10)
    gathered = torch.gather(dense, 1, ids.long().clamp(min=0))
    assert torch.equal(gathered, scores)
    # dense rows bit-for-bit equal to the oracle in chain order
    perm = sc.chain_order()
    perm = perm[perm < w["k"]]
    for u in (0, 17, 259):
        ref = orc.c_score_factor(w["U"][u], w["V"], w["ibase"], float(w["ubase"][u]), perm)
        np.testing.assert_array_equal(dense[u].cpu().numpy(), ref)
    auc = sc.auc_dense(users, dense).cpu().numpy()
    _, _, o_met, o_ev = _oracle(w, users, [10], 10, sc.chain_order(), want_auc=True)
    ev = o_ev.astype(bool)
    np.testing.assert_allclose(auc[ev], o_met[ev][:, 3], rtol=0, atol=1e-12)
    assert np.isnan(auc[~ev]).all()


def test_rank_full_vs_oracle(torch):
    w = _workload(10, 70, 900, 32, True)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    dense = sc.score_dense(users)
    ids, n_cand = sc.rank_full(users, dense)
    ids, n_cand, dn = ids.cpu().numpy(), n_cand.cpu().numpy(), dense.cpu().numpy()
    for u in range(w["n_users"]):
        ex = w["excl_idx"][w["excl_ptr"][u]: w["excl_ptr"][u + 1]]
        cand = np.setdiff1d(np.arange(w["n_items"]), ex)
        expect = orc.canonical_rank(cand, dn[u][cand])
        assert n_cand[u] == len(cand)
        np.testing.assert_array_equal(ids[u, : len(cand)], expect)
        assert (ids[u, len(cand):] == -1).all()


def test_ranked_list_metrics_and_rank_items(torch):
    w = _workload(11, 90, 700, 32, True, mean=30)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    dense = sc.score_dense(users)
    ranked, n_cand = sc.rank_full(users, dense)
    ks = [-1, 10, 50]
    per_user, evaluated, sums = sc.eval_ranked_metrics(users, ranked, n_cand, ks)
    per_user, evaluated = per_user.cpu().numpy(), evaluated.cpu().numpy().astype(bool)
    rk, nc, dn = ranked.cpu().numpy(), n_cand.cpu().numpy(), dense.cpu().numpy()
    assert evaluated.any()
    for u in range(w["n_users"]):
        pos = w["truth_idx"][w["truth_ptr"][u]: w["truth_ptr"][u + 1]]
        if len(pos) == 0:
            assert not evaluated[u] and np.isnan(per_user[u]).all()
            continue
        lst = rk[u, : nc[u]]
        for q, k in enumerate(ks):
            exp = [orc.ndcg(pos, lst, k), orc.recall(pos, lst, k), orc.precision(pos, lst, k),
                   orc.hit_ratio(pos, lst, k), orc.f_measure(pos, lst, k), orc.ncrr(pos, lst, k)]
            np.testing.assert_allclose(per_user[u, 6 * q: 6 * q + 6], exp, rtol=0, atol=1e-12)
        assert per_user[u, 18] == pytest.approx(orc.mrr(pos, lst), abs=1e-15)
        assert per_user[u, 19] == pytest.approx(orc.average_precision(lst, dn[u][lst].astype(np.float64), pos), abs=1e-12)
    s = sums.cpu().numpy()
    np.testing.assert_allclose(s[:-1], per_user[evaluated].sum(0), rtol=1e-12)
    # arbitrary candidate list of one user (Recommender.rank with item_indices)
    cand = np.random.default_rng(1).permutation(w["n_items"])[:333].astype(np.int32)
    got = sc.rank_items(dense[7], torch.from_numpy(cand).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, orc.canonical_rank(cand, dn[7][cand]))


@pytest.mark.parametrize("model", ["bpr", "mf", "wmf"])
def test_golden_reference_fixture(torch, toy, model):
    """Per-user NDCG/Recall/Precision@K from the HIP path against the values the reference
    itself produced (tests/golden/toy_all_models.npz)."""
    from test_oracle import _csr
    d = toy
    nu, ni = int(d["num_users"]), int(d["num_items"])
    thr = float(d["rating_threshold"])
    train, test, val = _csr(d, "train"), _csr(d, "test"), _csr(d, "val")
    excl_ptr, excl_idx, tr_ptr, tr_idx = [0], [], [0], []
    for u in range(nu):
        pos = orc.pos_items(test, u, thr)
        pos = pos[pos < ni]
        ex = np.union1d(orc.pos_items(train, u, thr), orc.pos_items(val, u, thr))
        ex = np.setdiff1d(ex[ex < ni], pos)
        excl_idx += ex.tolist(); excl_ptr.append(len(excl_idx))
        tr_idx += pos.tolist(); tr_ptr.append(len(tr_idx))
    if model == "bpr":
        U, V, ub, ib = d["bpr_U"], d["bpr_V"], None, d["bpr_Bi"]
    elif model == "mf":
        U, V, ub, ib = d["mf_U"], d["mf_V"], d["mf_Bu"], (np.float32(d["mf_mu"]) + d["mf_Bi"]).astype(np.float32)
    else:
        U, V, ub, ib = d["wmf_U"], d["wmf_V"], None, None
    from recommenders_amd import HipScorer
    sc = HipScorer(0)
    sc.set_factors(U, V, ub, ib)
    sc.set_exclusion_csr(np.asarray(excl_ptr, np.int64), np.asarray(excl_idx, np.int32))
    sc.set_truth_csr(np.asarray(tr_ptr, np.int64), np.asarray(tr_idx, np.int32))
    users = d[f"{model}__users"]
    ids, scores = sc.rank_topk(users, 20)
    per_user, evaluated, sums = sc.eval_topk_metrics(users, ids, [10, 5, 20])
    per_user = per_user.cpu().numpy()
    assert evaluated.cpu().numpy().all()
    names = [str(x) for x in d["metric_names"]]
    col = {"ndcg@10": 0, "recall@10": 1, "precision@10": 2, "ndcg@5": 6, "recall@20": 13}
    ref = d[f"{model}__per_user"]
    ref_head, ref_hs = d[f"{model}__ranked_head"], d[f"{model}__ranked_head_scores"]
    ids = ids.cpu().numpy()
    # the reference's head (top-20) wherever its own scores are separated by > 1e-5 relative
    gaps = np.abs(np.diff(ref_hs, axis=1))
    tie_free = (gaps > 1e-5 * np.maximum(1e-3, np.abs(ref_hs[:, :-1]))).all(axis=1)
    assert tie_free.mean() > 0.8
    np.testing.assert_array_equal(ids[tie_free][:, :19], ref_head[tie_free][:, :19])
    # fp32 scores within 1e-5 relative of the reference's
    s = scores.cpu().numpy()[tie_free][:, :19]
    r = ref_hs[tie_free][:, :19]
    assert np.abs(s - r).max() <= 1e-5 * max(1.0, np.abs(r).max())
    for j, n in enumerate(names):
        if n in col:
            np.testing.assert_allclose(per_user[tie_free][:, col[n]], ref[tie_free][:, j], atol=1e-6, rtol=0)
            # aggregate within 1e-6 of the reference's mean over the same users
            assert abs(per_user[tie_free][:, col[n]].mean() - ref[tie_free][:, j].mean()) < 1e-6
    # AUC through the dense path
    dense = sc.score_dense(users)
    auc = sc.auc_dense(users, dense).cpu().numpy()
    j = names.index("auc@-1")
    assert np.abs(auc - ref[:, j]).max() < 1e-4          # a near-tie may flip single pairs
    assert abs(auc.mean() - ref[:, j].mean()) < 1e-6


@pytest.mark.parametrize("K", [33, 100, 1000])
def test_partial_ranking_beyond_the_fused_cutoff(torch, K):
    """ifr_select_topk (Recommender.rank(k != -1), recommender.py:644-649) = the head of the full ranking: against
    ifr_rank_full on the same dense rows (ids and scores bit for bit), against the oracle's lists, and through
    rank_topk, which switches to it for K > 32.  Dense scores equal the fused kernel's, so K <= 32 heads agree too."""
    w = _workload(77 + K, 300, 3001, 64, True)
    sc = _scorer(w)
    users = np.r_[np.arange(w["n_users"]), [3, 3]].astype(np.int32)
    dense = sc.score_dense(users)
    ids, scores = sc.select_topk(users, dense, K)
    full, n_cand = sc.rank_full(users, dense)
    assert (n_cand >= K).all()
    assert torch.equal(ids, full[:, :K])
    assert torch.equal(scores, torch.gather(dense, 1, ids.long()))
    o_ids, _, _, _ = _oracle(w, users, [K], K, sc.chain_order())
    np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
    ids2, scores2 = sc.rank_topk(users, K, dense_batch_pairs=100 * 3001)          # several dense batches
    assert torch.equal(ids2, ids) and torch.equal(scores2, scores)
    fused, _ = sc.rank_topk(users, 32)
    assert torch.equal(fused, ids[:, :32])


def test_partial_ranking_ties_short_rows_and_nan(torch):
    torch_ = torch
    from recommenders_amd import HipScorer
    n_items, k = 700, 8
    U = np.zeros((6, k), np.float32); V = np.zeros((n_items, k), np.float32)
    U[:, 0] = 1.0
    V[:, 0] = np.repeat(np.arange(7, 0, -1), 100)                  # seven plateaus of 100 equal scores
    excl = [np.zeros(0, int), np.arange(0, 650), np.arange(n_items), np.arange(95, 105), np.arange(1, n_items, 2), np.zeros(0, int)]
    ptr = np.cumsum([0] + [len(e) for e in excl]).astype(np.int64)
    sc = HipScorer(0)
    sc.set_factors(U, V, None, None)
    sc.set_exclusion_csr(ptr, np.concatenate(excl).astype(np.int32))
    users = np.arange(6, dtype=np.int32)
    dense = sc.score_dense(users)
    dense[5, 300] = float("nan")                                     # NaN ranks first, like argsort()[::-1]
    for K in (40, 150, 1024):
        ids, scores = sc.select_topk(users, dense, K)
        full, n_cand = sc.rank_full(users, dense)
        want = full[:, :K] if full.shape[1] >= K else torch_.nn.functional.pad(full, (0, K - full.shape[1]), value=-1)
        assert torch_.equal(ids, want)
        ok = ids >= 0
        got = scores[ok]
        ref = torch_.gather(dense, 1, ids.clamp(min=0).long())[ok]
        assert torch_.equal(torch_.nan_to_num(got, nan=123.0), torch_.nan_to_num(ref, nan=123.0))
        assert torch_.isneginf(scores[~ok]).all()
    assert (ids[2] == -1).all() and ids[5, 0] == 300 and (ids[1, :50].cpu().numpy() == np.arange(650, 700)).all()


def test_partial_ranking_long_plateaus(torch):
    """ifr_select_topk when more items tie with the K-th candidate than its tie buffer holds (1 024): the ordered sweep.
    Constant rows and a plateau of 3 000 equal scores behind a short head; exclusions inside the plateau."""
    from recommenders_amd import HipScorer
    n_items, k = 5000, 8
    U = np.zeros((4, k), np.float32); V = np.zeros((n_items, k), np.float32)
    U[:, 0] = 1.0
    V[:, 0] = 1.0
    V[4000:4020, 0] = 2.0                                            # a head of 20 better items at high indices
    excl = [np.zeros(0, int), np.arange(0, 2000, 3), np.arange(4000, 4010), np.arange(0, 4990)]
    ptr = np.cumsum([0] + [len(e) for e in excl]).astype(np.int64)
    sc = HipScorer(0)
    sc.set_factors(U, V, None, None)
    sc.set_exclusion_csr(ptr, np.concatenate(excl).astype(np.int32))
    users = np.arange(4, dtype=np.int32)
    dense = sc.score_dense(users)
    dense[0, :] = 0.5                                                # one fully constant row
    full, n_cand = sc.rank_full(users, dense)
    for K in (7, 40, 1024):
        ids, scores = sc.select_topk(users, dense, K)
        want = full[:, :K].clone()
        want[torch.arange(K, device=want.device)[None, :] >= n_cand[:, None]] = -1
        assert torch.equal(ids, want)
        ok = ids >= 0
        assert torch.equal(scores[ok], torch.gather(dense, 1, ids.clamp(min=0).long())[ok])
        assert torch.isneginf(scores[~ok]).all()
    sc.close()


def test_dense_row_metrics_equal_the_ranked_list_metrics(torch):
    """ifr_eval_dense_metrics (positions of the positives by counting) against ifr_rank_full +
    ifr_eval_ranked_metrics on the same rows: every column, also with plateaus of tied scores, positives that are
    excluded or beyond the catalogue, rows without positives, and the unaligned-row code path."""
    w = _workload(12, 120, 901, 32, True, mean=40)
    # a few positives that are also excluded: they are not in the ranking, so they are never hits
    tp, ti = w["truth_ptr"], w["truth_idx"]
    ex_rows = [np.union1d(w["excl_idx"][w["excl_ptr"][u]: w["excl_ptr"][u + 1]], ti[tp[u]: tp[u + 1]][:2] if u % 5 == 0 else [])
               for u in range(w["n_users"])]
    w = dict(w, excl_ptr=np.cumsum([0] + [len(r) for r in ex_rows]).astype(np.int64),
             excl_idx=np.concatenate(ex_rows).astype(np.int32))
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    ks = [-1, 5, 10, 100, 2000]
    for variant in ("scores", "plateaus", "unaligned"):
        dense = sc.score_dense(users)
        if variant == "plateaus":
            dense = torch.round(dense * 4) / 4                      # many exact ties: order falls back on the item index
        if variant == "unaligned":
            wide = torch.empty((dense.shape[0], dense.shape[1] + 2), dtype=dense.dtype, device=dense.device)
            wide[:, : dense.shape[1]] = dense
            dense = wide[:, : dense.shape[1]]                       # row stride 903: rows are not 16-byte aligned
            dense = dense[1:]
            us = users[1:]
        else:
            us = users
        ranked, n_cand = sc.rank_full(us, dense.contiguous() if variant != "unaligned" else dense.clone())
        a, a_ev, a_sum = sc.eval_ranked_metrics(us, ranked, n_cand, ks)
        b, b_ev, b_sum = sc.eval_dense_metrics(us, dense, ks)
        assert torch.equal(a_ev, b_ev) and a_ev.any() and not a_ev.all()
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=0, atol=1e-13, equal_nan=True)
        np.testing.assert_allclose(b_sum.cpu().numpy(), a_sum.cpu().numpy(), rtol=1e-13)


@pytest.mark.parametrize("k,bias,K,chunks", [(128, False, 10, 0), (50, True, 16, 3), (200, False, 1, 0), (96, True, 10, 5),
                                             (128, True, 20, 0), (64, False, 32, 2)])
def test_topk_with_dense_rows_in_one_launch(torch, k, bias, K, chunks):
    """ifr_rank_topk_dense = ifr_rank_topk + ifr_score_dense bit for bit (lists and rows), ragged user batch,
    catalogue that is not a multiple of 32, any item chunking; its argument checks."""
    from recommenders_amd import IfrError
    w = _workload(4000 + k + K, 333, 2999, k, bias)
    sc = _scorer(w)
    if chunks:
        sc.set_option("chunks", chunks)
    users = torch.as_tensor(np.random.default_rng(K).permutation(w["n_users"])[:301].astype(np.int32), device="cuda")
    ids0, sc0 = sc.rank_topk(users, K)
    rows0 = sc.score_dense(users)
    assert sc.can_rank_topk_dense(K)
    buf = torch.full((users.numel() + 3, 3008 + 32), float("nan"), dtype=torch.float32, device="cuda")
    ids1, sc1, rows1 = sc.rank_topk_dense(users, K, buf)
    assert rows1.shape == rows0.shape
    assert torch.equal(ids0, ids1) and torch.equal(sc0, sc1)
    assert torch.equal(rows0, rows1)
    assert torch.isnan(buf[users.numel():]).all() and torch.isnan(buf[:, 3008:]).all()      # nothing written beside the rows
    ids2, sc2, rows2 = sc.rank_topk_dense(users[:5], K)                                       # allocates its own rows
    assert torch.equal(ids2, ids0[:5]) and torch.equal(rows2, rows0[:5])
    assert sc.can_rank_topk_dense(128) and not sc.can_rank_topk_dense(129)
    with pytest.raises(IfrError):
        sc.rank_topk_dense(users, 129)
    with pytest.raises(ValueError):
        sc.rank_topk_dense(users, K, torch.empty((users.numel(), 2999), dtype=torch.float32, device="cuda"))


@pytest.mark.parametrize("K", [1, 7, 16, 20, 32])
def test_merge_kernels_over_list_counts(torch, K):
    """The two merge kernels (all-pairs key counting up to 256 head entries in 1 / 2 / 4 keys per lane, binary
    searches beyond) give the oracle's lists for every number of partial lists, with ties and short rows."""
    w = _workload(6000 + K, 200, 2500, 64, True, mean=30)
    w["V"][40:140] = w["V"][40]                       # 100 items with identical scores per user: ties across lists
    w["ibase"][40:140] = w["ibase"][40]
    ptr, idx = w["excl_ptr"], w["excl_idx"]
    keep = np.setdiff1d(np.arange(2500, dtype=np.int32), np.arange(0, 2500, 290, dtype=np.int32))   # user 0 keeps 9 items
    w["excl_idx"] = np.concatenate([keep, idx[ptr[1]:]]).astype(np.int32)
    w["excl_ptr"] = np.concatenate([[0], ptr[1:] - ptr[1] + len(keep)]).astype(np.int64)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    o_ids, o_sc, _, _ = _oracle(w, users, [K], K, sc.chain_order())
    for chunks in (1, 2, 4, 5, 13, 40):               # 2 * chunks lists of K head entries each
        sc.set_option("chunks", chunks)
        ids, scores = sc.rank_topk(users, K)
        np.testing.assert_array_equal(ids.cpu().numpy(), o_ids, err_msg=f"chunks={chunks}")
        np.testing.assert_array_equal(scores.cpu().numpy(), o_sc, err_msg=f"chunks={chunks}")


def test_stale_or_out_of_range_truth_rows_are_safe(torch):
    """ifr_auc_dense must treat truth entries outside the catalogue as absent (like ifr_eval_dense_metrics),
    and a truth CSR uploaded for a larger catalogue must not survive a model with fewer items."""
    from recommenders_amd.engine import IfrError
    w = _workload(21, 64, 900, 32, False)
    sc = _scorer(w, truth=False)
    users = np.arange(w["n_users"], dtype=np.int32)
    # truth rows with entries beyond the catalogue and negative ones mixed into the valid positives
    ptr, idx = [0], []
    for u in range(w["n_users"]):
        good = w["truth_idx"][w["truth_ptr"][u]: w["truth_ptr"][u + 1]]
        row = np.concatenate([[-5], good, [w["n_items"], w["n_items"] + 70000]]).astype(np.int32) if u % 2 else good
        idx.append(row)
        ptr.append(ptr[-1] + len(row))
    sc.set_truth_csr(np.asarray(ptr, np.int64), np.concatenate(idx).astype(np.int32))
    dense = sc.score_dense(users)
    got = sc.auc_dense(users, dense).cpu().numpy()
    sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
    want = sc.auc_dense(users, dense).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    # a smaller catalogue drops the truth CSR with the exclusions: the metric entry points refuse instead of reading it
    small = _workload(22, 64, 300, 32, False)
    sc.set_factors(small["U"], small["V"])
    with pytest.raises(IfrError, match="truth CSR not set"):
        sc.auc_dense(users, sc.score_dense(users))


def test_target_shape_one_million_items_bit_exact(torch):
    """BASELINE configs[2] (C3) shape on the item axis: 1,000,000 items, k = 128, biases (MF), top-10, a ragged batch of
    users, exclusions that hit the first, a middle and the last item tile.  31,250 item tiles per sweep: the regime of
    one item chunk per workgroup column and byte offsets near 2^29.  Ids and scores bit-exact against the C oracle."""
    n_users, n_items, k, K = 300, 1_000_000, 128, 10
    rng = np.random.default_rng(3)
    from recommenders_amd import synth
    U, V, ub, ib = synth.factor_model(rng, n_users, n_items, k, True)
    rows = []
    for u in range(n_users):
        r = rng.choice(n_items, 50, replace=False)
        if u % 3 == 0:
            r = np.concatenate([r, [0, 1, 31, 500_000, 500_017, n_items - 32, n_items - 1]])
        rows.append(np.unique(r).astype(np.int32))
    ex_ptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)
    ex_idx = np.concatenate(rows)
    te_ptr = np.arange(n_users + 1, dtype=np.int64) * 10
    te_idx = np.sort(rng.integers(0, n_items, (n_users, 10)).astype(np.int32), axis=1).ravel()
    w = dict(U=U, V=V, ubase=ub, ibase=ib, excl_ptr=ex_ptr, excl_idx=ex_idx, truth_ptr=te_ptr, truth_idx=te_idx,
             n_users=n_users, n_items=n_items, k=k)
    # make the excluded border items the best ones of their users, so a missed exclusion would show in the list
    V[[0, 31, 500_000, n_items - 1]] *= 4.0
    sc = _scorer(w)
    users = np.arange(n_users - 37, dtype=np.int32)[::-1].copy()           # 263 users: two full tiles + a ragged one
    ids, scores = sc.rank_topk(users, K)
    per_user, evaluated, sums = sc.eval_topk_metrics(users, ids, [K])
    o_ids, o_sc, o_met, o_ev = _oracle(w, users, [K], K, sc.chain_order())
    np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
    np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
    np.testing.assert_allclose(per_user.cpu().numpy()[:, :3], o_met[:, :3], atol=1e-12)
    for chunks in (1, 7):                                                   # placement independent at this size too
        sc.set_option("chunks", chunks)
        ids2, scores2 = sc.rank_topk(users, K)
        np.testing.assert_array_equal(ids2.cpu().numpy(), o_ids)
        np.testing.assert_array_equal(scores2.cpu().numpy(), o_sc)
    sc.set_option("chunks", 0)
    # the same launch also writing the score rows (ifr_rank_topk_dense) on a few users, against ifr_score_dense
    few = users[:5]
    ids3, scores3, dense = sc.rank_topk_dense(few, K)
    np.testing.assert_array_equal(ids3.cpu().numpy(), o_ids[:5])
    np.testing.assert_array_equal(dense.cpu().numpy(), sc.score_dense(few).cpu().numpy())
    o_dense = orc.c_score_factor(U[few[0]], V, ib, float(ub[few[0]]), sc.chain_order()[sc.chain_order() < k])
    np.testing.assert_array_equal(dense[0].cpu().numpy(), o_dense)


# ------------------------------------------------------------------ 16 < K <= 128: candidate buffers + exact selection
@pytest.mark.parametrize("k,bias,K,n_items", [(128, False, 17, 3001), (50, True, 20, 3001), (128, True, 33, 20000),
                                              (64, False, 64, 20000), (128, False, 100, 26744), (200, True, 128, 9000),
                                              (32, True, 100, 1500)])           # 1500 items: the lists cannot bound K = 100
def test_wide_topk_bit_exact_vs_oracle(torch, k, bias, K, n_items):
    w = _workload(3000 + k + K, 333, n_items, k, bias)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    ids, scores = sc.rank_topk(users, K)
    o_ids, o_sc, _, _ = _oracle(w, users, [K], K, sc.chain_order())
    np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
    np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
    redone = sc.wide_overflows()                                               # the list path may redo a few users (K <= 100)
    sc.set_option("pair_max", 16)                                              # the candidate-buffer path for every K > 16
    ids1, scores1 = sc.rank_topk(users, K)
    np.testing.assert_array_equal(ids1.cpu().numpy(), o_ids)
    np.testing.assert_array_equal(scores1.cpu().numpy(), o_sc)
    assert sc.wide_overflows() == redone                                       # ... its buffers hold these workloads
    sc.set_option("pair_max", 128)                                             # ... and the list path for every K
    ids1, scores1 = sc.rank_topk(users, K)
    np.testing.assert_array_equal(ids1.cpu().numpy(), o_ids)
    np.testing.assert_array_equal(scores1.cpu().numpy(), o_sc)
    sc.set_option("pair_max", 100)
    # placement independent: any chunk count gives the same lists
    for chunks in (1, 5):
        sc.set_option("chunks", chunks)
        ids2, scores2 = sc.rank_topk(users[:140], K)
        np.testing.assert_array_equal(ids2.cpu().numpy(), o_ids[:140])
        np.testing.assert_array_equal(scores2.cpu().numpy(), o_sc[:140])


def test_wide_topk_metrics_and_dense_rows(torch):
    w = _workload(41, 500, 8000, 128, True)
    sc = _scorer(w)
    users = np.arange(w["n_users"], dtype=np.int32)
    ks = [100, 50, 10]
    ids, scores, dense = sc.rank_topk_dense(users, 100)
    o_ids, o_sc, o_met, o_ev = _oracle(w, users, ks, 100, sc.chain_order())
    np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
    np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
    np.testing.assert_array_equal(dense.cpu().numpy(), sc.score_dense(users).cpu().numpy())
    per_user, evaluated, sums = sc.eval_topk_metrics(users, ids, ks)
    ev = o_ev.astype(bool)
    np.testing.assert_array_equal(evaluated.cpu().numpy(), o_ev)
    cols = [6 * q + c for q in range(len(ks)) for c in range(3)]        # ndcg, recall, precision of each cut-off
    np.testing.assert_allclose(per_user.cpu().numpy()[ev][:, cols], o_met[ev][:, : 3 * len(ks)], rtol=0, atol=1e-12)


def test_wide_topk_ties_short_lists_and_overflow_fallback(torch):
    from recommenders_amd import HipScorer
    # constant scores: canonical order = ascending item index; exclusions drop items 0 and 3
    n_users, n_items, k = 130, 5000, 32
    w = dict(U=np.ones((n_users, k), np.float32), V=np.ones((n_items, k), np.float32), ubase=None, ibase=None,
             excl_ptr=np.arange(n_users + 1, dtype=np.int64) * 2, excl_idx=np.tile(np.array([0, 3], np.int32), n_users),
             truth_ptr=np.zeros(n_users + 1, np.int64), truth_idx=np.zeros(0, np.int32), n_users=n_users, n_items=n_items, k=k)
    sc = _scorer(w, truth=False)
    ids, scores = sc.rank_topk(np.arange(n_users, dtype=np.int32), 40)
    expect = np.setdiff1d(np.arange(42), [0, 3]).astype(np.int32)
    np.testing.assert_array_equal(ids.cpu().numpy(), np.tile(expect, (n_users, 1)))
    # fewer candidates than K
    rng = np.random.default_rng(3)
    excl = [np.arange(90), np.arange(60), np.zeros(0, int), np.arange(5, 90), np.arange(1, 90)]
    w2 = dict(U=rng.normal(size=(5, 16)).astype(np.float32), V=rng.normal(size=(90, 16)).astype(np.float32), ubase=None, ibase=None,
              excl_ptr=np.cumsum([0] + [len(e) for e in excl]).astype(np.int64), excl_idx=np.concatenate(excl).astype(np.int32),
              truth_ptr=np.zeros(6, np.int64), truth_idx=np.zeros(0, np.int32), n_users=5, n_items=90, k=16)
    sc2 = _scorer(w2, truth=False)
    u5 = np.arange(5, dtype=np.int32)
    ids, scores = sc2.rank_topk(u5, 50)
    o_ids, o_sc, _, _ = _oracle(w2, u5, [50], 50, sc2.chain_order())
    np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
    np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
    # scores that rise with the item index: every item beats the bound of its time, the candidate buffers overflow and
    # those users are ranked from dense rows -- same lists, and the health counter says so
    n_items = 40000
    V = np.zeros((n_items, 32), np.float32)
    V[:, 0] = np.arange(n_items, dtype=np.float32) * 1e-3
    U = np.zeros((200, 32), np.float32)
    U[:, 0] = 1.0
    U[150:, 0] = -1.0                       # these users see falling scores: no overflow
    w3 = dict(U=U, V=V, ubase=None, ibase=None, excl_ptr=np.arange(201, dtype=np.int64), excl_idx=np.full(200, n_items - 1, np.int32),
              truth_ptr=np.zeros(201, np.int64), truth_idx=np.zeros(0, np.int32), n_users=200, n_items=n_items, k=32)
    sc3 = _scorer(w3, truth=False)
    sc3.set_option("pair_max", 32)                # K = 64 through the candidate buffers (the list path would redo all 200)
    u3 = np.arange(200, dtype=np.int32)
    ids, scores = sc3.rank_topk(u3, 64)
    o_ids, o_sc, _, _ = _oracle(w3, u3, [64], 64, sc3.chain_order())
    np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
    np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
    assert 0 < sc3.wide_overflows() <= 150


def test_pair_topk_lost_item_is_caught(torch):
    """16 < K <= 32 (two 16-lists per user and item chunk): when more than 16 of a user's best items sit in ONE list --
    here the 24 best are the items whose index is 0..3 mod 8 inside the first tiles, i.e. one half-lane's rows of one
    chunk -- that list drops some of them; the merge notices and those users are ranked again from dense rows."""
    n_users, n_items, k = 300, 6000, 32
    rng = np.random.default_rng(12)
    U = rng.normal(0, 0.1, (n_users, k)).astype(np.float32)
    V = rng.normal(0, 0.1, (n_items, k)).astype(np.float32)
    U[:, 0] = 1.0
    V[:, 0] = 0.0
    best = np.array([8 * q + r for q in range(6) for r in range(4)])          # 24 items of one half-lane, tiles 0 and 1
    V[best, 0] = 50.0 + rng.random(len(best)).astype(np.float32)
    w = dict(U=U, V=V, ubase=None, ibase=None, excl_ptr=np.zeros(n_users + 1, np.int64), excl_idx=np.zeros(0, np.int32),
             truth_ptr=np.zeros(n_users + 1, np.int64), truth_idx=np.zeros(0, np.int32), n_users=n_users, n_items=n_items, k=k)
    sc = _scorer(w, truth=False)
    users = np.arange(n_users, dtype=np.int32)
    for K in (20, 32):
        before = sc.wide_overflows()
        ids, scores = sc.rank_topk(users, K)
        o_ids, o_sc, _, _ = _oracle(w, users, [K], K, sc.chain_order())
        np.testing.assert_array_equal(ids.cpu().numpy(), o_ids)
        np.testing.assert_array_equal(scores.cpu().numpy(), o_sc)
        if K == 20:
            assert sc.wide_overflows() - before == n_users                     # every user went through the fallback
        # ... after which the handle sends K >= 20 to the candidate-buffer path (this shape does not suit the lists)
    # the candidate-buffer path gives the same lists
    sc.set_option("pair_topk", 0)
    ids2, scores2 = sc.rank_topk(users, 20)
    o_ids, o_sc, _, _ = _oracle(w, users, [20], 20, sc.chain_order())
    np.testing.assert_array_equal(ids2.cpu().numpy(), o_ids)


def test_item_bases_survive_fast_waves(torch):
    """Regression for a race of round 1: with two LDS slots for the item bases a fast wave stored tile t + 1's bases
    over tile t - 1's while a slower wave of the workgroup still read them (wrong scores for a few users in 1 of 20
    launches at k = 32, where the MFMA zone is short).  Many launches of that shape, every one bit-exact."""
    n_users = 128 * 552 + 17
    w = _workload(6, n_users, 2048, 32, True, mean=12)
    users = np.arange(n_users, dtype=np.int32)
    sc = _scorer(w)
    o_ids, o_sc, _, _ = _oracle(w, users, [10], 10, sc.chain_order())
    for rep in range(25):
        for tb in (1, 4096):
            sc.set_option("target_blocks", tb)
            ids, scores = sc.rank_topk(users, 10)
            assert np.array_equal(ids.cpu().numpy(), o_ids) and np.array_equal(scores.cpu().numpy(), o_sc), (rep, tb)
