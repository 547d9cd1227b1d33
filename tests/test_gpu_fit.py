"""GPU tests of the model build entry points ifr_fit_ease / ifr_fit_rp3beta
through the C ABI.

Bars: B against the reference's own EASE.fit output (tests/golden/toy_all_models.npz, made by
tests/golden/make_golden.py) and against the oracle's restatement of the closed form
(oracle.ease_closed_form, recom_ease.py:78-95, pinned by that same B in test_oracle.py): 1e-8 relative to the largest entry (Cholesky here, LU there); the fp32 matrix the
scoring kernels read is exactly fl32(B)."""
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


def test_ease_fit_matches_reference_output(torch, toy):
    from recommenders_amd.data import Dataset
    from recommenders_amd.models import EASE
    d = toy
    train_set = Dataset.from_csr(_csr(d, "train"))
    dev = EASE(lamb=50, posB=True).fit(train_set)
    ref_B = d["ease_B"]
    np.testing.assert_allclose(dev.B, ref_B, rtol=0, atol=1e-8 * np.abs(ref_B).max())
    assert (np.diag(dev.B) == 0).all() and (dev.B >= 0).all()
    # the device-resident model scores like the one uploaded from the reference's B
    host = EASE(lamb=50, trainable=False, init_params={"B": ref_B}).fit(train_set)
    for u in (0, 3, train_set.num_users - 1):
        np.testing.assert_allclose(dev.score(u), host.score(u), rtol=1e-6, atol=1e-9)
        a, b = dev.rank(u, k=5)[0], host.rank(u, k=5)[0]
        assert set(a.tolist()) == set(b.tolist())


@pytest.mark.parametrize("posB,ratings", [(False, True), (True, False)])
def test_ease_fit_matches_closed_form_and_installs_fl32_B(torch, posB, ratings):
    from recommenders_amd import HipScorer
    rng = np.random.default_rng(5)
    n_users, n_items = 3000, 1537
    X = sp.random(n_users, n_items, density=0.02, format="csr", dtype=np.float32, random_state=np.random.RandomState(5),
                  data_rvs=lambda n: (rng.integers(1, 6, n) if ratings else np.ones(n)).astype(np.float32))
    X.sort_indices()
    sc = HipScorer(0)
    B = sc.fit_ease(X.indptr.astype(np.int64), X.indices, X.data, n_items, 500.0, posB, want_B=True)
    ref = orc.ease_closed_form(X, 500.0, posB)
    np.testing.assert_allclose(B, ref, rtol=0, atol=1e-8 * np.abs(ref).max())
    # one-item histories read rows of the installed matrix: exactly fl32(B)
    items = np.array([0, 7, 1024, n_items - 1], np.int32)
    sc.set_history_csr(np.arange(len(items) + 1, dtype=np.int64), items, np.ones(len(items), np.float32))
    rows = sc.score_dense(np.arange(len(items), dtype=np.int32)).cpu().numpy()
    np.testing.assert_array_equal(rows[:, :n_items], B[items].astype(np.float32))
    sc.close()


def test_ease_fit_rejects_bad_input(torch):
    from recommenders_amd import HipScorer
    sc = HipScorer(0)
    ptr = np.array([0, 2], np.int64)
    with pytest.raises(RuntimeError, match="ascending"):
        sc.fit_ease(ptr, np.array([3, 1], np.int32), np.ones(2, np.float32), 5, 10.0)
    with pytest.raises(RuntimeError, match="positive definite"):
        sc.fit_ease(ptr, np.array([1, 3], np.int32), np.ones(2, np.float32), 5, 0.0)      # items 0, 2, 4 never seen
    with pytest.raises(RuntimeError, match="non-finite"):
        sc.fit_ease(ptr, np.array([1, 3], np.int32), np.array([1.0, np.inf], np.float32), 5, 10.0)
    sc.close()


@pytest.mark.parametrize("n_users,n_items,mean_hist", [(2500, 1800, 20), (20000, 700, 40)])
def test_rp3_fit_installs_w_histories_and_penalty(torch, n_users, n_items, mean_hist):
    """ifr_fit_rp3beta against the host restatement of graph_recommender.py:24-55 (oracle.rp3_model_arrays, pinned
    by the reference's RP3 scores in test_oracle.py): the installed CSR matrix read back through one-item
    histories, and the rankings of the installed model against the same model uploaded from the host arrays."""
    from recommenders_amd import HipScorer, synth
    rng = np.random.default_rng(9)
    # the second shape: item degrees above 1000, sums of a thousand different float64 terms per entry of W
    ptr, idx, _, _ = synth.interactions(rng, n_users, n_items - 100, mean_hist, lo=0)   # the last 100 items: no user
    A = sp.csr_matrix((np.ones(len(idx)), idx, ptr), shape=(n_users, n_items))
    assert (A.getnnz(axis=0) == 0).any()                      # items nobody touched: col_scale = inf
    beta = 0.7
    P32, W32, pen = orc.rp3_model_arrays(A, beta)
    dev = HipScorer(0)
    w_nnz = dev.fit_rp3beta(ptr, idx, n_items, beta)
    assert w_nnz == W32.nnz
    users = np.arange(n_users + 1, dtype=np.int32)            # the last one has no history row
    host = HipScorer(0)
    host.set_item_matrix_csr(W32.indptr.astype(np.int64), W32.indices, W32.data, n_items, pen)
    host.set_history_csr(P32.indptr.astype(np.int64), P32.indices, P32.data)
    for s_ in (dev, host):
        s_.set_exclusion_csr(ptr, idx)
    a_ids, a_sc = dev.rank_topk(users, 10)
    b_ids, b_sc = host.rank_topk(users, 10)
    a_ids, b_ids, a_sc, b_sc = a_ids.cpu().numpy(), b_ids.cpu().numpy(), a_sc.cpu().numpy(), b_sc.cpu().numpy()
    # W is summed in SciPy's order (ascending user, float64): the device build IS the host build, bit for bit
    np.testing.assert_array_equal(a_sc, b_sc)
    np.testing.assert_array_equal(a_ids, b_ids)
    # W itself: rows of the installed matrix (col_scale folded in) through one-item histories
    items = np.array([0, 5, n_items // 2, n_items - 1], np.int32)
    dev.set_history_csr(np.arange(len(items) + 1, dtype=np.int64), items, np.ones(len(items), np.float32))
    rows = dev.score_dense(np.arange(len(items), dtype=np.int32)).cpu().numpy()[:, :n_items]
    with np.errstate(invalid="ignore"):
        want = (W32[items].toarray() * pen[None, :]).astype(np.float32)
    np.testing.assert_array_equal(rows, want)                # NaN (0 * inf) compares equal to NaN here
    # ... and run to run: a second build gives the same bytes
    dev2 = HipScorer(0)
    dev2.fit_rp3beta(ptr, idx, n_items, beta)
    dev2.set_history_csr(np.arange(len(items) + 1, dtype=np.int64), items, np.ones(len(items), np.float32))
    rows2 = dev2.score_dense(np.arange(len(items), dtype=np.int32)).cpu().numpy()[:, :n_items]
    np.testing.assert_array_equal(rows2, rows)
    dev.close(); host.close(); dev2.close()


def test_rp3_facade_device_build_matches_host_arrays(torch, toy):
    """RP3_Beta.fit (ifr_fit_rp3beta on the device) scores exactly like the same facade fed the oracle's SciPy arrays."""
    from recommenders_amd.data import Dataset
    from recommenders_amd.models import RP3_Beta
    d = toy
    train_set = Dataset.from_csr(_csr(d, "train"))
    dev = RP3_Beta(beta=float(d["rp3_beta"]), item_matrix="csr").fit(train_set)
    P_ui, W, deg = orc.rp3_item_item(sp.csr_matrix((_csr(d, "train") > 0).astype(np.float64)))
    host = RP3_Beta(beta=float(d["rp3_beta"]), item_matrix="csr",
                    init_params={"P_ui": P_ui, "W": W, "item_degrees": deg}).fit(train_set)
    assert dev.W is None and dev.w_nnz == host.W.nnz
    for u in range(0, train_set.num_users, 7):
        np.testing.assert_array_equal(dev.score(u), host.score(u))
