"""GPU tests of the candidate-set builder on the device (ifr_build_eval_csrs, csrc/ifr_evalcsr.hip): the truth and
exclusion CSRs it installs must equal the host build (`eval.build_eval_csrs`, SciPy algebra that restates
cornac/eval_methods/base_method.py:246-283) bit for bit, and ranking_eval through it must give what the reference gave."""
import numpy as np
import pytest
import scipy.sparse as sp

from recommenders_amd import DeviceCsr, HipScorer, ranking_eval
from recommenders_amd import eval as ev
from recommenders_amd.data import Dataset, DeviceDataset
from test_gpu_facade import _gap_sets, _metrics, _model, _sets
from test_oracle import _csr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _engine(n_users, n_items, k=8, seed=0):
    rng = np.random.default_rng(seed)
    sc = HipScorer(0)
    sc.set_factors(rng.normal(size=(n_users, k)).astype(np.float32), rng.normal(size=(n_items, k)).astype(np.float32))
    return sc


def _compare(sc, train, val, test, thr, n_items, row0, n_rows):
    st = sc.build_eval_csrs(train, val, test, thr, n_items, row0, n_rows)
    truth, excl, has_pos = ev.build_eval_csrs(train, test, val, thr, n_items, n_rows, row0)
    for which, ref in (("truth", truth), ("exclusion", excl)):
        indptr, indices = sc.read_eval_csr(which)
        np.testing.assert_array_equal(indptr, ref.indptr.astype(np.int64), err_msg=which)
        np.testing.assert_array_equal(indices, ref.indices.astype(np.int32), err_msg=which)
    np.testing.assert_array_equal(st["has_pos"].cpu().numpy().astype(bool), has_pos)
    assert st["truth_nnz"] == truth.nnz and st["excl_nnz"] == excl.nnz
    assert st["max_truth_row"] == (int(np.diff(truth.indptr).max()) if n_rows else 0)
    return truth, excl


def test_toy_fixture_with_validation_split(toy):
    d = toy
    train, test, val = _csr(d, "train"), _csr(d, "test"), _csr(d, "val")
    n_items = int(d["num_items"])
    sc = _engine(train.shape[0], n_items)
    n_rows = int(np.flatnonzero(np.diff(test.indptr)).max()) + 1
    truth, excl = _compare(sc, train, val, test, float(d["rating_threshold"]), n_items, 0, n_rows)
    assert truth.nnz > 0 and excl.nnz > 0
    _compare(sc, train, None, test, float(d["rating_threshold"]), n_items, 0, n_rows)          # no validation split
    _compare(sc, train, val, test, 1.0, n_items, 0, n_rows)                                    # everything is a positive
    _compare(sc, train, val, test, 1e9, n_items, 0, n_rows)                                    # nothing is
    sc.close()


def test_unknowns_fixture_with_items_and_users_beyond_the_model(unknowns):
    """exclude_unknowns=False: the catalogue (test_set.num_items) is larger than the model's, test rows exist for users
    the train matrix has no row for (base_method.py:266-270)."""
    d = unknowns
    train, test = _csr(d, "train"), _csr(d, "test")
    sc = _engine(train.shape[0], train.shape[1])
    n_rows = int(np.flatnonzero(np.diff(test.indptr)).max()) + 1
    assert test.shape[1] > train.shape[1] and n_rows > train.shape[0]
    for n_items in (test.shape[1], train.shape[1]):
        _compare(sc, train, None, test, float(d["rating_threshold"]), n_items, 0, n_rows)
    sc.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_random_splits_row_ranges_and_clipping(dtype):
    rng = np.random.default_rng(7)
    nu, ni = 700, 900
    def split(density, rows):
        m = sp.random(rows, ni, density=density, random_state=rng, format="csr", dtype=np.float64)
        m.data = np.ceil(m.data * 5).astype(dtype)
        m.sort_indices()
        return m
    train, val, test = split(0.05, nu), split(0.01, nu - 50), split(0.02, nu + 30)
    # a user with a row longer than a wave handles in one step, and empty rows
    dense_row = sp.csr_matrix((np.full(400, 5, dtype), (np.zeros(400, int), np.arange(0, 800, 2))), shape=(1, ni))
    train = sp.vstack([dense_row, train[1:]]).tocsr()
    test = sp.vstack([test[:5], sp.csr_matrix((3, ni), dtype=dtype), test[8:]]).tocsr()
    sc = _engine(nu + 30, ni)
    for thr in (3.0, 4.5):
        for row0, n_rows in ((0, nu + 30), (0, 100), (123, 321), (600, 130), (0, 0)):
            for n_items in (ni, 500):
                _compare(sc, train, val, test, thr, n_items, row0, n_rows)
    # device-resident inputs give the same
    dev = [DeviceCsr.of(m, sc.device) for m in (train, val, test)]
    st = sc.build_eval_csrs(dev[0], dev[1], dev[2], 3.0, ni, 0, nu + 30)
    truth, excl, _ = ev.build_eval_csrs(train, test, val, 3.0, ni, nu + 30)
    np.testing.assert_array_equal(sc.read_eval_csr("exclusion")[1], excl.indices)
    np.testing.assert_array_equal(sc.read_eval_csr("truth")[1], truth.indices)
    assert st["excl_nnz"] == excl.nnz
    sc.close()


@pytest.mark.parametrize("name", ["bpr", "wmf"])
def test_ranking_eval_through_the_device_builder(gap, name, monkeypatch):
    """The facade with the device builder forced on (threshold 0 ratings): same per-user values and means as the
    reference, the candidate sets of the lazy caches read back from the engine."""
    monkeypatch.setattr(ev, "DEVICE_BUILD_MIN_NNZ", 0)
    ev.clear_eval_cache()
    d = gap
    train_set, test_set, val_set = _gap_sets(d, name)
    model = _model(d, name, train_set)
    mets = _metrics(d)
    thr = float(d["rating_threshold"])
    avg, per = ranking_eval(model, mets, train_set, test_set, val_set=val_set, rating_threshold=thr, exclude_unknowns=True)
    src = model._hip_cache_owner.excl_source
    assert isinstance(src, ev.DeviceEvalCsrs) and model._hip_engine().exclusion_source is src
    users = d[f"{name}__users"]
    assert sorted(per[0].keys()) == users.tolist()
    ref = d[f"{name}__per_user"]
    for j, mt in enumerate(mets):
        got = np.asarray([per[j][int(u)] for u in users])
        np.testing.assert_allclose(got, ref[:, j], atol=1e-6, rtol=0, err_msg=mt.name)
    np.testing.assert_allclose(avg, d[f"{name}__avg"], atol=1e-6, rtol=0)
    # top-K metrics alone: the values stay on the device until read
    avg2, per2 = ranking_eval(model, mets[:3], train_set, test_set, val_set=val_set, rating_threshold=thr)
    assert per2[0]._vals is None
    np.testing.assert_allclose(avg2, d[f"{name}__avg"][:3], atol=1e-6, rtol=0)
    np.testing.assert_allclose([per2[1][int(u)] for u in users], ref[:, 1], atol=1e-6, rtol=0)
    assert per2[1]._vals is not None and per2[0]._vals is None
    # host form of the device-built exclusion rows = the host build; the lazy caches use them
    n_rows = int(users.max()) + 1
    _, excl, _ = ev.build_eval_csrs(train_set.csr_matrix, test_set.csr_matrix, val_set.csr_matrix, thr, train_set.num_items, n_rows)
    got = model._hip_cache_owner.excl_global()
    np.testing.assert_array_equal(got.indptr, excl.indptr)
    np.testing.assert_array_equal(got.indices, excl.indices)
    u = int(users[3])
    np.testing.assert_array_equal(model.ranked_items[u][:20], d[f"{name}__ranked_head"][3][:20])
    ev.clear_eval_cache()


def test_device_resident_splits_and_unknown_items(unknowns, monkeypatch):
    """DeviceDataset splits (the raw matrices already in device memory) with exclude_unknowns=False."""
    from collections import OrderedDict
    import torch
    ev.clear_eval_cache()
    d = unknowns
    train_set, test_set, _ = _sets(d)
    model = _model(d, "bpr", train_set)
    model.uid_map = OrderedDict((i, i) for i in range(int(d["total_users"])))
    model.iid_map = OrderedDict((i, i) for i in range(int(d["total_items"])))
    mets = _metrics(d)
    dev = torch.device("cuda:0")
    dtrain = DeviceDataset(DeviceCsr.of(train_set.csr_matrix, dev))
    dtest = DeviceDataset(DeviceCsr.of(test_set.csr_matrix, dev))
    avg, per = ranking_eval(model, mets, dtrain, dtest, rating_threshold=float(d["rating_threshold"]), exclude_unknowns=False)
    users = d["bpr__users"]
    assert sorted(per[0].keys()) == users.tolist()
    ref = d["bpr__per_user"]
    for j, mt in enumerate(mets):
        got = np.asarray([per[j][int(u)] for u in users])
        np.testing.assert_allclose(got, ref[:, j], atol=1e-6, rtol=0, err_msg=mt.name)
    np.testing.assert_allclose(avg, d["bpr__avg"], atol=1e-6, rtol=0)
    ranked = model.ranked_items[int(users[0])]
    assert len(ranked) == d["bpr__n_cand"][0]
    np.testing.assert_array_equal(ranked[:19], d["bpr__ranked_head"][0][:19])
    ev.clear_eval_cache()
