import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def toy():
    return load_golden("toy_all_models.npz")


@pytest.fixture(scope="session")
def gap():
    """Gap-guaranteed reference outputs for all five models (tests/golden/make_golden.py::make_gap): per model a test
    split that only holds users whose decisive reference scores are separated by > 1e-4 relative, so parity tests
    assert on every evaluated user."""
    return load_golden("gap_all_models.npz")


@pytest.fixture(scope="session")
def c1():
    return load_golden("c1_bpr_ml100k_shape.npz")


@pytest.fixture(scope="session")
def unknowns():
    return load_golden("unknowns_mf_bpr.npz")


@pytest.fixture(scope="session")
def list_metrics():
    """Reference ranking_eval values for k = -1 / 50 / 100 / 200 cut-offs, MRR, MAP, NCRR, HitRatio, FMeasure
    (tests/golden/make_golden.py::make_list_metrics)."""
    return load_golden("list_metrics_bpr_ease.npz")


def metric_objects(names):
    """['ndcg@50', 'mrr@-1', ...] -> recommenders_amd.metrics objects."""
    from recommenders_amd import metrics as M
    cls = {"ndcg": M.NDCG, "recall": M.Recall, "precision": M.Precision, "ncrr": M.NCRR, "hit": M.HitRatio,
           "f1": M.FMeasure}
    out = []
    for n in names:
        name, k = str(n).split("@")
        out.append({"auc": M.AUC, "mrr": M.MRR, "map": M.MAP}[name]() if name in ("auc", "mrr", "map")
                   else cls[name](k=int(k)))
    return out


def positives_clear(scores, cand, pos, rel=1e-5):
    """True when every positive's score is separated from both neighbours in the sorted candidate scores by more
    than fp32 rounding can move it: then the positives' positions -- all a list metric reads -- do not depend on
    whether the scores were summed in fp32 (reference BPR, the HIP path) or fp64."""
    s = np.asarray(scores, np.float64)[cand]
    order = np.argsort(s, kind="stable")
    ss = s[order]
    where = np.flatnonzero(np.isin(np.asarray(cand)[order], pos))
    for j in where:
        tol = rel * max(1e-3, abs(ss[j]))
        if (j > 0 and ss[j] - ss[j - 1] <= tol) or (j + 1 < len(ss) and ss[j + 1] - ss[j] <= tol):
            return False
    return True
