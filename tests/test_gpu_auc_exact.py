"""GPU test of ifr_auc_dense / ifr_auc_dense_f64 (cornac/metrics/ranking.py:446-485) against an exact NumPy pair count:
the per-user bucket table (five or more positives) and the binary search (fewer) must both give
#{(p, q): s_p > s_q} / (|pos| |neg|) EXACTLY -- ties, constant rows, score ranges of 1e-7, +-inf / NaN scores, rows of
1 ... 300 positives, exclusions, fp32 and float64 rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _expected(rows, pos, exc, n_items):
    out = np.full(len(pos), np.nan)
    for u in range(len(pos)):
        if len(pos[u]) == 0:
            continue
        neg = np.setdiff1d(np.arange(n_items), np.r_[pos[u], exc[u]])
        sp, sn = rows[u, pos[u]], rows[u, neg]
        with np.errstate(invalid="ignore"):
            wins = int((sp[:, None] > sn[None, :]).sum())
        out[u] = wins / (len(sp) * float(len(sn))) if len(sn) else np.nan
    return out


@pytest.mark.parametrize("f64", [False, True])
@pytest.mark.parametrize("kind", ["normal", "quantised", "constant", "tiny_range", "special"])
def test_auc_equals_pair_count(torch, kind, f64):
    from recommenders_amd import HipScorer
    rng = np.random.default_rng(100 + 2 * ["normal", "quantised", "constant", "tiny_range", "special"].index(kind) + int(f64))
    n_users, n_items = 24, 4100
    n_pos = [0, 1, 2, 4, 5, 6, 9, 17, 31, 32, 33, 64, 100, 300] + [int(v) for v in rng.integers(1, 60, n_users - 14)]
    pos = [np.sort(rng.choice(n_items, n, replace=False)) for n in n_pos]
    exc = [np.setdiff1d(rng.choice(n_items, int(rng.choice([0, 5, 200])), replace=False), p) for p in pos]
    sc = HipScorer(0)
    sc.set_factors(np.zeros((n_users, 8), np.float32), np.zeros((n_items, 8), np.float32))
    ptr = lambda rows: np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)
    cat = lambda rows: np.concatenate(rows).astype(np.int32)
    sc.set_truth_csr(ptr(pos), cat(pos))
    sc.set_exclusion_csr(ptr(exc), cat(exc))
    rows = rng.normal(size=(n_users, n_items))
    if kind == "quantised":
        rows = np.round(rows * 3) / 3                     # many ties, also between positives and negatives
    elif kind == "constant":
        rows[:] = 1.25
    elif kind == "tiny_range":
        rows = 1.0 + rows * 1e-7
    elif kind == "special":
        m = rng.random(rows.shape)
        rows[m < 0.01] = np.inf
        rows[(m >= 0.01) & (m < 0.02)] = -np.inf
        rows[(m >= 0.02) & (m < 0.03)] = np.nan
        for u, p in enumerate(pos):                       # (the positives keep finite scores)
            rows[u, p] = rng.normal(size=len(p))
    rows = rows.astype(np.float64 if f64 else np.float32)
    ld = (n_items + 31) // 32 * 32
    buf = torch.zeros((n_users, ld), dtype=torch.float64 if f64 else torch.float32, device="cuda")
    buf[:, :n_items] = torch.as_tensor(rows, device="cuda")
    got = sc.auc_dense(np.arange(n_users, dtype=np.int32), buf).cpu().numpy()
    np.testing.assert_array_equal(got, _expected(rows, pos, exc, n_items))
    sc.close()
