"""Seeded synthetic workloads of the shapes BASELINE.json names.

Factors ~ N(0, 0.1^2) fp32 (continuous => tie-free), per-user interaction counts ~
clipped log-normal, items drawn from a Zipf-like popularity law without replacement
inside a user, 80/20 train/test split per interaction, ratings 1.0.
"""
import numpy as np

CONFIGS = {
    # name: (n_users, n_items, k, mean interactions per user, seed, has_bias)
    "c1_bpr_ml100k": (943, 1682, 50, 106, 123, True),
    "c2_wmf_ml20m": (138493, 26744, 128, 145, 20, False),
    "c3_mf_shard": (1250000, 1000000, 128, 60, 3, True),   # one GPU's shard of 10M x 1M
}


def zipf_items(rng, n, n_items, a=0.8):
    """n item ids with P(rank r) ~ (r+1)^-a, by inverse CDF."""
    w = 1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), a)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    perm = rng.permutation(n_items).astype(np.int32)        # popularity rank -> item id
    r = np.searchsorted(cdf, rng.random(n), side="left")
    return perm[np.minimum(r, n_items - 1)]


def interactions(rng, n_users, n_items, mean_per_user, lo=5, hi=2000, test_frac=0.2):
    """-> (train_indptr, train_indices, test_indptr, test_indices) with sorted, disjoint rows."""
    sigma = 1.0
    # x1.15: about 13% of the draws are duplicates inside a user and are dropped below
    mu = np.log(1.15 * mean_per_user) - 0.5 * sigma * sigma
    cnt = np.clip(rng.lognormal(mu, sigma, n_users), lo, min(hi, n_items // 2)).astype(np.int64)
    users = np.repeat(np.arange(n_users, dtype=np.int64), cnt)
    items = zipf_items(rng, len(users), n_items).astype(np.int64)
    key = np.unique(users * n_items + items)                # drop duplicates, sort by (user, item)
    users, items = key // n_items, (key % n_items).astype(np.int32)
    is_test = rng.random(len(key)) < test_frac

    def csr(mask):
        ptr = np.zeros(n_users + 1, np.int64)
        ptr[1:] = np.bincount(users[mask], minlength=n_users)
        return np.cumsum(ptr), items[mask]

    tr_ptr, tr_idx = csr(~is_test)
    te_ptr, te_idx = csr(is_test)
    return tr_ptr, tr_idx, te_ptr, te_idx


def factor_model(rng, n_users, n_items, k, has_bias):
    U = rng.normal(0, 0.1, (n_users, k)).astype(np.float32)
    V = rng.normal(0, 0.1, (n_items, k)).astype(np.float32)
    ibase = rng.normal(0, 0.01, n_items).astype(np.float32) if has_bias else None
    ubase = rng.normal(0, 0.01, n_users).astype(np.float32) if has_bias else None
    return U, V, ubase, ibase


def make(name, n_users=None, n_items=None, seed=None):
    """Workload dict for a named config (optionally scaled down for parity tests)."""
    nu, ni, k, mean, sd, bias = CONFIGS[name]
    nu = nu if n_users is None else n_users
    ni = ni if n_items is None else n_items
    rng = np.random.default_rng(sd if seed is None else seed)
    U, V, ubase, ibase = factor_model(rng, nu, ni, k, bias)
    tr_ptr, tr_idx, te_ptr, te_idx = interactions(rng, nu, ni, mean)
    return dict(name=name, n_users=nu, n_items=ni, k=k, U=U, V=V, ubase=ubase, ibase=ibase,
                excl_ptr=tr_ptr, excl_idx=tr_idx, truth_ptr=te_ptr, truth_idx=te_idx)
