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


def make(name, n_users=None, n_items=None, seed=None, k=None):
    """Workload dict for a named config (optionally scaled down for parity tests; k: another factor dimension)."""
    nu, ni, k0, mean, sd, bias = CONFIGS[name]
    k = k0 if k is None else int(k)
    nu = nu if n_users is None else n_users
    ni = ni if n_items is None else n_items
    rng = np.random.default_rng(sd if seed is None else seed)
    U, V, ubase, ibase = factor_model(rng, nu, ni, k, bias)
    tr_ptr, tr_idx, te_ptr, te_idx = interactions(rng, nu, ni, mean)
    return dict(name=name, n_users=nu, n_items=ni, k=k, U=U, V=V, ubase=ubase, ibase=ibase,
                excl_ptr=tr_ptr, excl_idx=tr_idx, truth_ptr=te_ptr, truth_idx=te_idx)


# ---------------------------------------------------------------------------- item-item workloads
ITEM_CONFIGS = {
    # name: (n_users, n_items, mean history length, seed, neighbours per row of a sparse W or None)
    "c4_ease_mind": (262144, 50000, 30, 4, None),        # EASE, dense B 50K x 50K (10 GB fp32), MIND-large shape
    "c5_rp3_ebnerd": (1048576, 125000, 35, 5, 256),      # RP3-beta, sparse W, EB-NeRD shape
}


def ease_like_matrix(n_items, seed, device=None):
    """B = max(0, N(0, 1e-3)) with a zero diagonal (mirrors posB and the zeroed diagonal of
    recom_ease.py:86-92).  On `device` (a torch device) the matrix is generated there in row
    blocks; otherwise with NumPy."""
    if device is None:
        rng = np.random.default_rng(seed)
        B = np.maximum(0, rng.normal(0, 1e-3, (n_items, n_items))).astype(np.float32)
        np.fill_diagonal(B, 0.0)
        return B
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    B = torch.empty((n_items, n_items), dtype=torch.float32, device=device)
    step = 4096
    for r in range(0, n_items, step):
        blk = B[r:r + step]
        blk.normal_(0, 1e-3, generator=g)
        blk.clamp_(min=0)
    B.fill_diagonal_(0.0)
    return B


def sparse_item_matrix(rng, n_items, row_nnz):
    """CSR W with `row_nnz` strictly ascending, stratified-random columns per row and positive
    weights of the size RP3-beta's P_iu.P_ui has (~1/degree^2)."""
    base = (np.arange(row_nnz + 1, dtype=np.int64) * n_items) // row_nnz       # stratum k = [base[k], base[k+1])
    width = np.diff(base)
    cols = (base[None, :-1] + np.floor(rng.random((n_items, row_nnz)) * width[None, :]).astype(np.int64)).astype(np.int32)
    vals = (rng.random((n_items, row_nnz)) * 1e-3).astype(np.float32)
    ptr = np.arange(n_items + 1, dtype=np.int64) * row_nnz
    return ptr, cols.ravel(), vals.ravel()


def make_item(name, n_users=None, n_items=None, seed=None, device=None, row_nnz=None):
    """Item-item workload dict: histories (train rows with weights), exclusions (= the train items),
    truth (held-out items), and M either dense ("M") or CSR ("m_ptr", "m_idx", "m_val") + col_scale."""
    nu, ni, mean, sd, nnz = ITEM_CONFIGS[name]
    nu = nu if n_users is None else n_users
    ni = ni if n_items is None else n_items
    nnz = nnz if row_nnz is None else row_nnz
    rng = np.random.default_rng(sd if seed is None else seed)
    tr_ptr, tr_idx, te_ptr, te_idx = interactions(rng, nu, ni, mean, lo=5, hi=400)
    out = dict(name=name, n_users=nu, n_items=ni, h_ptr=tr_ptr, h_idx=tr_idx, excl_ptr=tr_ptr, excl_idx=tr_idx,
               truth_ptr=te_ptr, truth_idx=te_idx, M=None, m_ptr=None, col_scale=None)
    cnt = np.diff(tr_ptr)
    if nnz is None:                                  # EASE: x_uj = the rating (1.0: implicit clicks)
        out["h_val"] = np.ones(len(tr_idx), np.float32)
        out["M"] = ease_like_matrix(ni, sd if seed is None else seed, device)
    else:                                            # RP3-beta: x_uj = 1/deg(u), col_scale = deg(item)^-0.7
        out["h_val"] = np.repeat((1.0 / np.maximum(cnt, 1)).astype(np.float32), cnt)
        out["m_ptr"], out["m_idx"], out["m_val"] = sparse_item_matrix(rng, ni, min(nnz, ni))
        deg = np.bincount(tr_idx, minlength=ni).astype(np.float64)
        with np.errstate(divide="ignore"):
            out["col_scale"] = (1.0 / np.maximum(deg, 1.0) ** 0.7).astype(np.float32)
    return out


# ---------------------------------------------------------------------------- C3 on the device
def make_c3_device(n_users=None, n_items=None, seed=None, device=None, per_user=60, n_test=10):
    """configs[2] shard generated on the GPU (torch tensors on `device`): 1.25 M users x 60 interactions take minutes
    with the NumPy generator above and seconds here.  Factors and biases as in `factor_model`; every user has
    `per_user` distinct items, one drawn uniformly from each of `per_user` equal strata of the catalogue (ascending
    by construction, no duplicates), `n_test` of them at random held out as the truth row, the rest the exclusion row
    ("about 50 train + 10 test items per user").  Item popularity is uniform, not Zipf: the
    fused kernel's cost does not depend on which items are excluded."""
    import torch
    nu, ni, k, _, sd, _ = CONFIGS["c3_mf_shard"]
    nu = nu if n_users is None else n_users
    ni = ni if n_items is None else n_items
    g = torch.Generator(device=device)
    g.manual_seed(sd if seed is None else seed)
    f32 = dict(dtype=torch.float32, device=device)
    U = torch.empty((nu, k), **f32).normal_(0, 0.1, generator=g)
    V = torch.empty((ni, k), **f32).normal_(0, 0.1, generator=g)
    ubase = torch.empty((nu,), **f32).normal_(0, 0.01, generator=g)
    ibase = torch.empty((ni,), **f32).normal_(0, 0.01, generator=g)
    per_user = min(per_user, ni)
    n_test = min(n_test, per_user - 1)
    width = ni // per_user
    base = torch.arange(per_user, device=device, dtype=torch.int64) * width
    items = base[None, :] + (torch.rand((nu, per_user), generator=g, device=device) * width).long().clamp_(max=width - 1)
    pick = torch.rand((nu, per_user), generator=g, device=device).argsort(dim=1)[:, :n_test]
    is_test = torch.zeros((nu, per_user), dtype=torch.bool, device=device)
    is_test.scatter_(1, pick, True)
    truth_idx = items[is_test].to(torch.int32)                       # row-major order: ascending inside a row
    excl_idx = items[~is_test].to(torch.int32)
    n_ex = per_user - n_test
    return dict(name="c3_mf_shard", n_users=nu, n_items=ni, k=k, U=U, V=V, ubase=ubase, ibase=ibase,
                excl_ptr=torch.arange(nu + 1, dtype=torch.int64) * n_ex, excl_idx=excl_idx,
                truth_ptr=torch.arange(nu + 1, dtype=torch.int64) * n_test, truth_idx=truth_idx)


def host_sample(dw, m):
    """NumPy copies of what a CPU checker needs for the first m users of a device workload (all of V, m rows of U
    and of the CSRs); n_users stays the full count, `host_rows` = m."""
    m = int(min(m, dw["n_users"]))
    e_end, t_end = int(dw["excl_ptr"][m]), int(dw["truth_ptr"][m])
    cpu = lambda t: t.cpu().numpy()
    return dict(name=dw["name"], n_users=dw["n_users"], n_items=dw["n_items"], k=dw["k"], host_rows=m,
                U=cpu(dw["U"][:m]), V=cpu(dw["V"]), ubase=cpu(dw["ubase"][:m]), ibase=cpu(dw["ibase"]),
                excl_ptr=cpu(dw["excl_ptr"][: m + 1]), excl_idx=cpu(dw["excl_idx"][:e_end]),
                truth_ptr=cpu(dw["truth_ptr"][: m + 1]), truth_idx=cpu(dw["truth_idx"][:t_end]))
