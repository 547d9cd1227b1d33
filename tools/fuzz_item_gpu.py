"""Dev: randomised differential test of the item-item entry points (dense fp32, dense fp64, CSR) against the C oracle.
python tools/fuzz_item_gpu.py [cases] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from recommenders_amd import HipScorer
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
for case in range(cases):
    rng = np.random.default_rng(seed0 * 1000 + case)
    form = rng.choice(["dense", "dense_f64", "csr"])
    n_users = int(rng.choice([1, 33, 200, 700]))
    n_items = int(rng.choice([40, 1000, 4097, 9000, 20000])) if form == "csr" else int(rng.choice([40, 1000, 4097, 9000]))
    K = int(rng.choice([1, 5, 10, 20, 32]))
    if form == "csr" and rng.integers(2):
        K = int(rng.choice([33, 50, 64, 65, 100, 128]))
    mean_hist = float(rng.choice([0.5, 5, 40, 90, 150, 220, 300]))
    pen = bool(rng.integers(2))
    neg = bool(rng.integers(2))
    lens = np.minimum(rng.poisson(mean_hist, n_users), n_items)
    rows = [np.sort(rng.choice(n_items, int(l), replace=False)) for l in lens]
    h_ptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)
    h_idx = (np.concatenate(rows) if h_ptr[-1] else np.zeros(0)).astype(np.int32)
    h_val = (rng.integers(1, 6, len(h_idx)).astype(np.float32) if rng.integers(2) else rng.random(len(h_idx)).astype(np.float32))
    col_scale = (1.0 / rng.integers(1, 500, n_items) ** 0.7).astype(np.float32) if pen else None
    if pen and rng.integers(2) and form != "csr":                     # (a CSR matrix with entries in such a column is rejected at set time)
        col_scale[rng.integers(0, n_items, 3)] = np.inf                  # zero-degree items: 0 * inf = NaN, dropped
    erows = [np.sort(rng.choice(n_items, int(min(n_items, rng.poisson(rng.choice([0, 3, 400])))), replace=False)) for _ in range(n_users)]
    e_ptr = np.cumsum([0] + [len(r) for r in erows]).astype(np.int64)
    e_idx = (np.concatenate(erows) if e_ptr[-1] else np.zeros(0)).astype(np.int32)
    sc = HipScorer(0)
    users = np.r_[rng.permutation(n_users), [n_users + 2]].astype(np.int32)
    if form == "csr":
        row_nnz = float(rng.choice([2, 30, 300]))
        M = sp.random(n_items, n_items, density=min(1.0, row_nnz / n_items), format="csr", dtype=np.float32,
                      random_state=np.random.RandomState(case), data_rvs=lambda n: (rng.random(n).astype(np.float32) * 1e-2))
        if neg:
            M.data[::3] *= -1
        M.sort_indices()
        sc.set_item_matrix_csr(M.indptr.astype(np.int64), M.indices, M.data, n_items, col_scale)
    else:
        Md = np.maximum(0, rng.normal(0, 1e-2, (n_items, n_items))).astype(np.float32)
        if neg:
            Md -= np.float32(0.002) * (rng.random((n_items, n_items)) < 0.1)
        if form == "dense_f64":
            M64 = Md.astype(np.float64) * (1.0 + 1e-9 * rng.standard_normal(Md.shape))
            sc.set_item_matrix_dense(M64, col_scale, precision="f64")
        else:
            sc.set_item_matrix_dense(Md, col_scale)
    sc.set_history_csr(h_ptr, h_idx, h_val)
    sc.set_exclusion_csr(e_ptr, e_idx)
    sc.set_truth_csr(np.zeros(n_users + 1, np.int64), np.zeros(0, np.int32))
    if form == "csr" and rng.integers(2):
        sc.set_option("csr_cand_cap", int(rng.choice([16, 40, 100, 256])))
    if form == "csr":
        sc.set_option("csr_wave", int(rng.choice([0, 2])))              # the wave-per-user kernel forced (+ hand-over), or the workgroup kernel alone
    t = time.perf_counter()
    ids, scores = sc.rank_topk(users, K)
    ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
    for again in range(6):                                           # the same launch again: races show as run-to-run differences
        i2, s2 = sc.rank_topk(users, K)
        if not (np.array_equal(i2.cpu().numpy(), ids) and np.array_equal(s2.cpu().numpy(), scores, equal_nan=True)):
            print('   RUN-TO-RUN DIFFERENCE in repeat', again, flush=True)
            bad += 1
    ok = True
    if form == "dense_f64":
        rows32 = sc.score_dense(users).cpu().numpy()
        for i, u in enumerate(users):
            if u < n_users:
                o64, o32 = orc.c_score_item_dense_f64(M64, h_idx[h_ptr[u]:h_ptr[u + 1]], h_val[h_ptr[u]:h_ptr[u + 1]], col_scale)
                ex = e_idx[e_ptr[u]:e_ptr[u + 1]]
            else:
                o32, ex = np.zeros(n_items, np.float32) * (1.0 if col_scale is None else col_scale), np.zeros(0, np.int32)
            if u >= n_users:
                o64 = o32.astype(np.float64)
            keep = ~np.isnan(o32)
            ok &= np.array_equal(rows32[i], o32, equal_nan=True)
            # the float64 sums are what is ranked (canonical order: sum descending, item ascending); NaN scores never
            # rank (dropped like exclusions); the list's scores are the winners rounded to fp32
            cand = np.setdiff1d(np.flatnonzero(keep), ex).astype(np.int64)
            head = orc.canonical_rank(cand, (o64[cand] + 0.0), K)
            oi = np.full(K, -1, np.int32); oi[: len(head)] = head
            os_ = np.full(K, -np.inf, np.float32); os_[: len(head)] = o64[head].astype(np.float32)
            ok &= np.array_equal(ids[i], oi) and np.array_equal(scores[i], os_)
    else:
        shift = sc.item_fixed_point_shift() if form == "csr" else 0
        m_csr = (M.indptr, M.indices, M.data) if form == "csr" else None
        o_ids, o_sc, _, _ = orc.c_ranking_eval_item(None if form == "csr" else Md, m_csr, shift, n_items, col_scale, h_ptr, h_idx, h_val,
                                                    users, e_ptr, e_idx, np.zeros(n_users + 1, np.int64), np.zeros(0, np.int32),
                                                    [K], False, K, n_threads=8)
        ok = np.array_equal(ids, o_ids) and np.array_equal(scores, o_sc, equal_nan=True)
    bad += not ok
    print(f"case {case}: {form} users {n_users} items {n_items} K {K} hist {mean_hist} pen {pen} neg {neg} "
          f"{'ok' if ok else 'MISMATCH'} {time.perf_counter() - t:.1f}s", flush=True)
    if not ok and form != "dense_f64":
        r = np.flatnonzero((ids != o_ids).any(axis=1))[:3]
        for i in r:
            print("   row", i, "user", users[i], "\n    got", ids[i][:12], scores[i][:6], "\n    exp", o_ids[i][:12], o_sc[i][:6])
    sc.close()
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
