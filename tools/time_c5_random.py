"""Dev: the CSR top-K kernels on C5's shape with UNSTRATIFIED columns (every row's 256 columns drawn uniformly: Poisson
counts per window instead of the synthetic workload's even spread) -- segments longer than the prefetch are common."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
nu = 262144
w = synth.make_item("c5_rp3_ebnerd", n_users=nu)
ni, nnz = w["n_items"], 256
rng = np.random.default_rng(5)
cols = np.sort(rng.integers(0, ni, (ni, nnz), dtype=np.int64), axis=1)
# make rows strictly ascending: bump duplicates (rare) forward
for _ in range(4):
    dup = cols[:, 1:] <= cols[:, :-1]
    if not dup.any():
        break
    cols[:, 1:][dup] = cols[:, :-1][dup] + 1
    cols = np.sort(np.minimum(cols, ni - 1), axis=1)
ok = (cols[:, 1:] > cols[:, :-1]).all(axis=1)
print("rows strictly ascending:", ok.mean())
cols[~ok] = (np.arange(nnz)[None, :] * (ni // nnz) + 1)[: 1]      # (fallback for the few rows that still collide)
m_idx = cols.astype(np.int32).ravel()
m_ptr = np.arange(ni + 1, dtype=np.int64) * nnz
m_val = w["m_val"]
users = np.arange(nu, dtype=np.int32)
for name, idx in (("stratified", w["m_idx"]), ("uniform", m_idx)):
    for wave in (1, 0, 2):
        sc = HipScorer(0)
        sc.set_item_matrix_csr(m_ptr, idx, m_val, ni, w["col_scale"])
        sc.set_history_csr(w["h_ptr"], w["h_idx"], w["h_val"])
        sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
        sc.set_option("csr_wave", wave)
        sc.rank_topk(users, 10); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): sc.rank_topk(users, 10)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        print(f"{name} columns, csr_wave {wave}: {dt*1e3:.2f} ms per {nu} users = {nu/dt/1e6:.2f} M users/s", flush=True)
        sc.close()
