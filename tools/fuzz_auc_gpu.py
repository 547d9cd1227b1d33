"""Dev: randomised exact check of ifr_auc_dense / ifr_auc_dense_f64 (the bucket look-up table path and the binary
search path) against a NumPy pair count: ties (quantised scores), few / many positives, constant rows, +-inf and NaN
scores, exclusions.  python tools/fuzz_auc_gpu.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
bad = 0
for case in range(cases):
    rng = np.random.default_rng(seed0 * 1000 + case)
    n_users = int(rng.choice([1, 7, 60]))
    n_items = int(rng.choice([37, 1000, 4100, 26744]))
    f64 = bool(rng.integers(2))
    kind = rng.choice(["normal", "quantised", "constant", "tiny_range", "special"])
    mean_pos = float(rng.choice([1, 3, 6, 30, 300]))
    sc = HipScorer(0)
    sc.set_factors(np.zeros((n_users, 8), np.float32), np.zeros((n_items, 8), np.float32))
    pos = [np.sort(rng.choice(n_items, int(min(n_items // 2, max(0 if rng.random() < 0.1 else 1, rng.poisson(mean_pos)))), replace=False)) for _ in range(n_users)]
    exc = [np.setdiff1d(rng.choice(n_items, int(min(n_items // 3, rng.poisson(rng.choice([0, 5, 200])))), replace=False), p) for p in pos]
    t_ptr = np.cumsum([0] + [len(p) for p in pos]).astype(np.int64)
    e_ptr = np.cumsum([0] + [len(p) for p in exc]).astype(np.int64)
    sc.set_truth_csr(t_ptr, (np.concatenate(pos) if t_ptr[-1] else np.zeros(0)).astype(np.int32))
    sc.set_exclusion_csr(e_ptr, (np.concatenate(exc) if e_ptr[-1] else np.zeros(0)).astype(np.int32))
    rows = rng.normal(size=(n_users, n_items))
    if kind == "quantised":
        rows = np.round(rows * 3) / 3
    elif kind == "constant":
        rows[:] = 1.25
    elif kind == "tiny_range":
        rows = 1.0 + rows * 1e-7
    elif kind == "special":
        m = rng.random(rows.shape)
        rows[m < 0.01] = np.inf; rows[(m >= 0.01) & (m < 0.02)] = -np.inf; rows[(m >= 0.02) & (m < 0.03)] = np.nan
        for u, p in enumerate(pos):                      # (positives keep finite scores: their order defines the search)
            rows[u, p] = rng.normal(size=len(p))
    dt = np.float64 if f64 else np.float32
    rows = rows.astype(dt)
    ld = (n_items + 31) // 32 * 32
    buf = torch.zeros((n_users, ld), dtype=torch.float64 if f64 else torch.float32, device=dev)
    buf[:, :n_items] = torch.as_tensor(rows, device=dev)
    users = np.arange(n_users, dtype=np.int32)
    got = sc.auc_dense(users, buf).cpu().numpy()
    ok = True
    for u in range(n_users):
        if len(pos[u]) == 0:
            ok &= bool(np.isnan(got[u])); continue
        neg = np.setdiff1d(np.arange(n_items), np.r_[pos[u], exc[u]])
        sp, sn = rows[u, pos[u]], rows[u, neg]
        with np.errstate(invalid="ignore"):
            wins = int((sp[:, None] > sn[None, :]).sum())
        exp = wins / (len(sp) * float(len(sn))) if len(sn) else np.nan
        ok &= bool(got[u] == exp or (np.isnan(got[u]) and np.isnan(exp)))
    bad += not ok
    print(f"case {case}: users {n_users} items {n_items} {'f64' if f64 else 'f32'} {kind} pos {mean_pos} {'ok' if ok else 'MISMATCH'}", flush=True)
    sc.close()
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
