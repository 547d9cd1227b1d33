"""Dev: per-user facade calls (the reference's evaluator loops call model.rank / model.score one user at a time)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from recommenders_amd import synth
from recommenders_amd.data import Dataset
from recommenders_amd.models import WMF
nu = 20000
w = synth.make("c2_wmf_ml20m", n_users=nu)
ni = w["n_items"]
tr = sp.csr_matrix((np.ones(len(w["excl_idx"]), np.float32), w["excl_idx"], w["excl_ptr"]), shape=(nu, ni))
train_set = Dataset.from_csr(tr)
model = WMF(k=w["k"], trainable=False, init_params={"U": w["U"], "V": w["V"]}).fit(train_set)
model.rank(0, k=10)
import cProfile, pstats
for name, fn in [("rank(u, k=10)", lambda u: model.rank(u, k=10)), ("rank(u) full", lambda u: model.rank(u)),
                 ("rank(u, item_indices, k=10)", lambda u: model.rank(u, item_indices=cand, k=10)), ("score(u)", lambda u: model.score(u))]:
    cand = np.setdiff1d(np.arange(ni), tr[0].indices)
    n = 1000
    t = time.perf_counter()
    for u in range(n):
        fn(u)
    dt = time.perf_counter() - t
    print(f"{name}: {dt / n * 1e6:.0f} us per call, {n / dt:.0f} users/s", flush=True)
pr = cProfile.Profile(); pr.enable()
for u in range(300): model.rank(u, k=10)
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
