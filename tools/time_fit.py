"""Dev: time ifr_fit_ease on a C4-shaped click matrix: python tools/time_fit.py <n_items> <n_users>."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
ni = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nu = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
rng = np.random.default_rng(4)
ptr, idx, _, _ = synth.interactions(rng, nu, ni, 30, lo=5, hi=400)
val = np.ones(len(idx), np.float32)
print(f"X: {nu} users x {ni} items, nnz {len(idx)}, pairs {int((np.diff(ptr).astype(np.int64) ** 2).sum() // 2)}", flush=True)
sc = HipScorer(0)
for rep in range(2):
    t = time.perf_counter()
    sc.fit_ease(ptr, idx, val, ni, 500.0, True)
    torch.cuda.synchronize()
    print(f"fit_ease run {rep}: {time.perf_counter() - t:.2f} s", flush=True)
sc.set_history_csr(ptr, idx, val)
ids, scores = sc.rank_topk(np.arange(8, dtype=np.int32), 10)
print(ids[:2].cpu().numpy(), scores[:2].cpu().numpy(), flush=True)
