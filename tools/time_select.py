"""Dev: ifr_select_topk (partial ranking of dense rows) at C2-shaped rows and at 1 M-item rows; tie-heavy rows are
checked against the full ranking (ifr_rank_full) of the same rows."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
rng = np.random.default_rng(3)
for nu, ni, K in [(16384, 26744, 100), (16384, 26744, 1000), (512, 1000000, 100)]:
    ptr, idx, _, _ = synth.interactions(rng, nu, ni, 100)
    sc = HipScorer(0)
    sc.set_factors(rng.normal(0, 0.1, (nu, 32)).astype(np.float32), rng.normal(0, 0.1, (ni, 32)).astype(np.float32), None, None)
    sc.set_exclusion_csr(ptr, idx)
    users = torch.arange(nu, dtype=torch.int32, device="cuda")
    dense = sc.score_dense(users)
    sc.select_topk(users, dense, K); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): ids, scores = sc.select_topk(users, dense, K)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"{nu} rows x {ni} items, K={K}: {dt*1e3:.2f} ms  {nu*ni*4/dt/1e9:.0f} GB/s of rows (one read)", flush=True)
    # ties: quantised scores (a few hundred distinct values), and constant rows
    m = min(nu, 256)
    q = torch.round(dense[:m] * 20) / 20
    q[:8] = 0.25
    ranked, n_cand = sc.rank_full(users[:m], q)
    ids, scores = sc.select_topk(users[:m], q, K)
    ok = bool((ids.cpu() == ranked[:, :K].cpu()).all())
    print("   tie-heavy rows equal the head of the full ranking:", ok, flush=True)
    sc.close()
    del dense
