"""Dev: time ifr_fit_rp3beta on an EB-NeRD-shaped click pattern: python tools/time_fit_rp3.py <n_items> <n_users>."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
ni = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
nu = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
rng = np.random.default_rng(5)
ptr, idx, _, _ = synth.interactions(rng, nu, ni, 31, lo=5, hi=400)
print(f"A: {nu} users x {ni} items, nnz {len(idx)}", flush=True)
sc = HipScorer(0)
for rep in range(2):
    t = time.perf_counter()
    w = sc.fit_rp3beta(ptr, idx, ni, 0.7)
    torch.cuda.synchronize()
    print(f"fit_rp3beta run {rep}: {time.perf_counter() - t:.2f} s, nnz(W) = {w} ({w / ni:.0f} per row)", flush=True)
sc.set_exclusion_csr(ptr, idx)
for dbg in [int(a) for a in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0"])]:
    sc.set_option("debug_mode", dbg)
    for rep in range(2):
        t = time.perf_counter()
        ids, scores = sc.rank_topk(np.arange(min(nu, 65536), dtype=np.int32), 10)
        torch.cuda.synchronize()
    print(f"debug_mode {dbg}: rank_topk 65536 users: {time.perf_counter() - t:.3f} s", ids[0].cpu().numpy(), flush=True)
