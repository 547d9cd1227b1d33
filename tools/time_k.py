"""Dev: fused top-10 throughput at C2 shape for several factor dimensions (KD = 32 / 64 / 128 / 256 instantiations)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
nu, ni = 138493, 26744
rng = np.random.default_rng(1)
ptr, idx, _, _ = synth.interactions(rng, nu, ni, 145)
users = torch.arange(nu, dtype=torch.int32, device="cuda")
for k in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["50", "128", "200", "256"])]:
    U = rng.normal(0, 0.1, (nu, k)).astype(np.float32); V = rng.normal(0, 0.1, (ni, k)).astype(np.float32)
    sc = HipScorer(0)
    sc.set_factors(U, V, None, None)
    sc.set_exclusion_csr(ptr, idx)
    sc.rank_topk(users, 10); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): sc.rank_topk(users, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f"k={k}: {dt*1e3:.2f} ms  {nu/dt/1e6:.2f} M users/s  {2*nu*ni*k/dt/1e12:.1f} TFLOP/s algorithmic", flush=True)
    sc.close()
