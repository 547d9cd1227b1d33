"""Dev: the CSR top-K kernel on the C5 workload with every history drawn from 64 rows of W (L2-resident entries):
how much of the kernel's time is global-memory latency."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
nu = 262144
w = synth.make_item("c5_rp3_ebnerd", n_users=nu)
for hot in (False, True):
    h_idx = w["h_idx"]
    if hot:
        h_idx = ((h_idx.astype(np.int64) % 64) * 1900).astype(np.int32)
    sc = HipScorer(0)
    sc.set_item_matrix_csr(w["m_ptr"], w["m_idx"], w["m_val"], w["n_items"], w["col_scale"])
    sc.set_history_csr(w["h_ptr"], h_idx, w["h_val"])
    sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
    users = np.arange(nu, dtype=np.int32)
    sc.rank_topk(users, 10); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): sc.rank_topk(users, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"hot rows {hot}: {dt*1e3:.2f} ms per {nu} users = {dt/nu*1e9:.0f} ns/user", flush=True)
    sc.close()
