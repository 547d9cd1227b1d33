"""Dev: how the CSR top-K kernel's time per user scales with the number of 4096-column windows (catalogue size) at a
fixed number of W entries per user."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
nu = 262144
for ni, nnz in [(125000, 256), (62500, 256), (31250, 256), (125000, 64), (125000, 16)]:
    w = synth.make_item("c5_rp3_ebnerd", n_users=nu, n_items=ni, row_nnz=nnz)
    sc = HipScorer(0)
    sc.set_item_matrix_csr(w["m_ptr"], w["m_idx"], w["m_val"], w["n_items"], w["col_scale"])
    sc.set_history_csr(w["h_ptr"], w["h_idx"], w["h_val"])
    sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
    users = np.arange(nu, dtype=np.int32)
    sc.rank_topk(users, 10); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): sc.rank_topk(users, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"items {ni} ({(ni + 4095) // 4096} windows), {nnz} entries/row: {dt*1e3:.2f} ms per {nu} users = {dt/nu*1e9:.0f} ns/user", flush=True)
    sc.close()
