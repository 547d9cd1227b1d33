"""Dev: the CSR top-K path over the neighbour counts SURVEY.md names for C5 (100 ... 1000 per row of W), K = 10.
python tools/time_c5_nnz.py [users]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
for nnz in (500, 700, 1000):
    w = synth.make_item("c5_rp3_ebnerd", n_users=nu, row_nnz=nnz)
    sc = HipScorer(0)
    sc.set_item_matrix_csr(w["m_ptr"], w["m_idx"], w["m_val"], w["n_items"], w["col_scale"])
    sc.set_history_csr(w["h_ptr"], w["h_idx"], w["h_val"])
    sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
    users = np.arange(nu, dtype=np.int32)
    entries = float(nnz) * float(np.diff(w["h_ptr"]).sum())
    for wave in (1, 0):
        sc.set_option("csr_wave", wave)
        sc.rank_topk(users, 10); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): sc.rank_topk(users, 10)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        print(f"{nnz} neighbours per row, csr_wave {wave}: {dt*1e3:.2f} ms per {nu} users = {nu/dt/1e6:.2f} M users/s, "
              f"{8 * entries / dt / 1e12:.2f} TB/s of algorithmic bytes", flush=True)
    sc.close()
    del w
