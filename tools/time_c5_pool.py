"""Dev: CSR top-K at pool sizes (32 < K <= 128) on the C5 workload: the fused kernel against dense rows + ifr_select_topk
(the route those K took before).  python tools/time_c5_pool.py [users]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
from recommenders_amd import engine as eng_mod
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
w = synth.make_item("c5_rp3_ebnerd", n_users=nu)
sc = HipScorer(0)
sc.set_item_matrix_csr(w["m_ptr"], w["m_idx"], w["m_val"], w["n_items"], w["col_scale"])
sc.set_history_csr(w["h_ptr"], w["h_idx"], w["h_val"])
sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
users = np.arange(nu, dtype=np.int32)
def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
for K in (32, 50, 100, 128):
    fused = timed(lambda: sc.rank_topk(users, K))
    ids_f, sc_f = sc.rank_topk(users, K)
    lim = eng_mod.MAX_FUSED_TOPK_ITEM_CSR
    eng_mod.MAX_FUSED_TOPK_ITEM_CSR = 0                      # the rows route
    try:
        rows = timed(lambda: sc.rank_topk(users, K), reps=1)
        ids_r, sc_r = sc.rank_topk(users, K)
    finally:
        eng_mod.MAX_FUSED_TOPK_ITEM_CSR = lim
    same = bool(torch.equal(ids_f, ids_r))
    print(f"K {K}: fused {fused*1e3:.2f} ms ({nu/fused/1e6:.2f} M users/s)   dense rows + select {rows*1e3:.2f} ms ({nu/rows/1e6:.2f} M users/s)"
          f"   ids equal {same}", flush=True)
