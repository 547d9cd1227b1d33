import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
from oracle import oracle as orc
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ni = int(sys.argv[2]) if len(sys.argv) > 2 else None
w = synth.make_item("c5_rp3_ebnerd", n_users=nu, n_items=ni)
sc = HipScorer(0)
sc.set_item_matrix_csr(w["m_ptr"], w["m_idx"], w["m_val"], w["n_items"], w["col_scale"])
sc.set_history_csr(w["h_ptr"], w["h_idx"], w["h_val"])
sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
users = np.arange(nu, dtype=np.int32)
ids, scores = sc.rank_topk(users, 10)
ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
shift = sc.item_fixed_point_shift()
print("shift", shift, orc.item_fixed_point_shift(w["m_val"], w["h_ptr"], w["h_val"]))
o_ids, o_sc, _, _ = orc.c_ranking_eval_item(None, (w["m_ptr"], w["m_idx"], w["m_val"]), shift, w["n_items"], w["col_scale"],
    w["h_ptr"], w["h_idx"], w["h_val"], users, w["excl_ptr"], w["excl_idx"], np.zeros(nu+1, np.int64), np.zeros(0, np.int32), [10], False, 10, n_threads=16)
bad = np.flatnonzero((ids != o_ids).any(1))
print("mismatching users", len(bad), "of", nu)
rows = sc.score_dense(users[:256]).cpu().numpy()
for u in bad[:3]:
    print("user", u, "hist", w["h_ptr"][u+1]-w["h_ptr"][u])
    print(" gpu", ids[u], scores[u]); print(" cpu", o_ids[u], o_sc[u])
    if u < 256:
        hi = w["h_idx"][w["h_ptr"][u]: w["h_ptr"][u + 1]]; hv = w["h_val"][w["h_ptr"][u]: w["h_ptr"][u + 1]]
        ref = orc.c_score_item_csr(w["m_ptr"], w["m_idx"], w["m_val"], w["n_items"], hi, hv, shift, w["col_scale"])
        d = np.flatnonzero(rows[u] != ref)
        print(" dense-row diffs", len(d), d[:10], rows[u][d[:5]], ref[d[:5]])
