"""Dev: repeat the two-region grid case and report any run whose lists differ from the oracle's."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_gpu_parity import _workload, _scorer, _oracle
n_users = 128 * 552 + 17
w = _workload(6, n_users, 2048, 32, True, mean=12)
users = np.arange(n_users, dtype=np.int32)
sc = _scorer(w)
o_ids, o_sc, _, _ = _oracle(w, users, [10], 10, sc.chain_order())
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for tb, ts in ((1, 1), (1, 0), (4096, 1)):
        sc.set_option("target_blocks", tb); sc.set_option("tail_split", ts)
        ids, scores = sc.rank_topk(users, 10)
        ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
        r = np.flatnonzero((ids != o_ids).any(axis=1) | (scores != o_sc).any(axis=1))
        if len(r):
            bad += 1
            print(f"rep {rep} target_blocks {tb} tail_split {ts}: {len(r)} rows differ, first rows {r[:8]} (user tiles {np.unique(r // 128)[:10]})", flush=True)
            i = r[0]
            print("   got", ids[i], scores[i][:4], "\n   exp", o_ids[i], o_sc[i][:4], flush=True)
print("bad runs:", bad)
