"""Dev: wall time of the facade ranking_eval (the reference's call) at C2 scale next to the raw engine loop."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from recommenders_amd import synth, ranking_eval, metrics as M
from recommenders_amd.data import Dataset
from recommenders_amd.models import WMF
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 138493
w = synth.make("c2_wmf_ml20m", n_users=nu)
ni = w["n_items"]
tr = sp.csr_matrix((np.ones(len(w["excl_idx"]), np.float32), w["excl_idx"], w["excl_ptr"]), shape=(nu, ni))
te = sp.csr_matrix((np.ones(len(w["truth_idx"]), np.float32), w["truth_idx"], w["truth_ptr"]), shape=(nu, ni))
t = time.perf_counter(); train_set, test_set = Dataset.from_csr(tr), Dataset.from_csr(te); print(f"datasets {time.perf_counter()-t:.2f} s", flush=True)
model = WMF(k=w["k"], trainable=False, init_params={"U": w["U"], "V": w["V"]}).fit(train_set)
mets = [M.NDCG(k=10), M.Recall(k=10), M.Precision(k=10)]
# one-time process costs (HIP context, code-object load, the engine's upload of the factors) are not the evaluation's
t = time.perf_counter(); torch.zeros(1, device="cuda").sum().item(); print(f"HIP context {time.perf_counter()-t:.2f} s", flush=True)
t = time.perf_counter(); eng = model._hip_engine(); eng.rank_topk(np.arange(64, dtype=np.int32), 10); torch.cuda.synchronize()
print(f"engine (factors uploaded, kernels loaded) {time.perf_counter()-t:.2f} s", flush=True)
import cProfile, pstats
for rep in range(3):
    model.ranked_items.clear() if hasattr(model.ranked_items, "clear") else None
    if rep == 0 and len(sys.argv) > 2:
        pr = cProfile.Profile(); pr.enable()
        ranking_eval(model, mets, train_set, test_set, verbose=False); torch.cuda.synchronize()
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
        continue
    t = time.perf_counter()
    avg, per_user = ranking_eval(model, mets, train_set, test_set, verbose=False)
    torch.cuda.synchronize()
    print(f"ranking_eval run {rep}: {time.perf_counter()-t:.3f} s  {nu/(time.perf_counter()-t)/1e6:.2f} M users/s  avg {avg}", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
ranking_eval(model, mets, train_set, test_set, verbose=False); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
for name, ms in [("list metrics [NDCG(-1), Recall(100), MRR, MAP]", [M.NDCG(k=-1), M.Recall(k=100), M.MRR(), M.MAP()]),
                 ("AUC + NDCG@10", [M.AUC(), M.NDCG(k=10)])]:
    for rep in range(2):
        t = time.perf_counter()
        avg, _ = ranking_eval(model, ms, train_set, test_set, verbose=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(f"{name}: {dt:.3f} s  {nu/dt/1e6:.2f} M users/s  avg {[round(a, 6) for a in avg]}", flush=True)
# per-user access to the lazy result caches (block materialisation)
avg, _ = ranking_eval(model, mets, train_set, test_set, verbose=False)
us = list(model.ranked_items.keys())[:8192]
t = time.perf_counter()
tot = 0
for u in us:
    tot += len(model.ranked_items[u]) + len(model.item_scores[u]) + len(model.item_scores_mapped_indices[u])
dt = time.perf_counter() - t
pr = cProfile.Profile(); pr.enable()
for u in list(model.ranked_items.keys())[8192:8192 + 4096]:
    tot += len(model.ranked_items[u])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
print(f"lazy caches: {len(us)} users read in {dt:.3f} s = {1e6 * dt / len(us):.1f} us per user ({tot} entries)", flush=True)
