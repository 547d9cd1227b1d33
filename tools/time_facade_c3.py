"""Dev: first-call / repeated wall time of the facade ranking_eval at one GPU's shard of the 10 M x 1 M configuration
(1.25 M users x 1 M items, 75 M ratings held in device memory), next to the fused kernel alone."""
import sys, os, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import DeviceCsr, HipScorer, synth, ranking_eval, metrics as M
from recommenders_amd import eval as ev
from recommenders_amd.data import DeviceDataset

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
dev = torch.device("cuda:0")
w = synth.make_c3_device(n_users=nu, device=dev)
ni = w["n_items"]
ones = lambda t: torch.ones(t.numel(), dtype=torch.float32, device=dev)
train = DeviceDataset(DeviceCsr(w["excl_ptr"].to(dev), w["excl_idx"], ones(w["excl_idx"]), (nu, ni)))
test = DeviceDataset(DeviceCsr(w["truth_ptr"].to(dev), w["truth_idx"], ones(w["truth_idx"]), (nu, ni)))
sc = HipScorer(0)
sc.set_factors(w["U"], w["V"], w["ubase"], w["ibase"])


class Fitted:                      # the facade only needs a fitted model; the engine is handed over
    is_fitted = True


mets = [M.NDCG(k=10), M.Recall(k=10), M.Precision(k=10)]
users = torch.arange(nu, dtype=torch.int32, device=dev)
sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"]); sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
sc.rank_topk(users[:4096], 10); torch.cuda.synchronize()
t = time.perf_counter(); ids, _ = sc.rank_topk(users, 10); torch.cuda.synchronize(); t_kernel = time.perf_counter() - t
print(f"fused kernel alone: {t_kernel:.3f} s", flush=True)
del ids
ev.clear_eval_cache()
for rep in range(3):
    t = time.perf_counter()
    avg, per = ranking_eval(Fitted(), mets, train, test, rating_threshold=1.0, _engine=sc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"ranking_eval call {rep}: {dt:.3f} s = {dt / t_kernel:.2f} x the kernel; avg {[round(a, 6) for a in avg]}; "
          f"host max RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.2f} GB", flush=True)
t = time.perf_counter(); v = per[0][int(nu // 2)]; print(f"first read of user_results: {time.perf_counter() - t:.3f} s (value {v:.4f})")
# the builder alone
ups = [train.csr_matrix, None, test.csr_matrix]
torch.cuda.synchronize(); t = time.perf_counter()
st = sc.build_eval_csrs(ups[0], ups[1], ups[2], 1.0, ni, 0, nu); torch.cuda.synchronize()
print(f"ifr_build_eval_csrs alone: {time.perf_counter() - t:.3f} s ({st['truth_nnz']} truth + {st['excl_nnz']} exclusion entries)")
