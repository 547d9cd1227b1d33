"""Dev: throughput of the k = -1 path (dense rows + full ranking + list metrics) at C2 size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
nu = 39568
w = synth.make("c2_wmf_ml20m", n_users=nu)
sc = HipScorer(0)
sc.set_factors(w["U"], w["V"], w["ubase"], w["ibase"])
sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
step = (1 << 28) // w["n_items"]
users = torch.arange(nu, dtype=torch.int32, device="cuda")
def run():
    for b in range(0, nu, step):
        bu = users[b:b + step]
        dense = sc.score_dense(bu)
        ranked, n_cand = sc.rank_full(bu, dense)
        pl, _, _ = sc.eval_ranked_metrics(bu, ranked, n_cand, [10, -1])
    torch.cuda.synchronize()
run()
for name in ("all",):
    t = time.perf_counter(); run(); dt = time.perf_counter() - t
    print(f"dense + rank_full + list metrics: {dt*1e3:.1f} ms for {nu} users = {nu/dt/1e3:.1f} K users/s", flush=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
bu = users[:step]
ev[0].record(); dense = sc.score_dense(bu); ev[1].record(); ranked, n_cand = sc.rank_full(bu, dense); ev[2].record()
pl, _, _ = sc.eval_ranked_metrics(bu, ranked, n_cand, [10, -1]); ev[3].record(); torch.cuda.synchronize()
print("one batch of", step, "users: dense %.2f ms, rank_full %.2f ms, metrics %.2f ms" % (ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])))
for K in (100, 1000):
    sc.rank_topk(users, K); torch.cuda.synchronize()
    t = time.perf_counter(); ids, _ = sc.rank_topk(users, K); nr = (ids >= 0).sum(dim=1).to(torch.int32)
    pl, _, _ = sc.eval_ranked_metrics(users, ids, nr, [K]); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"dense + select_topk(K={K}) + list metrics: {dt*1e3:.1f} ms for {nu} users = {nu/dt/1e3:.1f} K users/s", flush=True)
bu = users[:step]
dense = sc.score_dense(bu)
for K in (100, 1000):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    sc.select_topk(bu, dense, K); torch.cuda.synchronize()
    ev[0].record(); ids, _ = sc.select_topk(bu, dense, K); ev[1].record()
    nr = (ids >= 0).sum(dim=1).to(torch.int32); pl, _, _ = sc.eval_ranked_metrics(bu, ids, nr, [K]); ev[2].record(); torch.cuda.synchronize()
    print(f"K={K}: select_topk {ev[0].elapsed_time(ev[1]):.2f} ms, count + list metrics {ev[1].elapsed_time(ev[2]):.2f} ms per {step} users", flush=True)
def run2():
    for b in range(0, nu, step):
        bu = users[b:b + step]
        dense = sc.score_dense(bu)
        pl, _, _ = sc.eval_dense_metrics(bu, dense, [10, -1])
    torch.cuda.synchronize()
run2()
t = time.perf_counter(); run2(); dt = time.perf_counter() - t
print(f"dense + eval_dense_metrics (k = 10, -1, MRR, MAP): {dt*1e3:.1f} ms for {nu} users = {nu/dt/1e3:.1f} K users/s", flush=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
bu = users[:step]; dense = sc.score_dense(bu)
ev[0].record(); sc.eval_dense_metrics(bu, dense, [10, -1]); ev[1].record(); torch.cuda.synchronize()
print(f"eval_dense_metrics alone: {ev[0].elapsed_time(ev[1]):.2f} ms per {step} users")
