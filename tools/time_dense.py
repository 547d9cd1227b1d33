"""Dev: dense score rows from ifr_score_dense vs from the fused kernel's STORE variant (ifr_rank_topk_dense, K = 1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
w = synth.make("c2_wmf_ml20m")
sc = HipScorer(0)
sc.set_factors(w["U"], w["V"], w["ubase"], w["ibase"]); sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"]); sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
ld = (w["n_items"] + 31) // 32 * 32
for nb in (10035, 40000, w["n_users"]):
    users = torch.arange(nb, dtype=torch.int32, device="cuda")
    buf = torch.empty((nb, ld), dtype=torch.float32, device="cuda")
    for name, fn in (("score_dense", lambda: sc.score_dense(users, buf[:, : w["n_items"]])),
                     ("rank_topk_dense K=1", lambda: sc.rank_topk_dense(users, 1, buf)),
                     ("eval_dense_metrics", lambda: sc.eval_dense_metrics(users, buf[:, : w["n_items"]], [-1, 100]))):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        print(f"n={nb:6d} {name:22s} {dt*1e3:8.3f} ms  {dt/nb*1e9:7.1f} ns/user", flush=True)
