"""Dev: the full C3 shard (1.25 M users x 1 M items) ranked under different launch plans must give identical lists
(placement independence at the target size), and K = 10 must be the head of K = 20 / K = 100."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
dw = synth.make_c3_device(n_users=nu, device=torch.device("cuda", 0))
sc = HipScorer(0)
sc.set_factors(dw["U"], dw["V"], dw["ubase"], dw["ibase"])
sc.set_exclusion_csr(dw["excl_ptr"], dw["excl_idx"])
users = torch.arange(nu, dtype=torch.int32, device="cuda")
def run(K, **opts):
    for k, v in opts.items():
        sc.set_option(k, v)
    t = time.perf_counter()
    ids, scores = sc.rank_topk(users, K)
    torch.cuda.synchronize()
    print(f"  K {K} {opts}: {time.perf_counter() - t:.2f} s, redone users so far {sc.wide_overflows()}", flush=True)
    return ids, scores
ref_ids, ref_sc = run(10, chunks=0, tail_split=1)
bad = 0
for opts in (dict(chunks=0, tail_split=0), dict(chunks=2, tail_split=1), dict(chunks=3, tail_split=1)):
    ids, scs = run(10, **opts)
    ok = torch.equal(ids, ref_ids) and torch.equal(scs, ref_sc)
    bad += not ok
    print("   identical to the default plan:", ok, flush=True)
sc.set_option("chunks", 0)
for K in (20, 100):
    ids, scs = run(K)
    ok = torch.equal(ids[:, :10], ref_ids) and torch.equal(scs[:, :10], ref_sc)
    bad += not ok
    print(f"   head of K = {K} equals the K = 10 lists:", ok, flush=True)
print("inconsistencies:", bad)
sys.exit(1 if bad else 0)
