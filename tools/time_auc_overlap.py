"""Dev: the AUC variant at C2 shape, serial (one batch: fused kernel writes the rows, then the AUC kernel reads them)
against PIPELINED batches on two streams (the AUC kernel of batch b runs while the fused kernel scores batch b + 1).
python tools/time_auc_overlap.py [batches,...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
dev = torch.device("cuda:0")
w = synth.make("c2_wmf_ml20m", seed=synth.CONFIGS["c2_wmf_ml20m"][4])
sc = HipScorer(0)
sc.set_factors(w["U"], w["V"], w["ubase"], w["ibase"])
sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
n, K = w["n_users"], 10
users = torch.arange(n, dtype=torch.int32, device=dev)
ld = (w["n_items"] + 31) // 32 * 32
ids = torch.empty((n, K), dtype=torch.int32, device=dev)
scores = torch.empty((n, K), dtype=torch.float32, device=dev)
auc_out = torch.empty((n,), dtype=torch.float64, device=dev)
buf = torch.empty((n, ld), dtype=torch.float32, device=dev)
side = torch.cuda.Stream(dev)

def serial():
    _, _, dense = sc.rank_topk_dense(users, K, buf, ids, scores)
    auc_out[:] = sc.auc_dense(users, dense)

def pipelined(nb):
    step = (n + nb - 1) // nb
    main = torch.cuda.current_stream(dev)
    evs = []
    for b in range(0, n, step):
        ub = users[b:b + step]
        _, _, dense = sc.rank_topk_dense(ub, K, buf[b:b + step], ids[b:b + step], scores[b:b + step])
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            auc_out[b:b + step] = sc.auc_dense(ub, dense)
    done = torch.cuda.Event(); done.record(side)
    main.wait_event(done)

def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3

serial(); torch.cuda.synchronize()
ref = auc_out.clone(); ref_ids = ids.clone()
print(f"serial, one batch: {timed(serial):.2f} ms per pass", flush=True)
for nb in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["2", "4", "8"])]:
    ms = timed(lambda: pipelined(nb))
    same = bool(torch.equal(torch.nan_to_num(auc_out), torch.nan_to_num(ref)) and torch.equal(ids, ref_ids))
    print(f"pipelined, {nb} batches, AUC on a second stream: {ms:.2f} ms per pass   same results {same}", flush=True)
