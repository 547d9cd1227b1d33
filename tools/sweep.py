"""Dev tool: time the fused kernel under option sweeps on the C2 workload (GPU box).
The debug_mode ablations (bits 0-2) need a library built with `make -C recommenders_amd/csrc ABLATE=1`."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth

def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2_wmf_ml20m"
    nu = int(sys.argv[2]) if len(sys.argv) > 2 else None
    ni = int(sys.argv[3]) if len(sys.argv) > 3 else None
    w = synth.make(cfg, n_users=nu, n_items=ni)
    sc = HipScorer(0)
    sc.set_factors(w["U"], w["V"], w["ubase"], w["ibase"])
    sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
    n = w["n_users"]
    users = torch.arange(n, dtype=torch.int32, device="cuda")
    flops = 2.0 * n * w["n_items"] * w["k"]
    def run(tag, reps=8):
        for _ in range(2): sc.rank_topk(users, 10)
        torch.cuda.synchronize()
        sc.profile_enable(True)
        for _ in range(reps): sc.rank_topk(users, 10)
        torch.cuda.synchronize()
        ms, l, _ = sc.profile_read(); sc.profile_enable(False)
        print(f"{tag:40s} {ms/l:8.3f} ms  {flops/(ms/l*1e-3)/1e12:7.1f} TF", flush=True)
    dbgs = [int(x) for x in os.environ.get("SWEEP_DBG", "0,2").split(",")]
    chs = [int(x) for x in os.environ.get("SWEEP_CHUNKS", "1,2,3,4,6,8").split(",")]
    for dbg in dbgs:
        sc.set_option("debug_mode", dbg)
        for ch in chs:
            sc.set_option("chunks", ch)
            run(f"dbg={dbg} chunks={ch}")
main()
