"""Dev: time the CSR top-K kernel alone on the C5 workload (debug_mode ablations change results)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
nu = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
w = synth.make_item("c5_rp3_ebnerd", n_users=nu)
sc = HipScorer(0)
sc.set_item_matrix_csr(w["m_ptr"], w["m_idx"], w["m_val"], w["n_items"], w["col_scale"])
sc.set_history_csr(w["h_ptr"], w["h_idx"], w["h_val"])
sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
users = np.arange(nu, dtype=np.int32)
caps = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
for dbg, cap in [(int(a), c) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"]) for c in caps]:
    sc.set_option("debug_mode", dbg)
    sc.set_option("csr_pass_entries", cap)
    sc.rank_topk(users, 10); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): sc.rank_topk(users, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"dbg {dbg} cap {cap}: {dt*1e3:.2f} ms  {nu/dt/1e6:.2f} M users/s", flush=True)
    import ctypes
    out = (ctypes.c_ulonglong * 16)()
    n = ctypes.c_int64(0)
    sc._lib.ifr_debug_read_stamps(sc._h, out, 16, ctypes.byref(n))
    if n.value == 16 and out[9]:
        names = ["prologue", "plan", "bounds+excl+stage", "sweep", "scan+bound", "select", "merge", "zeros+out"]
        nus, passes = out[9], out[8]
        tot = sum(out[k] for k in range(8))
        print("   raw", list(out))
        print(f"   phases (cycles per user of thread 0; {passes / nus:.2f} passes/user; total {tot / nus:.0f}): " +
              ", ".join(f"{nm} {out[k] / nus:.0f}" for k, nm in enumerate(names)) +
              f" | inside sweep (thread 0's wave): loads {out[10] / nus:.0f}, fixed+hash {out[11] / nus:.0f}, atomics {out[12] / nus:.0f}", flush=True)
