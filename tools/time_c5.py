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
KS = [int(a) for a in sys.argv[4].split(',')] if len(sys.argv) > 4 else [10]
caps = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
for dbg, cap in [(int(a), c) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"]) for c in caps]:
    sc.set_option("debug_mode", dbg)
    sc.set_option("csr_cand_cap", cap)
    sc.set_option("csr_wave", int(sys.argv[5]) if len(sys.argv) > 5 else 1)
    for K in KS:
        sc.rank_topk(users, K); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): sc.rank_topk(users, K)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        print(f"dbg {dbg} cap {cap} K {K}: {dt*1e3:.2f} ms  {nu/dt/1e6:.2f} M users/s", flush=True)
        import ctypes
        out = (ctypes.c_ulonglong * 16)()
        n = ctypes.c_int64(0)
        sc._lib.ifr_debug_read_stamps(sc._h, out, 16, ctypes.byref(n))
        if n.value == 16 and out[15]:
            names = ["issue next", "consume", "adds land", "barrier 1", "claims+appends", "barrier 2", "read s_nq", "decide+merge", "zeros+out", "prologue"]
            print("   cycles per user (wave 0): " + ", ".join(f"{nm} {out[k] / out[15]:.0f}" for k, nm in enumerate(names)) + f" | total {sum(out[:10]) / out[15]:.0f}", flush=True)
