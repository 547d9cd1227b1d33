"""Dev tool: per-phase cycle split of the fused kernel's tile loop (debug bit 7, s_memtime stamps).
Needs a library built with the stamps compiled in: make -C recommenders_amd/csrc clean all STAMPS=1."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer, synth
w = synth.make("c2_wmf_ml20m")
sc = HipScorer(0)
sc.set_factors(w["U"], w["V"], w["ubase"], w["ibase"]); sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
users = torch.arange(w["n_users"], dtype=torch.int32, device="cuda")
sc._lib.ifr_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
for base in [int(x) for x in sys.argv[1:]] or [0, 2, 3, 64, 66, 67]:
    sc.set_option("chunks", int(os.environ.get("IFR_CHUNKS", "4"))); sc.set_option("debug_mode", base | 128)
    sc.rank_topk(users, 10)
    torch.cuda.synchronize(); sc.profile_enable(True); sc.rank_topk(users, 10); torch.cuda.synchronize()
    ms, _, _ = sc.profile_read(); sc.profile_enable(False)
    buf = np.zeros(4 * 8 * 20000, np.uint64); n = ctypes.c_int64(0)
    sc._check(sc._lib.ifr_debug_read_stamps(sc._h, buf.ctypes.data, len(buf), ctypes.byref(n)))
    a = buf[: n.value].reshape(-1, 8).astype(np.float64)
    a = a[a[:, 3] > 100]
    per = a[:, :3] / a[:, 3:4]
    print(f"   hit tiles {a[:,6].sum() / a[:,3].sum():.2f} of tiles; per hit tile: scan+append {a[:,4].sum() / max(a[:,6].sum(), 1):.0f}, flush+retry {a[:,5].sum() / max(a[:,6].sum(), 1):.0f}")
    print(f"dbg={base:3d}: per tile cycles (memtime ticks): top->firstMFMA {per[:,0].mean():7.0f}  MFMA zone {per[:,1].mean():7.0f}  "
          f"zoneB+store+barrier {per[:,2].mean():7.0f}  total {per.sum(1).mean():7.0f}   waves {len(a)}  kernel {ms:.3f} ms  wave-busy ticks/us {a[:, :3].sum() / len(a) / (ms * 1e3) * (len(a) / (4 * (512 if base < 64 else 256))):.0f}", flush=True)
