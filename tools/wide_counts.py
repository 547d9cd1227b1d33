"""Dev: distribution of the per-user candidate counts of the 16 < K <= 128 path (which users overflow, and where they sit)."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommenders_amd import HipScorer, synth
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
items = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
dev = torch.device("cuda", 0)
dw = synth.make_c3_device(n_users=n, n_items=items, seed=3, device=dev)
sc = HipScorer(0)
sc.set_factors(dw["U"], dw["V"], dw["ubase"], dw["ibase"])
sc.set_exclusion_csr(dw["excl_ptr"], dw["excl_idx"])
users = torch.arange(n, dtype=torch.int32, device=dev)
for rep in range(2):
    sc.rank_topk(users, K)
buf = np.zeros(n, np.uint32)
m = ctypes.c_int64(0)
sc._check(sc._lib.ifr_debug_wide_counts(sc._h, buf.ctypes.data, n, ctypes.byref(m)))
c = buf[: m.value].astype(np.int64)
print("K", K, "n", n, "items", items, "mean", c.mean(), "pcts", np.percentile(c, [1, 10, 50, 90, 99, 100]))
t = c.reshape(-1, 128)
print("per user tile: mean of tile means", t.mean(1).mean(), "min/max tile mean", t.mean(1).min(), t.mean(1).max())
print("tile means (first 40):", np.round(t.mean(1)[:40]).astype(int))
print("within-tile: user-to-user std / mean", (t.std(1) / t.mean(1)).mean())
np.set_printoptions(linewidth=200)
print("all tile means:", np.round(t.mean(1)).astype(int))
