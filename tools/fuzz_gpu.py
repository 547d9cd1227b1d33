"""Dev: randomised differential test of the fused / wide top-K entry points against the C oracle.
python tools/fuzz_gpu.py [cases] [seed].  Every case draws a shape, K, a score structure (continuous, quantised = many
ties, falling or rising with the item index) and an exclusion density, and compares ids + scores bit for bit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recommenders_amd import HipScorer
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
for case in range(cases):
    rng = np.random.default_rng(seed0 * 1000 + case)
    n_users = int(rng.choice([1, 7, 64, 129, 300, 1000, 2500]))
    n_items = int(rng.choice([40, 333, 1500, 3001, 9000, 26744, 60000]))
    k = int(rng.choice([8, 32, 50, 64, 100, 128, 200, 256]))
    K = int(rng.choice([1, 5, 10, 16, 17, 20, 32, 33, 50, 64, 100, 128]))
    bias = bool(rng.integers(2))
    kind = rng.choice(["normal", "ties", "falling", "rising", "popular"])
    U = rng.normal(0, 0.1, (n_users, k)).astype(np.float32)
    V = rng.normal(0, 0.1, (n_items, k)).astype(np.float32)
    if kind == "ties":
        U, V = np.round(U * 8) / 8, np.round(V * 8) / 8
    if kind in ("falling", "rising", "popular"):
        ramp = np.linspace(1.0, 0.0, n_items, dtype=np.float32)
        if kind == "rising":
            ramp = ramp[::-1].copy()
        V[:, 0] = ramp * (3.0 if kind != "popular" else 0.5)
        U[:, 0] = 1.0
    ub = rng.normal(0, 0.01, n_users).astype(np.float32) if bias else None
    ib = rng.normal(0, 0.01, n_items).astype(np.float32) if bias else None
    dens = float(rng.choice([0.0, 0.01, 0.2, 0.9]))
    rows = [np.sort(rng.choice(n_items, int(min(n_items, rng.poisson(dens * n_items))), replace=False)) for _ in range(n_users)]
    if kind == "popular":                                            # exclusions on the best items
        rows = [np.union1d(r, np.arange(min(n_items, int(rng.integers(0, 40))))) for r in rows]
    ptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.int64)
    idx = (np.concatenate(rows) if ptr[-1] else np.zeros(0)).astype(np.int32)
    sc = HipScorer(0)
    sc.set_factors(U, V, ub, ib)
    sc.set_exclusion_csr(ptr, idx)
    tptr = np.zeros(n_users + 1, np.int64)
    sc.set_truth_csr(tptr, np.zeros(0, np.int32))
    users = rng.permutation(n_users).astype(np.int32)
    if rng.integers(2):
        sc.set_option("chunks", int(rng.integers(1, 9)))
    t = time.perf_counter()
    ids, scores = sc.rank_topk(users, K)
    ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
    for again in range(6):                                           # the same launch again: races show as run-to-run differences
        i2, s2 = sc.rank_topk(users, K)
        if not (np.array_equal(i2.cpu().numpy(), ids) and np.array_equal(s2.cpu().numpy(), scores, equal_nan=True)):
            print('   RUN-TO-RUN DIFFERENCE in repeat', again, flush=True)
            bad += 1
    kp = sc.chain_order()
    o_ids, o_sc, _, _ = orc.c_ranking_eval_factor(U, V, ib, ub, users, ptr, idx, tptr, np.zeros(0, np.int32), [K], False, K,
                                                  kperm=kp[kp < k], n_threads=8)
    ok = np.array_equal(ids, o_ids) and np.array_equal(scores, o_sc, equal_nan=True)
    if case % 3 == 0:                                                # the variant that also writes the score rows
        ref_rows = sc.score_dense(users)
        for again in range(3):
            i3, s3, rows = sc.rank_topk_dense(users, K)
            ok &= np.array_equal(i3.cpu().numpy(), o_ids) and np.array_equal(s3.cpu().numpy(), o_sc, equal_nan=True)
            ok &= bool(torch.equal(rows[:, :n_items], ref_rows[:, :n_items]))
    bad += not ok
    print(f"case {case}: users {n_users} items {n_items} k {k} K {K} bias {bias} {kind} excl {dens} overflow {sc.wide_overflows()} "
          f"{'ok' if ok else 'MISMATCH'} {time.perf_counter() - t:.1f}s", flush=True)
    if not ok:
        r = np.flatnonzero((ids != o_ids).any(axis=1) | (scores != o_sc).any(axis=1))[:3]
        for i in r:
            print("   row", i, "user", users[i], "\n    got", ids[i][:12], scores[i][:6], "\n    exp", o_ids[i][:12], o_sc[i][:6])
    sc.close()
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
