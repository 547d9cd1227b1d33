"""bench.py -- users ranked/sec (full-catalogue top-10 + NDCG/Recall) on N MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for
N > 1 the driver launches it under torch.distributed.run, one rank per GPU.  Rank 0 prints
ONE JSON line.

Headline workload at N = 1 (and per rank for N > 1, weak scaling): BASELINE.json configs[1],
"WMF k=128 on MovieLens-20M shape" = 138,493 users x 26,744 items, ~20 M interactions,
synthetic (recommenders_amd/synth.py).  One step = one pass of the hot path over all of the
rank's users: fused score + exclusion + top-10 (ifr_rank_topk), NDCG@10 / Recall@10 /
Precision@10 (ifr_eval_topk_metrics), the all-reduce of the metric sums (N > 1) and the
D2H copy of the top-10 lists.  Model and CSR data are resident in HBM before timing starts.

The same JSON line carries `extra_configs`: the other BASELINE.json configurations, timed in
this process after the headline with their own step counts, each with its own `roofline`,
and at N = 1 `cpu_baseline` + `parity_on_cpu_sample`:
  c3_mf_shard    configs[2] as one GPU sees it: 1,250,000 users (10 M / 8) x 1,000,000 items, k = 128, biases;
                 at N ranks every rank ranks its own 1.25 M-user shard, so N = 8 is the 10 M x 1 M job
  c4_ease_mind   configs[3]: EASE-style dense B 50 K x 50 K                      (N = 1 only)
  c5_rp3_ebnerd  configs[4]: RP3-beta-style CSR W, 125 K items x 256 neighbours   (N = 1 only)
`--config X` makes X the headline instead (no extras); `--no-extra` drops them.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0            # same guide: HBM3E ~8 TB/s

# rocprofv3 --pmc passes of `python bench.py --config X` committed under profiles/ (tools/prof.sh): HBM bytes per launch
# of the dominant kernel.  A constant from that file, not a measurement of the run that prints it: `traffic_source`.
PROFILED = {"c2_wmf_ml20m": ("r01_c2_pmc.json", 138493, 26744), "c4_ease_mind": ("r01_c4_pmc.json", 262144, 50000),
            "c5_rp3_ebnerd": ("r01_c5_pmc.json", 1048576, 125000), "c3_mf_shard": ("r02_c3_pmc.json", 1250000, 1000000)}
for _cfg in list(PROFILED):           # a newer round's profile of the same command wins
    _base = PROFILED[_cfg][0].split("_", 1)[1]
    for _rnd in ("r02", "r03"):
        if os.path.exists(os.path.join(ROOT, "profiles", f"{_rnd}_{_base}")):
            PROFILED[_cfg] = (f"{_rnd}_{_base}",) + PROFILED[_cfg][1:]


def host_threads():
    """Threads for the CPU baseline: the CPU share this process was given (scheduler affinity, capped by the
    OMP_NUM_THREADS the box exports for one GPU's share), not the machine's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    env = os.environ.get("OMP_NUM_THREADS", "")
    if env.isdigit() and int(env) > 0:
        n = min(n, int(env))
    return max(1, n)


def host_description(threads):
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cores": threads, "host_cpus": os.cpu_count(), "cpu_model": model,
            "threads_source": "scheduler affinity capped by OMP_NUM_THREADS (the box's CPU share for one GPU)"}


def cpu_baseline(w, n_sample, chain, topk, want_auc=False):
    """The oracle's C port of the per-user reference loop (oracle/ifr_oracle.c), timed on this
    box's host cores over a bounded user sample of the same workload."""
    from oracle import oracle as orc
    threads = host_threads()
    users = np.arange(n_sample, dtype=np.int32)
    kp = chain[chain < w["k"]]
    t = time.perf_counter()
    ids, sc, met, ev = orc.c_ranking_eval_factor(
        w["U"], w["V"], w["ibase"], w["ubase"], users, w["excl_ptr"], w["excl_idx"], w["truth_ptr"], w["truth_idx"],
        [topk], want_auc, topk, kperm=kp, n_threads=threads)
    dt = time.perf_counter() - t
    out = dict(value=n_sample / dt, unit="users/s", kind="port",
               sample=f"{n_sample} users x {w['n_items']} items of the same workload, "
                      f"oracle/ifr_oracle.c orc_ranking_eval_factor, {dt:.1f} s")
    out.update(host_description(threads))
    return out, ids, met, ev


def cpu_baseline_item(w, n_sample, shift, topk):
    """The same per-user loop for an item-item model (oracle/ifr_oracle.c orc_ranking_eval_item)."""
    from oracle import oracle as orc
    threads = host_threads()
    users = np.arange(n_sample, dtype=np.int32)
    M = w["M"]
    if M is not None and not isinstance(M, np.ndarray):
        M = M.cpu().numpy()
    m_csr = None if w["m_ptr"] is None else (w["m_ptr"], w["m_idx"], w["m_val"])
    t = time.perf_counter()
    ids, sc, met, ev = orc.c_ranking_eval_item(M, m_csr, shift, w["n_items"], w["col_scale"], w["h_ptr"], w["h_idx"],
                                               w["h_val"], users, w["excl_ptr"], w["excl_idx"], w["truth_ptr"],
                                               w["truth_idx"], [topk], False, topk, n_threads=threads)
    dt = time.perf_counter() - t
    out = dict(value=n_sample / dt, unit="users/s", kind="port",
               sample=f"{n_sample} users x {w['n_items']} items of the same workload, "
                      f"oracle/ifr_oracle.c orc_ranking_eval_item, {dt:.1f} s")
    out.update(host_description(threads))
    return out, ids, met, ev


def cpu_baseline_item_f64(w, n_sample, topk):
    """The float64 option of the dense item matrix on the CPU: the oracle's fp64 fma chain per user
    (orc_score_item_dense_f64) and the canonical ranking of the float64 sums."""
    from oracle import oracle as orc
    # only the rows of M the sampled users' histories read (float64 copies of all 50 K rows would be 20 GB of host memory)
    e_end = int(w["h_ptr"][n_sample])
    rows, h_local = np.unique(w["h_idx"][:e_end], return_inverse=True)
    M = w["M"]
    M = (M[rows] if isinstance(M, np.ndarray) else M[rows.astype(np.int64)].cpu().numpy()).astype(np.float64)
    ids = np.full((n_sample, topk), -1, np.int32)
    t = time.perf_counter()
    for u in range(n_sample):
        a, b = w["h_ptr"][u], w["h_ptr"][u + 1]
        s64, _ = orc.c_score_item_dense_f64(M, h_local[a:b], w["h_val"][a:b], w["col_scale"])
        cand = np.setdiff1d(np.arange(w["n_items"]), w["excl_idx"][w["excl_ptr"][u]: w["excl_ptr"][u + 1]])
        head = orc.canonical_rank(cand, s64[cand], topk)
        ids[u, : len(head)] = head
    dt = time.perf_counter() - t
    out = dict(value=n_sample / dt, unit="users/s", kind="port",
               sample=f"{n_sample} users x {w['n_items']} items of the same workload, float64 matrix, "
                      f"oracle orc_score_item_dense_f64 + canonical ranking of the float64 sums, 1 thread, {dt:.1f} s")
    out.update(host_description(1))
    return out, ids


def item_algorithmic_bytes(w):
    """dense M: 4*I bytes per history entry; CSR M: 8 bytes (index + value) per
    nonzero of every history row (the score vector stays in LDS, so its 4*I bytes do not count)."""
    n, I = w["n_users"], w["n_items"]
    if w["M"] is not None:
        return 4.0 * I * float(w["h_ptr"][n])
    row_nnz = np.diff(w["m_ptr"])
    return 8.0 * float(row_nnz[w["h_idx"][: w["h_ptr"][n]]].sum())


def cpu_strong(w, n_sample, topk=10):
    """Informational: batched sgemm + argpartition on all host threads."""
    t = time.perf_counter()
    done = 0
    for b in range(0, n_sample, 1024):
        us = np.arange(b, min(b + 1024, n_sample))
        S = w["U"][us] @ w["V"].T
        for r, u in enumerate(us):
            S[r, w["excl_idx"][w["excl_ptr"][u]: w["excl_ptr"][u + 1]]] = -np.inf
        part = np.argpartition(-S, topk, axis=1)[:, :topk]
        np.take_along_axis(S, part, 1).argsort(1)
        done += len(us)
    dt = time.perf_counter() - t
    return dict(value=done / dt, unit="users/s", kind="numpy sgemm+argpartition", sample=f"{done} users, {dt:.1f} s")


def measured_traffic(config, n_users, n_items):
    """(HBM bytes per launch of the dominant kernel, source file) from the committed rocprofv3 PMC passes of this same
    command; (None, None) when the workload differs from the profiled one."""
    if config not in PROFILED:
        return None, None
    name, nu, ni = PROFILED[config]
    path = os.path.join(ROOT, "profiles", name)
    if (n_users, n_items) != (nu, ni) or not os.path.exists(path):
        return None, None
    with open(path) as f:
        return json.load(f)["hbm_traffic_per_launch_bytes"]["total_corrected"], "profiles/" + name


class Dist:
    """The N > 1 plumbing: RCCL ("nccl") all-reduce on the device, or through host memory in a gloo rehearsal."""

    def __init__(self, world, dev):
        self.world, self.dev = world, dev

    def all_reduce(self, t, op=None):
        import torch.distributed as dist
        op = dist.ReduceOp.SUM if op is None else op
        if dist.get_backend() == "nccl":
            dist.all_reduce(t, op=op)
            return t
        c = t.cpu()
        dist.all_reduce(c, op=op)
        return c.to(t.device)

    def fence(self):
        import torch
        torch.cuda.synchronize(self.dev)
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(self.dev)


_HOST_WORKLOADS = {}      # (config, k, rank): NumPy workloads are reused by the variants of one configuration


def build_workload(config, args, rank, dev, var):
    """(engine with model + CSRs resident, workload dict with the HOST arrays the CPU baseline samples)."""
    from recommenders_amd import HipScorer, synth
    sc = HipScorer(dev.index)
    for opt in (args.dev or "").split(","):                   # --dev name=value,...: development knobs (tools/), never the default run
        if opt:
            name, value = opt.split("=")
            sc.set_option(name, int(value))
    if config in synth.ITEM_CONFIGS:
        key = (config, None, rank)
        if key not in _HOST_WORKLOADS:
            _HOST_WORKLOADS.clear()
            _HOST_WORKLOADS[key] = synth.make_item(config, n_users=args.users, n_items=args.items,
                                                   seed=synth.ITEM_CONFIGS[config][3] + 1000 * rank, device=dev)
        w = _HOST_WORKLOADS[key]
        if w["M"] is not None:
            # item_f64 (variant, not the headline arithmetic): the float64 option of the dense item matrix
            sc.set_item_matrix_dense(w["M"], w["col_scale"], precision="f64" if var["item_f64"] else "f32")
        else:
            sc.set_item_matrix_csr(w["m_ptr"], w["m_idx"], w["m_val"], w["n_items"], w["col_scale"])
        sc.set_history_csr(w["h_ptr"], w["h_idx"], w["h_val"])
        sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
        sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
    elif config == "c3_mf_shard" and not args.host_synth:
        # 1.25 M users x 60 interactions: generated on the device (seconds instead of minutes of NumPy); the CPU
        # baseline gets host copies of V and of the sampled users' rows only
        dw = synth.make_c3_device(n_users=args.users, n_items=args.items, seed=synth.CONFIGS[config][4] + 1000 * rank,
                                  device=dev)
        sc.set_factors(dw["U"], dw["V"], dw["ubase"], dw["ibase"])
        sc.set_exclusion_csr(dw["excl_ptr"], dw["excl_idx"])
        sc.set_truth_csr(dw["truth_ptr"], dw["truth_idx"])
        m = min(var["cpu_sample"], dw["n_users"])
        w = synth.host_sample(dw, m)
        del dw
    else:
        key = (config, var["k"], rank)
        if key not in _HOST_WORKLOADS:
            _HOST_WORKLOADS.clear()
            _HOST_WORKLOADS[key] = synth.make(config, n_users=args.users, n_items=args.items,
                                              seed=synth.CONFIGS[config][4] + 1000 * rank, k=var["k"])
        w = _HOST_WORKLOADS[key]
        sc.set_factors(w["U"], w["V"], w["ubase"], w["ibase"])
        sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
        sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
    return sc, w


def variant(args, config, **over):
    """What one timed entry runs: list length, AUC, the float64 item matrix, a factor dimension other than the
    configuration's, the size of the CPU sample."""
    v = dict(topk=args.topk, auc=False, item_f64=False, k=None,
             cpu_sample=args.cpu_sample if args.cpu_sample is not None else
             {"c2_wmf_ml20m": 65536, "c3_mf_shard": 2048}.get(config, 16384))
    v.update(over)
    return v


def run_config(config, args, steps, warmup, world, rank, dev, D, with_cpu, var):
    """Time `steps` passes of the hot path over this rank's users of `config` (variant `var`); rank 0 returns the
    result fields."""
    import torch
    from recommenders_amd import synth
    TOPK, auc, item_f64 = var["topk"], var["auc"], var["item_f64"]
    sc, w = build_workload(config, args, rank, dev, var)
    item_model = w.get("M", None) is not None or w.get("m_ptr", None) is not None
    n = w["n_users"]
    users = torch.arange(n, dtype=torch.int32, device=dev)
    # two sets of output buffers: the D2H copy of step i's top-10 lists runs on its own stream while step i + 1
    # computes (the last step's copy is waited for by the fence that ends the timed region)
    ids2 = [torch.empty((n, TOPK), dtype=torch.int32, device=dev) for _ in range(2)]
    scores2 = [torch.empty((n, TOPK), dtype=torch.float32, device=dev) for _ in range(2)]
    h_ids2 = [torch.empty((n, TOPK), dtype=torch.int32).pin_memory() for _ in range(2)]
    h_scores2 = [torch.empty((n, TOPK), dtype=torch.float32).pin_memory() for _ in range(2)]
    copy_stream = torch.cuda.Stream(dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    step_no = [0]

    # AUC variant: dot-product models let the fused kernel write the score rows the AUC kernel reads
    # (ifr_rank_topk_dense, up to a 16 GB buffer per batch of users); item-item models score the rows separately
    auc_fused = bool(auc and sc.can_rank_topk_dense(TOPK))
    auc_ld = (w["n_items"] + 31) // 32 * 32
    auc_rows = max(1, min(n, ((1 << 32) if auc_fused else (1 << 28)) // auc_ld)) if auc else 0
    auc_buf = torch.empty((auc_rows, auc_ld), dtype=torch.float32, device=dev) if auc else None
    auc_out = torch.empty((n,), dtype=torch.float64, device=dev) if auc else None

    def step():
        buf = step_no[0] & 1
        step_no[0] += 1
        ids, scores = ids2[buf], scores2[buf]
        torch.cuda.current_stream(dev).wait_event(copied[buf])      # the copy that last read these buffers is done
        if auc_fused:
            for b in range(0, n, auc_rows):
                ub = users[b:b + auc_rows]
                _, _, dense = sc.rank_topk_dense(ub, TOPK, auc_buf, ids[b:b + auc_rows], scores[b:b + auc_rows])
                auc_out[b:b + auc_rows] = sc.auc_dense(ub, dense)
        else:
            sc.rank_topk(users, TOPK, ids, scores)
        per_user, evaluated, sums = sc.eval_topk_metrics(users, ids, [TOPK])
        if auc and not auc_fused:
            for b in range(0, n, auc_rows):
                ub = users[b:b + auc_rows]
                dense = sc.score_dense(ub, auc_buf[: ub.numel(), : w["n_items"]])
                auc_out[b:b + auc_rows] = sc.auc_dense(ub, dense)
        if world > 1:
            sums = D.all_reduce(sums)                  # RCCL: per-metric sums + evaluated-user count
        ready[buf].record()
        copy_stream.wait_event(ready[buf])
        with torch.cuda.stream(copy_stream):
            h_ids2[buf].copy_(ids, non_blocking=True)
            h_scores2[buf].copy_(scores, non_blocking=True)
            copied[buf].record()
        return sums

    for _ in range(warmup):
        step()
    D.fence()
    sc.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        sums = step()
    D.fence()
    dt = time.perf_counter() - t0
    sc.profile_enable(False)
    kern_ms, launches, pairs = sc.profile_read()
    if world > 1:
        import torch.distributed as dist
        tmax = D.all_reduce(torch.tensor([dt], dtype=torch.float64, device=dev), dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank != 0:
        sc.close()
        return None

    sums = sums.cpu().numpy()
    avg_ms = kern_ms / max(launches, 1)
    traffic, traffic_source = measured_traffic(config, n, w["n_items"])
    if (item_f64 and config in synth.ITEM_CONFIGS) or auc or TOPK != 10 or var["k"] is not None:
        traffic, traffic_source = None, None                      # the committed counter pass is the fp32 matrix
    if item_model:
        bytes_per_launch = item_algorithmic_bytes(w) * (2.0 if (item_f64 and w["M"] is not None) else 1.0)   # fp64 rows: 8 B per entry
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        dense = w["M"] is not None
        kind = "EASE-style dense item-item B" if dense else "RP3-beta-style sparse item-item W"
        workload = (f"{config}: {kind}, {n} users/GPU x {w['n_items']} items, "
                    f"{float(w['h_ptr'][n]) / n:.1f} history items/user"
                    + ("" if dense else f", {int(np.diff(w['m_ptr']).mean())} neighbours/row")
                    + f", top-{TOPK} + NDCG@{TOPK}/Recall@{TOPK}/Precision@{TOPK}")
        roof = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": achieved / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": "item_item_kernel" if dense else "item_sparse_wave_kernel + item_sparse_topk_kernel (the users it hands over)", "avg_launch_ms": avg_ms, "launches": launches,
                "bytes_per_launch": bytes_per_launch}
        if dense:
            roof["note"] = ("algorithmic rate (4 or 8 bytes per entry of every history row): rows of B that concurrently scored users "
                            "share are served from L2 / Infinity Cache, so it can exceed what HBM delivers (about 6.3 TB/s measured); "
                            "`traffic` is the HBM-side byte count of the committed counter pass")
        if not dense:
            # the HBM roof is the contract's statement for this byte-moving path; what limits the kernel is instruction
            # issue (constants of the committed counter run, like `traffic`)
            roof["limiter"] = "instruction issue of the wave-per-user kernel (VALU about 75 % of SIMD cycles), not the memory path: see profiles/r03_c5_sq_counters.json"
        cfg = {"workload": workload, "users_per_gpu": n, "items": w["n_items"], "topk": TOPK,
               "parallelism": f"users sharded x{world}, item matrix replicated"}
    else:
        flops_per_launch = 2.0 * n * w["n_items"] * w["k"]       # 2*I*k per user x users per launch
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": "score_topk_kernel", "avg_launch_ms": avg_ms, "launches": launches,
                "flops_per_launch": flops_per_launch}
        cfg = {"workload": f"{config}: dot-product model, {n} users/GPU x {w['n_items']} items, "
                           f"k={w['k']}, top-{TOPK} + NDCG@{TOPK}/Recall@{TOPK}/Precision@{TOPK}",
               "users_per_gpu": n, "items": w["n_items"], "k": w["k"], "topk": TOPK,
               "parallelism": f"users sharded x{world}, item factors replicated"}
    out = {
        "value": world * n * steps / dt,
        "unit": "users/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": 1e3 * dt / steps,
        "scaling": "weak",
        "dtype": "f64" if (item_f64 and config in synth.ITEM_CONFIGS and w.get("M") is not None) else "f32",
        "data": "synthetic",
        "config": cfg,
        "roofline": roof,
        "variant": f"top-{TOPK} + NDCG/Recall/Precision@{TOPK}" + ((" + AUC (score rows written by the fused kernel)" if auc_fused else " + AUC (dense rows)") if auc else ""),
        "eval": {f"ndcg@{TOPK}": sums[0] / sums[-1], f"recall@{TOPK}": sums[1] / sums[-1],
                 f"precision@{TOPK}": sums[2] / sums[-1], "n_evaluated": float(sums[-1])},
    }
    if auc:
        out["eval"]["auc"] = float(torch.nanmean(auc_out).item())
    if not item_model and TOPK > 16:
        sc.set_option("wide_stats", 1)             # one more, untimed pass that counts the candidates
        step()
        torch.cuda.synchronize(dev)
        wu, wc, wk = sc.wide_stats()
        out["wide_path"] = {"overflow_users": sc.wide_overflows(), "users": wu, "candidates_per_user": wc / max(wu, 1),
                            "kept_per_user": wk / max(wu, 1)}
    if with_cpu:
        m = min(var["cpu_sample"], n, w.get("host_rows", n))
        o_met = None
        if item_model:
            shift = sc.item_fixed_point_shift() if w["M"] is None else 0
            if item_f64 and w["M"] is not None:
                base, o_ids = cpu_baseline_item_f64(w, m, TOPK)
            else:
                base, o_ids, _, _ = cpu_baseline_item(w, m, shift, TOPK)
        else:
            base, o_ids, o_met, o_ev = cpu_baseline(w, m, sc.chain_order(), TOPK, want_auc=auc)
            if config == "c2_wmf_ml20m" and var["k"] is None and not auc and TOPK == 10:
                out["cpu_baseline_strong"] = cpu_strong(w, min(2048, n))
        out["cpu_baseline"] = base
        out["parity_on_cpu_sample"] = {"users": m, "top10_ids_equal" if TOPK == 10 else f"top{TOPK}_ids_equal": bool((h_ids2[(step_no[0] - 1) & 1][:m].numpy() == o_ids).all())}
        if auc and o_met is not None:
            ev = o_ev.astype(bool)
            out["parity_on_cpu_sample"]["auc_equal"] = bool((auc_out[:m].cpu().numpy()[ev] == o_met[ev][:, -1]).all())
    sc.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, help="headline configuration (default c2_wmf_ml20m with the others as extra_configs)")
    ap.add_argument("--users", type=int, default=None, help="override users per rank (debug only)")
    ap.add_argument("--items", type=int, default=None, help="override catalogue size (debug only)")
    ap.add_argument("--cpu-sample", type=int, default=None, help="users in the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline only")
    ap.add_argument("--host-synth", action="store_true", help="c3_mf_shard: NumPy generator (Zipf items) instead of the device one")
    ap.add_argument("--topk", type=int, default=10, help="list length K (default 10, the metric's; 17..128 takes the candidate-buffer variant of the fused kernel)")
    ap.add_argument("--item-f64", action="store_true", help="c4_ease_mind variant: float64 item matrix and sums (twice the bytes; use --no-cpu)")
    ap.add_argument("--dev", default="", help="development only: comma-separated library options name=value (debug_mode, chunks, "
                    "target_blocks, pair_max, wide_chunk_cap, ...) set on every engine; the library itself reads no environment variable")
    ap.add_argument("--k", type=int, default=None, help="factor dimension other than the configuration's (c2 shape with k = 200: the reference's WMF default)")
    ap.add_argument("--auc", action="store_true", help="second variant of SURVEY.md 8d: also AUC (dense score rows + ifr_auc_dense)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU; a rehearsal with more ranks than GPUs (IFR_BENCH_BACKEND=gloo on a one-GPU box) shares devices
    local = local % max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        backend = os.environ.get("IFR_BENCH_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    D = Dist(world, dev)

    headline = args.config or "c2_wmf_ml20m"
    with_cpu = world == 1 and not args.no_cpu
    res = run_config(headline, args, args.steps, args.warmup, world, rank, dev, D, with_cpu,
                     variant(args, headline, auc=args.auc, item_f64=args.item_f64, k=args.k))
    extras = []
    if args.config is None and not args.no_extra and args.users is None and args.items is None and args.topk == 10 \
            and not args.auc and args.k is None:
        # (name, config, steps, warm-up, variant): bounded so that the whole default run stays within a few minutes.
        # The C2 variants come first (they reuse the headline's host workload): every number DESIGN.md quotes for a
        # variant of the path is measured by the driver's own run, with its own roofline and CPU parity sample.
        plan = []
        if world == 1:
            plan += [("c2_auc", "c2_wmf_ml20m", 10, 3, dict(auc=True, cpu_sample=4096)),
                     ("c2_top100", "c2_wmf_ml20m", 10, 3, dict(topk=100, cpu_sample=4096)),
                     ("c2_k200", "c2_wmf_ml20m", 10, 3, dict(k=200, cpu_sample=4096))]
        plan += [("c3_mf_shard", "c3_mf_shard", 2, 1, {})]
        if world == 1:
            plan += [("c4_ease_mind", "c4_ease_mind", 3, 1, {}),
                     ("c4_ease_mind_f64", "c4_ease_mind", 2, 1, dict(item_f64=True, cpu_sample=512)),
                     ("c5_rp3_ebnerd", "c5_rp3_ebnerd", 6, 2, {})]
        for name, cfg_name, k_steps, k_warm, over in plan:
            torch.cuda.empty_cache()
            r = run_config(cfg_name, args, k_steps, k_warm, world, rank, dev, D, with_cpu, variant(args, cfg_name, **over))
            if r is not None:
                r["name"] = name
                extras.append(r)
    if rank == 0:
        out = {"metric": "users ranked/sec (full-catalogue top-10 + NDCG/Recall)"}
        out.update(res)
        out["higher_is_better"] = True
        out["vs_baseline"] = None
        if extras:
            out["extra_configs"] = extras
        order = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline"]
        out = {**{k: out[k] for k in order}, **{k: v for k, v in out.items() if k not in order}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
