"""bench.py -- users ranked/sec (full-catalogue top-10 + NDCG/Recall) on N MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for
N > 1 the driver launches it under torch.distributed.run, one rank per GPU.  Rank 0 prints
ONE JSON line.

Workload at N = 1 (and per rank for N > 1, weak scaling): BASELINE.json configs[1],
"WMF k=128 on MovieLens-20M shape" = 138,493 users x 26,744 items, ~20 M interactions,
synthetic (recommenders_amd/synth.py).  One step = one pass of the hot path over all of the
rank's users: fused score + exclusion + top-10 (ifr_rank_topk), NDCG@10 / Recall@10 /
Precision@10 (ifr_eval_topk_metrics), the all-reduce of the metric sums (N > 1) and the
D2H copy of the top-10 lists.  Model and CSR data are resident in HBM before timing starts.
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
TOPK = 10


def cpu_baseline(w, n_sample, chain):
    """The oracle's C port of the per-user reference loop (oracle/ifr_oracle.c), timed on this
    box's host cores over a bounded user sample of the same workload."""
    from oracle import oracle as orc
    threads = min(16, os.cpu_count() or 1)
    users = np.arange(n_sample, dtype=np.int32)
    kp = chain[chain < w["k"]]
    t = time.perf_counter()
    ids, sc, met, ev = orc.c_ranking_eval_factor(
        w["U"], w["V"], w["ibase"], w["ubase"], users, w["excl_ptr"], w["excl_idx"], w["truth_ptr"], w["truth_idx"],
        [TOPK], False, TOPK, kperm=kp, n_threads=threads)
    dt = time.perf_counter() - t
    return dict(value=n_sample / dt, unit="users/s", cores=threads, kind="port",
                sample=f"{n_sample} users x {w['n_items']} items of the same workload, "
                       f"oracle/ifr_oracle.c orc_ranking_eval_factor, {dt:.1f} s"), ids, met, ev


def cpu_baseline_item(w, n_sample, shift):
    """The same per-user loop for an item-item model (oracle/ifr_oracle.c orc_ranking_eval_item)."""
    from oracle import oracle as orc
    threads = min(16, os.cpu_count() or 1)
    users = np.arange(n_sample, dtype=np.int32)
    M = w["M"]
    if M is not None and not isinstance(M, np.ndarray):
        M = M.cpu().numpy()
    m_csr = None if w["m_ptr"] is None else (w["m_ptr"], w["m_idx"], w["m_val"])
    t = time.perf_counter()
    ids, sc, met, ev = orc.c_ranking_eval_item(M, m_csr, shift, w["n_items"], w["col_scale"], w["h_ptr"], w["h_idx"],
                                               w["h_val"], users, w["excl_ptr"], w["excl_idx"], w["truth_ptr"],
                                               w["truth_idx"], [TOPK], False, TOPK, n_threads=threads)
    dt = time.perf_counter() - t
    return dict(value=n_sample / dt, unit="users/s", cores=threads, kind="port",
                sample=f"{n_sample} users x {w['n_items']} items of the same workload, "
                       f"oracle/ifr_oracle.c orc_ranking_eval_item, {dt:.1f} s"), ids, met, ev


def item_algorithmic_bytes(w):
    """dense M: 4*I bytes per history entry; CSR M: 8 bytes (index + value) per
    nonzero of every history row (the score vector stays in LDS, so its 4*I bytes do not count)."""
    n, I = w["n_users"], w["n_items"]
    if w["M"] is not None:
        return 4.0 * I * float(w["h_ptr"][n])
    row_nnz = np.diff(w["m_ptr"])
    return 8.0 * float(row_nnz[w["h_idx"][: w["h_ptr"][n]]].sum())


def cpu_strong(w, n_sample):
    """Informational: batched sgemm + argpartition on all host threads."""
    t = time.perf_counter()
    done = 0
    for b in range(0, n_sample, 1024):
        us = np.arange(b, min(b + 1024, n_sample))
        S = w["U"][us] @ w["V"].T
        for r, u in enumerate(us):
            S[r, w["excl_idx"][w["excl_ptr"][u]: w["excl_ptr"][u + 1]]] = -np.inf
        part = np.argpartition(-S, TOPK, axis=1)[:, :TOPK]
        np.take_along_axis(S, part, 1).argsort(1)
        done += len(us)
    dt = time.perf_counter() - t
    return dict(value=done / dt, unit="users/s", kind="numpy sgemm+argpartition", sample=f"{done} users, {dt:.1f} s")


PROFILED = {"c2_wmf_ml20m": ("r01_c2_pmc.json", 138493, 26744), "c4_ease_mind": ("r01_c4_pmc.json", 262144, 50000),
            "c5_rp3_ebnerd": ("r01_c5_pmc.json", 1048576, 125000)}


def measured_traffic(config, n_users, n_items):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/*_pmc.json, made by tools/prof.sh); None when the workload differs."""
    if config not in PROFILED:
        return None
    name, nu, ni = PROFILED[config]
    path = os.path.join(ROOT, "profiles", name)
    if (n_users, n_items) != (nu, ni) or not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)["hbm_traffic_per_launch_bytes"]["total_corrected"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2_wmf_ml20m")
    ap.add_argument("--users", type=int, default=None, help="override users per rank (debug only)")
    ap.add_argument("--items", type=int, default=None, help="override catalogue size (debug only)")
    ap.add_argument("--cpu-sample", type=int, default=None, help="users in the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--auc", action="store_true", help="second variant of SURVEY.md 8d: also AUC (dense score rows + ifr_auc_dense)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from recommenders_amd import HipScorer, synth
    if args.cpu_sample is None:
        args.cpu_sample = 65536 if args.config == "c2_wmf_ml20m" else (4096 if args.config == "c3_mf_shard" else 16384)

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

    # each rank owns one user shard of the configured size (weak scaling); item-side parameters replicated
    item_model = args.config in synth.ITEM_CONFIGS
    sc = HipScorer(local)
    if item_model:
        w = synth.make_item(args.config, n_users=args.users, n_items=args.items,
                            seed=synth.ITEM_CONFIGS[args.config][3] + 1000 * rank, device=dev)
        if w["M"] is not None:
            sc.set_item_matrix_dense(w["M"], w["col_scale"])
        else:
            sc.set_item_matrix_csr(w["m_ptr"], w["m_idx"], w["m_val"], w["n_items"], w["col_scale"])
        sc.set_history_csr(w["h_ptr"], w["h_idx"], w["h_val"])
    else:
        w = synth.make(args.config, n_users=args.users, n_items=args.items,
                       seed=synth.CONFIGS[args.config][4] + 1000 * rank)
        sc.set_factors(w["U"], w["V"], w["ubase"], w["ibase"])
    sc.set_exclusion_csr(w["excl_ptr"], w["excl_idx"])
    sc.set_truth_csr(w["truth_ptr"], w["truth_idx"])
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
    auc_fused = bool(args.auc and sc.can_rank_topk_dense(TOPK))
    auc_ld = (w["n_items"] + 31) // 32 * 32
    auc_rows = max(1, min(n, ((1 << 32) if auc_fused else (1 << 28)) // auc_ld)) if args.auc else 0
    auc_buf = torch.empty((auc_rows, auc_ld), dtype=torch.float32, device=dev) if args.auc else None
    auc_out = torch.empty((n,), dtype=torch.float64, device=dev) if args.auc else None

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
        if args.auc and not auc_fused:
            for b in range(0, n, auc_rows):
                ub = users[b:b + auc_rows]
                dense = sc.score_dense(ub, auc_buf[: ub.numel(), : w["n_items"]])
                auc_out[b:b + auc_rows] = sc.auc_dense(ub, dense)
        if world > 1:
            sums = all_reduce(sums)                    # RCCL: per-metric sums + evaluated-user count
        ready[buf].record()
        copy_stream.wait_event(ready[buf])
        with torch.cuda.stream(copy_stream):
            h_ids2[buf].copy_(ids, non_blocking=True)
            h_scores2[buf].copy_(scores, non_blocking=True)
            copied[buf].record()
        return sums

    def all_reduce(t, op=None):
        """In place on the device with RCCL; through host memory in a gloo rehearsal."""
        op = dist.ReduceOp.SUM if op is None else op
        if dist.get_backend() == "nccl":
            dist.all_reduce(t, op=op)
            return t
        c = t.cpu()
        dist.all_reduce(c, op=op)
        return c.to(t.device)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    sc.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sums = step()
    fence()
    dt = time.perf_counter() - t0
    sc.profile_enable(False)
    kern_ms, launches, pairs = sc.profile_read()
    if world > 1:
        tmax = all_reduce(torch.tensor([dt], dtype=torch.float64, device=dev), dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        sums = sums.cpu().numpy()
        avg_ms = kern_ms / max(launches, 1)
        if item_model:
            bytes_per_launch = item_algorithmic_bytes(w)
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            dense = w["M"] is not None
            kind = "EASE-style dense item-item B" if dense else "RP3-beta-style sparse item-item W"
            workload = (f"{args.config}: {kind}, {n} users/GPU x {w['n_items']} items, "
                        f"{float(w['h_ptr'][n]) / n:.1f} history items/user"
                        + ("" if dense else f", {int(np.diff(w['m_ptr']).mean())} neighbours/row")
                        + f", top-{TOPK} + NDCG@{TOPK}/Recall@{TOPK}/Precision@{TOPK}")
            roof = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": achieved / PEAK_HBM_GBS, "traffic": measured_traffic(args.config, n, w["n_items"]),
                    "kernel": "item_item_kernel" if dense else "item_sparse_topk_kernel", "avg_launch_ms": avg_ms, "launches": launches,
                    "bytes_per_launch": bytes_per_launch}
            cfg = {"workload": workload, "users_per_gpu": n, "items": w["n_items"], "topk": TOPK,
                   "parallelism": f"users sharded x{world}, item matrix replicated"}
        else:
            flops_per_launch = 2.0 * n * w["n_items"] * w["k"]       # 2*I*k per user x users per launch
            achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                    "traffic": measured_traffic(args.config, n, w["n_items"]),
                    "kernel": "score_topk_kernel", "avg_launch_ms": avg_ms, "launches": launches,
                    "flops_per_launch": flops_per_launch}
            cfg = {"workload": f"{args.config}: dot-product model, {n} users/GPU x {w['n_items']} items, "
                               f"k={w['k']}, top-{TOPK} + NDCG@{TOPK}/Recall@{TOPK}/Precision@{TOPK}",
                   "users_per_gpu": n, "items": w["n_items"], "k": w["k"], "topk": TOPK,
                   "parallelism": f"users sharded x{world}, item factors replicated"}
        out = {
            "metric": "users ranked/sec (full-catalogue top-10 + NDCG/Recall)",
            "value": world * n * args.steps / dt,
            "unit": "users/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": cfg,
            "roofline": roof,
            "variant": "top-10 + NDCG/Recall/Precision@10" + ((" + AUC (score rows written by the fused kernel)" if auc_fused else " + AUC (dense rows)") if args.auc else ""),
            "eval": {"ndcg@10": sums[0] / sums[-1], "recall@10": sums[1] / sums[-1],
                     "precision@10": sums[2] / sums[-1], "n_evaluated": float(sums[-1])},
        }
        if args.auc:
            out["eval"]["auc"] = float(torch.nanmean(auc_out).item())
        if world == 1 and not args.no_cpu:
            m = min(args.cpu_sample, n)
            if item_model:
                shift = sc.item_fixed_point_shift() if w["M"] is None else 0
                base, o_ids, _, _ = cpu_baseline_item(w, m, shift)
            else:
                base, o_ids, _, _ = cpu_baseline(w, m, sc.chain_order())
                out["cpu_baseline_strong"] = cpu_strong(w, min(2048, n))
            out["cpu_baseline"] = base
            out["parity_on_cpu_sample"] = {"users": m, "top10_ids_equal": bool((h_ids2[(step_no[0] - 1) & 1][:m].numpy() == o_ids).all())}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
