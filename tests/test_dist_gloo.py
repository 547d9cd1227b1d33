"""world_size-2 gloo test of the multi-GPU path's host logic: user sharding, the single
all-reduce of metric sums, and the optional gather of per-user values (the compute inside each
rank is the oracle-backed fake engine; on GPUs the same code runs over RCCL)."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_golden
    from recommenders_amd import eval as ifr_eval
    from recommenders_amd import metrics as M
    from test_host_logic import _datasets, _engine
    d = load_golden("toy_all_models.npz")
    train_set, test_set, val_set = _datasets(d)
    full = _engine(d, "bpr", test_set.num_users)
    made = []

    def shard_engine(r0, r1):
        """What model._hip_engine(rows=(r0, r1)) does on a GPU: an engine that holds only those user rows."""
        from fake_engine import OracleEngine
        eng = OracleEngine(full.U[r0:r1], full.V, None if full.ubase is None else full.ubase[r0:r1], full.ibase)
        eng.row0 = r0
        made.append(eng)
        return eng

    fake = types.SimpleNamespace(is_fitted=True)
    mets = [M.NDCG(k=10), M.Recall(k=10), M.AUC()]
    avg, per = ifr_eval.ranking_eval(fake, mets, train_set, test_set, val_set,
                                     rating_threshold=float(d["rating_threshold"]), _engine=shard_engine, distributed=True)
    avg_local, per_local = ifr_eval.ranking_eval(fake, mets, train_set, test_set, val_set,
                                                 rating_threshold=float(d["rating_threshold"]), _engine=shard_engine,
                                                 distributed=True, gather_user_results=False)
    mine = sorted(per_local[0].keys())
    eng = made[-1]
    # the rank's engine saw only its own rows: user factors, exclusion rows and truth rows of [first user, last user]
    span = mine[-1] - mine[0] + 1
    assert eng.row0 == mine[0] and eng.n_users == span
    assert len(eng.ex[0]) - 1 == span and len(eng.tr[0]) - 1 == span
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), avg=np.asarray(avg), users=np.asarray(sorted(per[0].keys())),
             ndcg=np.asarray([per[0][u] for u in sorted(per[0].keys())]), avg_local=np.asarray(avg_local),
             n_local=len(per_local[0]), first=mine[0], last=mine[-1], rows=eng.n_users,
             total_rows=test_set.num_users)
    dist.destroy_process_group()


def test_two_rank_sharded_eval_matches_single_process(tmp_path, toy):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    # single-process result on the same inputs
    from recommenders_amd import eval as ifr_eval
    from recommenders_amd import metrics as M
    from test_host_logic import _datasets, _engine
    train_set, test_set, val_set = _datasets(toy)
    eng = _engine(toy, "bpr", test_set.num_users)
    avg, per = ifr_eval.ranking_eval(types.SimpleNamespace(is_fitted=True), [M.NDCG(k=10), M.Recall(k=10), M.AUC()],
                                     train_set, test_set, val_set, rating_threshold=float(toy["rating_threshold"]),
                                     _engine=eng, distributed=False)
    np.testing.assert_allclose(r0["avg"], avg, rtol=0, atol=1e-12)
    np.testing.assert_allclose(r1["avg"], avg, rtol=0, atol=1e-12)          # every rank holds the reduced mean
    np.testing.assert_allclose(r0["avg_local"], avg, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(r0["users"], sorted(per[0].keys()))       # gathered per-user values are complete
    np.testing.assert_allclose(r0["ndcg"], [per[0][u] for u in sorted(per[0].keys())], atol=0)
    assert int(r0["n_local"]) + int(r1["n_local"]) == len(per[0]) and abs(int(r0["n_local"]) - int(r1["n_local"])) <= 1
    # each rank held only its own user rows: two disjoint ranges, neither the whole user set
    assert int(r0["last"]) < int(r1["first"])
    assert int(r0["rows"]) + int(r1["rows"]) <= int(r0["total_rows"]) and int(r0["rows"]) < int(r0["total_rows"])
