"""Dev: the facade ranking_eval with TWO ranks (gloo) on the GPU box -- each rank builds its own user range through
the device builder (row0 > 0 on rank 1) and an engine over its rows only; every rank must return the reference's means.
Run: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/dist_facade_gpu.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist
from recommenders_amd import ranking_eval
from recommenders_amd import eval as ev
from conftest import load_golden
from test_gpu_facade import _gap_sets, _metrics, _model

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
d = load_golden("gap_all_models.npz")
for force_device in (False, True):
    ev.DEVICE_BUILD_MIN_NNZ = 0 if force_device else 1 << 22
    ev.clear_eval_cache()
    for name in ("bpr", "wmf"):
        train_set, test_set, val_set = _gap_sets(d, name)
        model = _model(d, name, train_set)
        mets = _metrics(d)
        avg, per = ranking_eval(model, mets, train_set, test_set, val_set=val_set, rating_threshold=float(d["rating_threshold"]),
                                distributed=True)
        users = d[f"{name}__users"]
        ref = d[f"{name}__per_user"]
        assert sorted(per[0].keys()) == users.tolist(), "gathered users differ"
        for j, mt in enumerate(mets):
            got = np.asarray([per[j][int(u)] for u in users])
            np.testing.assert_allclose(got, ref[:, j], atol=1e-6, rtol=0, err_msg=mt.name)
        np.testing.assert_allclose(avg, d[f"{name}__avg"], atol=1e-6, rtol=0)
        eng = model._hip_engine_cache[-1] if hasattr(model, "_hip_engine_cache") else None
        src = model._hip_cache_owner.excl_source
        print(f"rank {rank}/{world} {name} device builder {force_device}: ok, {len(users)} users, exclusion rows from row {model._hip_cache_owner.row0}, "
              f"{'device-built' if isinstance(src, ev.DeviceEvalCsrs) else 'host-built'}", flush=True)
dist.barrier()
dist.destroy_process_group()
