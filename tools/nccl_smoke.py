"""Dev: the process-group calls bench.py makes at N > 1 (RCCL init bound to the device, all-reduce of a device tensor,
barrier, destroy), run with one rank -- a one-GPU box cannot hold two RCCL ranks ("Duplicate GPU")."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
t = torch.tensor([1.5, 2.5, 3.0], dtype=torch.float64, device=dev)
dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize(dev)
print("backend", dist.get_backend(), "world", dist.get_world_size(), "sum", t.tolist())
dist.destroy_process_group()
