"""Multi-GPU partitioning of the evaluation: users are independent units
(cornac/eval_methods/base_method.py:254-299 reads only row u of every matrix), so ranks take
contiguous user ranges, item-side parameters are replicated, and the only exchange is ONE
all-reduce(SUM) of [sum metric_0 .. sum metric_m, n_evaluated] (RCCL over xGMI on GPUs, gloo
in the CPU tests).  Gathering per-user values is optional and uses all_gather of padded rows."""
import numpy as np


def world_and_rank(distributed="auto"):
    """(world_size, rank) of the default process group, or (1, 0)."""
    if distributed is False:
        return 1, 0
    try:
        import torch.distributed as dist
    except ImportError:
        return 1, 0
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    if distributed is True:
        raise RuntimeError("distributed=True but torch.distributed is not initialised")
    return 1, 0


def shard_range(n, rank, world):
    """Contiguous [lo, hi) of n units for `rank`: sizes differ by at most one."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _comm_device(device):
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        return device if device is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def all_reduce_sums(sums, world, device=None):
    """Sum a small fp64 vector over all ranks (identity when world == 1)."""
    if world <= 1:
        return np.asarray(sums, np.float64)
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.asarray(sums, np.float64), device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def gather_rows(keys, values, world, device=None):
    """Concatenate every rank's (keys int64 [n_r], values fp64 [n_r, c]) in rank order."""
    if world <= 1:
        return keys, values
    import torch
    import torch.distributed as dist
    dev = _comm_device(device)
    n = torch.tensor([len(keys)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m, c = max(counts), values.shape[1]
    buf = torch.zeros((m, c + 1), dtype=torch.float64, device=dev)
    if len(keys):
        buf[: len(keys), 0] = torch.as_tensor(np.asarray(keys, np.float64), device=dev)
        buf[: len(keys), 1:] = torch.as_tensor(np.asarray(values, np.float64), device=dev)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    rows = np.concatenate([o[:cnt].cpu().numpy() for o, cnt in zip(out, counts)], axis=0)
    return rows[:, 0].astype(np.int64), rows[:, 1:]
