"""Row sharding of the sampling path over the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL on ROCm, "gloo" on CPU for tests).

The path shards by batch index: a rank owns a contiguous block of rows, creates its own (mu, kappa) rows and draws
its noise from the Philox streams of its GLOBAL row numbers, so samples never cross ranks.  The only exchange is the
sum of a handful of loss scalars (KL sum, reconstruction sum, log-prob sum, count): ONE all-reduce of a packed
float64 buffer per step; dividing by the global count afterwards reproduces the reference's `.mean()`
(model/spherical_vae.py:134,136)."""
import torch
import torch.distributed as dist


def shard_rows(n_rows, rank, world):
    """[begin, end) of the rows rank owns; blocks differ by at most one row and cover [0, n_rows) in rank order"""
    base, rem = divmod(int(n_rows), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def allreduce_scalars(*values, group=None):
    """Sum python floats / 0-dim tensors over the ranks with a single all-reduce of one packed float64 buffer.
    Returns a float64 tensor of len(values) on the device of the first tensor argument (CPU if none)."""
    dev = next((v.device for v in values if torch.is_tensor(v)), torch.device("cpu"))
    buf = torch.stack([v.detach().to(device=dev, dtype=torch.float64).reshape(()) if torch.is_tensor(v)
                       else torch.tensor(float(v), device=dev, dtype=torch.float64) for v in values])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def global_mean(local_sum, local_count, group=None):
    """mean over all ranks' rows of a quantity whose per-rank sum and count are given"""
    s = allreduce_scalars(local_sum, local_count, group=group)
    return s[0] / s[1]
