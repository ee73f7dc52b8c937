"""Multi-GPU plumbing of the hot path: one process per GPU, frames shard across ranks, no
data-path collective in a forward pass; timing is fenced by a barrier and reduced with MAX
(what bench.py reports).  torch.distributed backend "nccl" is RCCL on ROCm; "gloo" is used by the
CPU tests."""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend="nccl", device=None):
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def frames_for_rank(rank, world, frames_per_gpu, first_frame=0):
    """Weak scaling: rank r owns frame ids [first + r*F, first + (r+1)*F)."""
    assert 0 <= rank < world
    lo = first_frame + rank * frames_per_gpu
    return list(range(lo, lo + frames_per_gpu))


def fence(device=None):
    """synchronize + barrier + synchronize: nothing of the timed region leaks across it."""
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def _coll_device(device):
    """Collectives run on device tensors under RCCL and on host tensors under gloo."""
    if device is None or dist.get_backend() == "gloo":
        return "cpu"
    return device


def reduce_max(value, device=None):
    """MAX over ranks of a python float (the slowest rank's time is the job's time)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum_int(value, device=None):
    if not (dist.is_available() and dist.is_initialized()):
        return int(value)
    t = torch.tensor([value], dtype=torch.int64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def job_throughput(units_this_rank, seconds_this_rank, device=None):
    """Whole-job units/s = (sum of units over ranks) / (max time over ranks)."""
    return reduce_sum_int(units_this_rank, device) / reduce_max(seconds_this_rank, device)
