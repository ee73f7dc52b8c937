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


def init(backend="nccl", device=None, force=False):
    """force: initialise the process group for a single process too (bench.py's GLX_BENCH_FORCE_DP plumbing mode: the
    N > 1 code path over real RCCL on one GPU)."""
    rank, local_rank, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29653")
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


def gather_floats(values, device=None):
    """[[values of rank 0], [values of rank 1], ...] on every rank (diagnostics: per-rank step time, exchange time)."""
    vals = [float(v) for v in values]
    if not (dist.is_available() and dist.is_initialized()):
        return [vals]
    t = torch.tensor(vals, dtype=torch.float64, device=_coll_device(device))
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def job_throughput(units_this_rank, seconds_this_rank, device=None):
    """Whole-job units/s = (sum of units over ranks) / (max time over ranks)."""
    return reduce_sum_int(units_this_rank, device) / reduce_max(seconds_this_rank, device)


class GradBucket:
    """Data-parallel gradient exchange of the training step (what DDP does for tools/train.py:
    `torch.nn.parallel.DistributedDataParallel`, pcdet/models/__init__.py + tools/train.py:132-137):
    all parameter gradients packed into ONE flat fp32 buffer and averaged with a single all-reduce.
    The sparse backbone has ~40 small parameter tensors (4.8 MB together); xGMI rings are per-link
    bound and pay a fixed latency per collective, so one message per step instead of forty.

    Usage after backward(): bucket.allreduce_()  (no-op in a single process)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)
        self.views = [v.view_as(p) for v, p in zip(self.flat.split(self.sizes), self.params)]

    def pack(self):
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        if grads:
            torch._foreach_copy_(self.views, grads)
        return self.flat

    def unpack(self):
        """Copy the reduced values back INTO the existing .grad tensors (a replayed training graph
        keeps writing to those addresses; re-pointing .grad at the bucket would orphan them)."""
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
        torch._foreach_copy_([p.grad for p in self.params], self.views)

    def allreduce_(self, average=True):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        self.pack()
        buf = self.flat if dist.get_backend() != "gloo" or not self.flat.is_cuda else self.flat.cpu()
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        if average:
            buf.div_(dist.get_world_size())
        if buf is not self.flat:
            self.flat.copy_(buf)
        self.unpack()
