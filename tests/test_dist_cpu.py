"""world_size-2 gloo test of the multi-GPU path's host logic (glenet_amd.dist): frames shard
disjointly, the fence is a real barrier, time is reduced with MAX, throughput is whole-job."""
import os
import socket
import sys
import time

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from glenet_amd import dist as gdist
    r, lr, w = gdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    frames = gdist.frames_for_rank(r, w, 4, first_frame=1000)
    gdist.fence()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))             # rank 1 is the slow one
    gdist.fence()
    dt = time.perf_counter() - t0             # both ranks waited for the slow one at the fence
    local = 0.05 * (rank + 1)
    mx = gdist.reduce_max(local)
    thr = gdist.job_throughput(len(frames) * 10, local)
    q.put((rank, frames, dt, mx, thr))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, f0, dt0, mx0, thr0), (r1, f1, dt1, mx1, thr1) = res
    assert f0 == [1000, 1001, 1002, 1003] and f1 == [1004, 1005, 1006, 1007]
    assert dt0 >= 0.095 and dt1 >= 0.095            # the fence held rank 0 until rank 1 arrived
    assert abs(mx0 - 0.10) < 1e-9 and abs(mx1 - 0.10) < 1e-9
    assert abs(thr0 - 80 / 0.10) < 1e-6 and thr0 == thr1


def test_single_process_defaults():
    sys.path.insert(0, ROOT)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    from glenet_amd import dist as gdist
    assert gdist.env_world() == (0, 0, 1)
    assert gdist.frames_for_rank(0, 1, 4) == [0, 1, 2, 3]
    assert gdist.reduce_max(1.5) == 1.5 and gdist.job_throughput(8, 2.0) == 4.0


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from glenet_amd import dist as gdist
    gdist.init(backend="gloo")
    torch.manual_seed(0)                                  # same parameters on every rank
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3, bias=False))
    net[2].weight.requires_grad_(rank >= 0)
    frozen = torch.nn.Parameter(torch.ones(4), requires_grad=False)
    bucket = gdist.GradBucket(list(net.parameters()) + [frozen])
    x = torch.full((2, 5), float(rank + 1))              # rank-dependent data -> different gradients
    net(x).square().sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    bucket.allreduce_()
    q.put((rank, [g.numpy() for g in local], [p.grad.numpy().copy() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_bucket_allreduce():
    """GradBucket: one flat all-reduce leaves every rank with the mean of the ranks' gradients."""
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (_, l0, a0), (_, l1, a1) = res
    assert any(not np.allclose(x, y) for x, y in zip(l0, l1))       # the ranks really differed
    for x, y, u, v in zip(l0, l1, a0, a1):
        np.testing.assert_allclose(u, (x + y) / 2, rtol=1e-6, atol=1e-7)
        assert np.array_equal(u, v)
