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


def test_bench_launcher_starts_one_rank_per_gpu(tmp_path):
    """`bench.py --gpus N` outside torchrun: N child processes with the torchrun environment variables
    (the driver's calling convention for N = 1 passes --gpus too; N > 1 must not silently run one rank)."""
    import argparse
    import json
    sys.path.insert(0, ROOT)
    import bench
    script = tmp_path / "rank.py"
    script.write_text("import os, json, sys\n"
                      "d = {k: os.environ[k] for k in ('RANK','LOCAL_RANK','WORLD_SIZE','MASTER_ADDR','MASTER_PORT')}\n"
                      "d['argv'] = sys.argv[1:]\n"
                      "open(os.path.join(%r, 'rank%%s.json' %% d['RANK']), 'w').write(json.dumps(d))\n"
                      "sys.exit(3 if d['RANK'] == '1' and '--fail' in sys.argv else 0)\n" % str(tmp_path))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    rc = bench.spawn_ranks(argparse.Namespace(gpus=3), ["--gpus", "3", "--steps", "5"], script=str(script))
    assert rc == 0
    got = [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in range(3)]
    assert [g["RANK"] for g in got] == ["0", "1", "2"] and all(g["WORLD_SIZE"] == "3" for g in got)
    assert len({g["MASTER_PORT"] for g in got}) == 1 and all(g["MASTER_ADDR"] == "127.0.0.1" for g in got)
    assert got[0]["argv"] == ["--gpus", "3", "--steps", "5"]
    assert bench.spawn_ranks(argparse.Namespace(gpus=2), ["--fail"], script=str(script)) == 3


def _model_grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from glenet_amd import dist as gdist, glenet_vr as gvr, synth
    gdist.init(backend="gloo")
    torch.manual_seed(0)
    model = gvr.GLENetVR(synth.KITTI)                   # parameters only: no kernel runs on the CPU
    params = [p for p in model.parameters() if p.requires_grad]
    g = torch.Generator().manual_seed(100 + rank)
    for i, p in enumerate(params):
        if i % 7 != 3:                                   # some gradients missing, as after a partial backward
            p.grad = torch.randn(p.shape, generator=g)
    local = [None if p.grad is None else p.grad.clone() for p in params]
    bucket = gdist.GradBucket(params)
    addr = [None if p.grad is None else p.grad.data_ptr() for p in params]
    bucket.allreduce_()
    same_addr = all(a is None or a == p.grad.data_ptr() for a, p in zip(addr, params))
    sums = [float(p.grad.double().sum()) for p in params]
    probe = params[10].grad.flatten()[:5].tolist()
    q.put((rank, len(params), bucket.flat.numel(), same_addr, sums, probe,
           [None if x is None else float(x.double().sum()) for x in local]))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_over_the_glenet_vr_parameter_list():
    """The flat all-reduce of the data-parallel step on the real model's parameters (143 parameter tensors, 7.6 M
    values): every rank ends with the mean, written INTO the existing .grad tensors (a replayed update graph
    reads those addresses)."""
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_model_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    (_, n0, flat0, same0, s0, pr0, l0), (_, n1, flat1, same1, s1, pr1, l1) = res
    assert n0 == n1 > 100 and flat0 == flat1 > 7_000_000
    assert same0 and same1
    assert pr0 == pr1
    for a, b, x, y in zip(s0, s1, l0, l1):
        assert a == b
        want = ((x or 0.0) + (y or 0.0)) / 2
        np.testing.assert_allclose(a, want, rtol=1e-4, atol=1e-3)


def _ranges_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import types
    import torch
    import torch.distributed as dist
    from glenet_amd import dist as gdist
    from glenet_amd.optim import FlatAdamW
    gdist.init(backend="gloo")
    gen = torch.Generator().manual_seed(rank + 1)
    # the optimizer's bookkeeping without its device buffers: five parameters in one flat buffer, 16-byte aligned slices
    params = [torch.nn.Parameter(torch.zeros(s)) for s in (6, 10, 3, 8, 5)]
    offsets, n = [], 0
    for p in params:
        offsets.append(n)
        n += (p.numel() + 3) // 4 * 4
    flat = torch.randn(n, generator=gen)
    opt = types.SimpleNamespace(params=params, offsets=offsets, n=n, flat_grad=flat.clone(), grad_scale=1.0)
    early, late, ei, li = FlatAdamW.buckets(opt, params[1:3])             # the "late" run sits in the middle: two early ranges
    whole = flat.clone()
    dist.all_reduce(whole, op=dist.ReduceOp.SUM)
    handles = FlatAdamW.allreduce_ranges_(opt, early, async_op=True)
    FlatAdamW.allreduce_ranges_(opt, [late])
    for h in handles:
        h.wait()
    try:
        FlatAdamW.buckets(opt, [params[0], params[2]])
        contiguous_checked = False
    except ValueError:
        contiguous_checked = True
    q.put((rank, early, late, ei, li, whole.numpy(), opt.flat_grad.numpy().copy(), opt.grad_scale, contiguous_checked))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_in_two_buckets_equals_one_flat_all_reduce():
    """FlatAdamW.buckets / allreduce_ranges_ (the data-parallel step's two-bucket exchange, glenet_vr.StaticTrainStep.capture(
    buckets=2)): the early ranges (asynchronous handles) + the late range cover the flat buffer exactly once and leave what ONE
    all-reduce of the whole buffer leaves, bit for bit; 1 / world is folded into grad_scale; a late set that is not one
    contiguous run is refused."""
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_ranges_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, early, late, ei, li, whole, bucketed, scale, checked in res:
        assert early == [(0, 8), (24, 40)] and late == (8, 24) and ei == [0, 3, 4] and li == [1, 2]
        assert np.array_equal(whole, bucketed) and scale == 0.5 and checked
    assert np.array_equal(res[0][6], res[1][6])
