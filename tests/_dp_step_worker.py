"""Worker of tests/test_dist_gpu.py: one rank of a 2-rank data-parallel GLENet-VR training step.
GLX_DIST_BACKEND=gloo (default): both ranks share the box's single GPU, the collective runs over gloo (host copies) --
plumbing only, the arithmetic is the RCCL path's.  GLX_DIST_BACKEND=nccl: one GPU per rank (LOCAL_RANK), the flat gradient
all-reduce over real RCCL / xGMI -- what tests/test_dist_gpu.py runs when two GPUs are visible."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    from glenet_amd import dist as gdist
    from glenet_amd import glenet_vr as gvr
    import test_train_step_gpu as helpers
    rank, local_rank, world = gdist.env_world()
    backend = os.environ.get("GLX_DIST_BACKEND", "gloo")
    di = local_rank % torch.cuda.device_count() if backend == "nccl" else 0
    dev = torch.device("cuda", di)
    torch.cuda.set_device(di)
    gdist.init(backend, device=dev)
    cdev = dev if backend == "nccl" else "cpu"          # where collectives of diagnostics tensors run
    lr = 1e-3
    batches = [helpers._batch(dev, [60, 61], 6000), helpers._batch(dev, [62, 63], 6000)]
    npts = max(b[0].shape[0] for b in batches) + 700

    def build(seed_shift):
        m = helpers._small_model(dev)
        if seed_shift:                      # rank 1 starts from DIFFERENT weights: the broadcast must repair that
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(0.01)
        R, P = m.roi_cfg["NMS_TRAIN"][1], m.roi_cfg["TARGET"]["ROI_PER_IMAGE"]
        gen = torch.Generator(device=dev).manual_seed(4)
        m.fixed_draws = (torch.rand((2, R), device=dev, generator=gen), torch.rand((2, P), device=dev, generator=gen))
        return m

    caps = {}
    probe = gvr.StaticTrainStep(build(0), 2, npts, max_gt=16, lr=lr, seed_rois_with_gt=helpers.JIT)
    for b in batches:
        for k, v in probe.calibrate(b[0], b[1]).items():
            caps[k] = max(caps.get(k, 0), v)
    del probe

    # ---- the data-parallel step: rank r takes batch r
    model = build(rank)
    pipe = gvr.StaticTrainStep(model, 2, npts, max_gt=16, lr=lr, seed_rois_with_gt=helpers.JIT, capacities=caps)
    pipe.data_parallel()                    # broadcast from rank 0 + grad_scale = 1 / world
    pipe.load(*batches[rank])
    buckets = int(os.environ.get("GLX_TEST_GRAD_BUCKETS", "1"))     # 2: the exchange in two buckets, two fwd + bwd graphs
    pipe.capture(split=True, buckets=buckets)
    g_local = None
    if os.environ.get("GLX_DP_DEBUG"):             # the step taken apart: this rank's gradient before the exchange
        from glenet_amd import _lib
        pipe.replay()
        if buckets == 2:
            pipe.graph2.replay()
        torch.cuda.synchronize()
        g_local = pipe.step_optimizer.flat_grad.detach().clone()
        pipe.exchange()
        pipe.update_graph.replay()
        _lib.bump_weights_epoch()
    else:
        pipe.step()
    torch.cuda.synchronize()
    pipe.check()
    opt = pipe.step_optimizer
    p_dp = opt.flat_param.detach().cpu()
    g_dp = (opt.flat_grad.detach() * opt.grad_scale).cpu()          # what the update consumed
    lo, hi = p_dp.clone().to(cdev), p_dp.clone().to(cdev)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    ranks_equal = bool(torch.equal(lo, hi))

    # ---- the same update in ONE process: gradients of batch 0 and batch 1 from the same start, averaged
    ref = gvr.StaticTrainStep(build(0), 2, npts, max_gt=16, lr=lr, seed_rois_with_gt=helpers.JIT, capacities=caps)
    ref.split = True                        # enqueue() = forward + backward + pack, no update
    grads = []
    for b in batches:
        ref.load(*b)
        ref.enqueue()
        torch.cuda.synchronize()
        grads.append(ref.step_optimizer.flat_grad.detach().clone())
    g_ref = (grads[0] + grads[1]) / 2
    ref.step_optimizer.flat_grad.copy_(g_ref)
    ref.step_optimizer.step(packed=True)
    torch.cuda.synchronize()
    p_ref = ref.step_optimizer.flat_param.detach().cpu()
    g_ref = g_ref.cpu()
    scale = float(g_ref.abs().max())
    dp_, dg = (p_dp - p_ref).abs(), (g_dp - g_ref).abs()
    out = dict(rank=rank, ranks_equal=ranks_equal, grad_scale=opt.grad_scale, n=int(p_dp.numel()),
               grad_max_abs=scale, grad_err_max=float(dg.max()), grad_err_over_1e4=int((dg > 1e-4 * scale).sum()),
               param_err_max=float(dp_.max()), param_err_mean=float(dp_.mean()),
               grads_differ_between_batches=float((grads[0] - grads[1]).abs().max()), lr=lr,
               step_count=int(opt.step_count), backend=dist.get_backend(), device=di, world=dist.get_world_size(),
               buckets=buckets, graphs=1 + (pipe.graph2 is not None) + (pipe.update_graph is not None))
    if g_local is not None:
        e_loc = float((g_local - grads[rank]).abs().max())
        print("DPDEBUG rank %d: own recorded gradient vs own eager gradient of the same batch: %.3e (scale %.3e)"
              % (rank, e_loc, float(grads[rank].abs().max())), flush=True)
    if os.environ.get("GLX_DP_DEBUG"):             # which parameters carry the deviation
        off = 0
        worst = []
        for name, prm in model.named_parameters():
            n = prm.numel()
            e = float(dg[off:off + n].max()) if n else 0.0
            worst.append((e, name, n))
            off += n
        for e, name, n in sorted(worst, reverse=True)[:8]:
            print("DPDEBUG rank %d %-60s n=%d err %.3e" % (rank, name, n, e), flush=True)
    print("DPRESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
