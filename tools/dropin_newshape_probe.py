"""Where a NEW-SHAPE training step in the reference's module layout spends its host time (GPU box): the step is run on batches
with different voxel counts, each step timed on the host with a final synchronise, and one new-shape step under cProfile."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import dropin, glenet_vr as gvr, synth  # noqa: E402

dev = torch.device("cuda", 0)
K = synth.KITTI
seed = torch.tensor([0.2, -0.15, 0.05, 0.1, -0.05, 0.03, 0.08], device=dev)


def batch(first):
    frames = [synth.kitti_frame(first + i) for i in range(4)]
    pts = torch.from_numpy(np.concatenate([f[0] for f in frames])).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f[0]), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    gt = torch.zeros(4, 16, 8, device=dev)
    unc = torch.full((4, 16, 7), 0.05, device=dev)
    for i, f in enumerate(frames):
        gt[i, :len(f[1]), :7] = torch.from_numpy(f[1]).to(dev)
        gt[i, :len(f[1]), 7] = 1
    return pts, bidx, gt, unc


mode = sys.argv[1] if len(sys.argv) > 1 else "base"
if mode == "nocudnn":
    torch.backends.cudnn.enabled = False
if mode == "gemm":
    dropin.pointwise_as_gemm()
with dropin.reference_layout():
    torch.manual_seed(0)
    m = gvr.GLENetVR(K, bev_channels_last=False).to(dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)

    def step(b):
        opt.zero_grad(set_to_none=True)
        loss, _ = m.training_step(b[0], b[1], 4, b[2], b[3], seed_rois_with_gt=seed)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 10.0)
        opt.step()
        m.last = None
    b0 = batch(0)
    for _ in range(3):
        step(b0)
    torch.cuda.synchronize()
    for j in range(1, 7):
        b = batch(4 * j)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(b)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(b)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s: batch %d first time %.1f ms, second time %.1f ms" % (mode, j, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    b = batch(40)
    pr = cProfile.Profile()
    pr.enable()
    step(b)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(14)
