"""Where an eager training step in the reference's module layout (glenet_amd.dropin.reference_layout) spends its time:
stage marks of GLENetVR.second_stage_losses with events, and the heaviest kernels from torch's profiler.  GPU box only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import dropin, glenet_vr as gvr, synth  # noqa: E402

dev = torch.device("cuda", 0)
K = synth.KITTI
frames = [synth.kitti_frame(i) for i in range(4)]
pts = torch.from_numpy(np.concatenate([f[0] for f in frames])).to(dev)
bidx = torch.from_numpy(np.concatenate([np.full(len(f[0]), i, np.int32) for i, f in enumerate(frames)])).to(dev)
gt = torch.zeros(4, 16, 8, device=dev)
unc = torch.full((4, 16, 7), 0.05, device=dev)
for i, f in enumerate(frames):
    gt[i, :len(f[1]), :7] = torch.from_numpy(f[1]).to(dev)
    gt[i, :len(f[1]), 7] = 1
seed = torch.tensor([0.2, -0.15, 0.05, 0.1, -0.05, 0.03, 0.08], device=dev)
layout = os.environ.get("LAYOUT", "1") == "1"
ctx = dropin.reference_layout() if layout else __import__("contextlib").nullcontext()
with ctx:
    torch.manual_seed(0)
    m = gvr.GLENetVR(K, bev_channels_last=not layout).to(dev).train()
    marks = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e))
    m.mark = mark

    def step():
        m.zero_grad(set_to_none=True)
        marks.clear()
        mark("start")
        loss, _ = m.training_step(pts, bidx, 4, gt, unc, seed_rois_with_gt=seed)
        mark("forward done")
        loss.backward()
        mark("backward done")
        m.last = None
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    step()
    torch.cuda.synchronize()
    t0 = marks[0][1]
    for name, e in marks[1:]:
        print("%9.2f ms  %s" % (t0.elapsed_time(e), name))
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
