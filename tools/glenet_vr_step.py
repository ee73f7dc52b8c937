"""Wall-clock of the composed GLENet-VR training step (glenet_amd.glenet_vr) on one GPU, 4 KITTI-shaped frames:
exact-shape eager step, shape-static eager step, one HIP graph; batches cycled.
usage: python tools/glenet_vr_step.py [--steps 30] [--batches 8] [--no-eager]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import glenet_vr as gvr, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--batches", type=int, default=8)
ap.add_argument("--no-eager", action="store_true")
ap.add_argument("--split", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)
B, K = 4, synth.KITTI
JIT = [0.2, -0.15, 0.05, 0.1, -0.05, 0.03, 0.08]


def batch(ids, max_gt=16):
    frames = [synth.kitti_frame(i) for i in ids]
    pts = torch.from_numpy(np.concatenate([f[0] for f in frames])).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f[0]), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    gt = torch.zeros(len(ids), max_gt, 8, device=dev)
    unc = torch.zeros(len(ids), max_gt, 7, device=dev)
    for i, (fid, f) in enumerate(zip(ids, frames)):
        k = len(f[1])
        gt[i, :k, :7] = torch.from_numpy(f[1]).to(dev)
        gt[i, :k, 7] = 1
        unc[i, :k] = torch.from_numpy(synth.gt_uncertainty(fid, k)).to(dev)
    return pts, bidx, gt, unc


batches = [batch(list(range(B * j, B * j + B))) for j in range(args.batches)]
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
model = gvr.GLENetVR(K).to(dev).train()
seed = torch.tensor(JIT, device=dev)

if not args.no_eager:
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01)

    def eager(j):
        pts, bidx, gt, unc = batches[j % len(batches)]
        opt.zero_grad(set_to_none=True)
        loss, parts = model.training_step(pts, bidx, B, gt, unc, seed_rois_with_gt=seed)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        return loss
    for j in range(3):
        loss = eager(j)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(10):
        loss = eager(j)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("exact-shape eager step: %.2f ms = %.0f frames/s; loss %.4f" % (dt * 1e3, B / dt, float(loss)))
    del loss
    model.last = None

pipe = gvr.StaticTrainStep(model, B, 80000, max_gt=16, lr=1e-3, seed_rois_with_gt=JIT)
pipe.calibrate(batches[0][0], batches[0][1], headroom=1.5)
pipe.load(*batches[0])
for _ in range(2):
    pipe.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for j in range(10):
    pipe.load(*batches[j % len(batches)])
    pipe.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
pipe.check()
print("shape-static eager step: %.2f ms = %.0f frames/s; loss %.4f" % (dt * 1e3, B / dt, float(pipe.loss)))
pipe.capture(split=args.split)
for j in range(3):
    pipe.load(*batches[j % len(batches)])
    pipe.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for j in range(args.steps):
    pipe.load(*batches[j % len(batches)])
    pipe.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
pipe.check()
print("HIP graph%s: %.2f ms/step = %.0f frames/s; loss %.4f; parts %s" % (
    " (split: fwd+bwd | update)" if args.split else "", dt * 1e3, B / dt, float(pipe.loss),
    {k: round(float(v), 4) for k, v in pipe.parts.items()}))

# host cost of handing one step to the device: replay() called on an idle device returns after the enqueue, the
# device finishes later -- if the two are close, the step is bound by the host's graph launch, not by the kernels
enq, tot = [], []
for j in range(10):
    pipe.load(*batches[j % len(batches)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3)
    tot.append((t2 - t0) * 1e3)
print("one replay on an idle device: host enqueue %.2f ms (min %.2f), enqueue -> device idle %.2f ms (min %.2f)"
      % (sum(enq) / len(enq), min(enq), sum(tot) / len(tot), min(tot)))
