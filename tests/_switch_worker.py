"""Worker of tests/test_switches_gpu.py: one small recorded GLENet-VR training step under whatever GLX_* environment the parent
set (the switches are read when the package / library loads, so each setting needs its own process).  Prints one JSON line:
loss terms of two replays and the gradient norm."""
import copy
import json
import sys

import numpy as np
import torch


def main():
    from glenet_amd import glenet_vr as gvr
    from glenet_amd import synth
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = False
    cfg = copy.deepcopy(gvr.ROI_HEAD_CFG)
    cfg.update(NMS_TRAIN=(1024, 128, 0.8), DP_RATIO=0.0)
    cfg["TARGET"] = dict(cfg["TARGET"], ROI_PER_IMAGE=32)
    model = gvr.GLENetVR(synth.KITTI, roi_cfg=cfg).to(dev).train()
    ids, n = [80, 81], 6000
    frames = [synth.kitti_frame(i, num_points=n) for i in ids]
    pts = torch.from_numpy(np.concatenate([f[0] for f in frames])).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f[0]), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    gt = torch.zeros(2, 16, 8, device=dev)
    unc = torch.zeros(2, 16, 7, device=dev)
    for i, (fid, f) in enumerate(zip(ids, frames)):
        k = len(f[1])
        gt[i, :k, :7] = torch.from_numpy(f[1]).to(dev)
        gt[i, :k, 7] = 1
        unc[i, :k] = torch.from_numpy(synth.gt_uncertainty(fid, k)).to(dev)
    gen = torch.Generator(device=dev).manual_seed(9)
    model.fixed_draws = (torch.rand((2, 128), device=dev, generator=gen), torch.rand((2, 32), device=dev, generator=gen))
    pipe = gvr.StaticTrainStep(model, 2, pts.shape[0] + 500, max_gt=16, lr=0.0, seed_rois_with_gt=[0.2, -0.15, 0.05, 0.1, -0.05, 0.03, 0.08])
    pipe.calibrate(pts, bidx)
    pipe.load(pts, bidx, gt, unc)
    pipe.capture()
    out = []
    for _ in range(2):
        pipe.step()
        torch.cuda.synchronize()
        out.append({k: float(v) for k, v in pipe.parts.items()})
        out[-1]["loss"] = float(pipe.loss)
    pipe.check()
    print(json.dumps(dict(steps=out, grad_norm=float(pipe.grad_norm), launches="graph")))


if __name__ == "__main__":
    sys.exit(main())
