"""RoI-grid pooling + FC stage alone (GPU box), for rocprofv3: the inputs come from one forward pass."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import detector as det, synth  # noqa: E402

K = synth.KITTI
dev = torch.device("cuda", 0)
B = 4
frames = [synth.kitti_frame(i)[0] for i in range(B)]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
torch.manual_seed(0)
flow = det.VoxelRCNNFlow(K).to(dev).eval()
if os.environ.get("ROI_TRAIN"):
    # training path of the stage (tensor ops + group_points autograd), forward + backward
    with torch.no_grad():
        bd = flow(pts, bidx, B)
        rois = bd["rois"][:, :128].contiguous()
    flow.roi_pool.train()
    flow.roi_fc.train()

    def tstage():
        for p in list(flow.roi_pool.parameters()) + list(flow.roi_fc.parameters()):
            p.grad = None
        pooled = flow.roi_pool(rois, bd["multi_scale_3d_features"], bd["multi_scale_3d_strides"], B)
        c, r = flow.roi_fc(pooled)
        (c.square().mean() + r.square().mean()).backward()

    for _ in range(3):
        tstage()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tstage()
    torch.cuda.synchronize()
    print("RoI-grid pool + FC, TRAINING path fwd+bwd: %.2f ms (%d RoIs x 216 grid points, 3 scales)"
          % ((time.perf_counter() - t0) / 10 * 1e3, rois.shape[0] * rois.shape[1]))
    sys.exit(0)
with torch.no_grad():
    bd = flow(pts, bidx, B)
    rois = bd["rois"]

    def stage():
        pooled = flow.roi_pool(rois, bd["multi_scale_3d_features"], bd["multi_scale_3d_strides"], B)
        return flow.roi_fc(pooled)

    for _ in range(5):
        stage()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        stage()
    torch.cuda.synchronize()
    print("RoI-grid pool + FC: %.2f ms (%d RoIs x 216 grid points, 3 scales)" % ((time.perf_counter() - t0) / 20 * 1e3, rois.shape[0] * rois.shape[1]))
