"""Time of the RoI target layer (glenet_amd.roi_targets.ProposalTargetLayer): 4 frames x 512 proposals x 40
ground-truth rows -> 128 samples per frame (GLENet_VR.yaml:141-153)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import roi_targets  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
B, R, G = 4, 512, 40
gt = np.zeros((B, G, 8), np.float32)
rois = np.zeros((B, R, 7), np.float32)
for b in range(B):
    n = 25
    gt[b, :n, :7] = np.concatenate([rng.uniform([0, -40, -2], [70, 40, 0], (n, 3)), rng.uniform([3, 1.4, 1.3], [4.5, 1.9, 1.8], (n, 3)),
                                    rng.uniform(-3.1, 3.1, (n, 1))], 1)
    gt[b, :n, 7] = 1
    src = rng.integers(0, n, R)
    rois[b] = gt[b, src, :7] + rng.normal(0, 1, (R, 7)) * np.array([0.8, 0.5, 0.1, 0.2, 0.1, 0.1, 0.2]) * (rng.random((R, 1)) < 0.7) \
        + rng.uniform(-10, 10, (R, 7)) * np.array([1, 1, 0, 0, 0, 0, 0]) * (rng.random((R, 1)) < 0.4)
layer = roi_targets.ProposalTargetLayer(dict(ROI_PER_IMAGE=128, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True, CLS_SCORE_TYPE="roi_iou",
                                             CLS_FG_THRESH=0.75, CLS_BG_THRESH=0.25, CLS_BG_THRESH_LO=0.1, HARD_BG_RATIO=0.8,
                                             REG_FG_THRESH=0.55))
bd = {"rois": torch.from_numpy(rois.astype(np.float32)).to(dev), "roi_scores": torch.rand(B, R, device=dev),
      "roi_labels": torch.ones(B, R, dtype=torch.long, device=dev), "gt_boxes": torch.from_numpy(gt).to(dev)}
for _ in range(5):
    td = layer(bd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    td = layer(bd)
torch.cuda.synchronize()
full = (time.perf_counter() - t0) / 50
t0 = time.perf_counter()
for _ in range(50):
    layer.match_and_sample(bd["rois"], bd["roi_labels"], bd["gt_boxes"])
torch.cuda.synchronize()
core = (time.perf_counter() - t0) / 50
print("RoI target layer, %d frames x %d proposals -> 128 samples: %.3f ms (matching + sampling kernels with their "
      "uniform draws %.3f ms); foreground per frame %s" % (B, R, full * 1e3, core * 1e3, td["reg_valid_mask"].sum(1).tolist()))
