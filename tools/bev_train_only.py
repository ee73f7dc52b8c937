"""BEV backbone + anchor head as the TRAINING step runs them (GLENetVR's own modules: channels-last, fused
BatchNorm, split-stream conv backward), forward + backward -- the target of tools/pmc_bev_r03.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import glenet_vr as gvr, synth  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = gvr.GLENetVR(synth.KITTI, bev_channels_last=True).to(dev).train()
x = torch.randn(4, 256, 200, 176, device=dev).to(memory_format=torch.channels_last).requires_grad_(True)
for _ in range(int(os.environ.get("BEV_ITERS", "6"))):
    bd = model.dense_head(model.backbone_2d({"spatial_features": x}))
    (bd["cls_preds"].sum() + bd["box_preds"].sum() + bd["dir_cls_preds"].sum()).backward()
    x.grad = None
    model.zero_grad(set_to_none=True)
torch.cuda.synchronize()
print("bev train ok")
