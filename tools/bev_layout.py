"""BEV backbone + anchor head, training mode, forward + backward on 4 KITTI-sized maps: NCHW vs channels_last
(MIOpen picks different solvers per layout), with MIOpen immediate mode.  usage: python tools/bev_layout.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import dense_path as dp  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
x0 = torch.randn(4, 256, 200, 176, device=dev) * (torch.rand(4, 1, 200, 176, device=dev) < 0.1)


def run(cl, bench):
    torch.backends.cudnn.benchmark = bench
    bev, head = dp.BEVBackbone(256).to(dev).train(), dp.AnchorHead(256, 1, 2).to(dev).train()
    x = x0.clone()
    if cl:
        bev, head = bev.to(memory_format=torch.channels_last), head.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)

    def step():
        d = head(bev({"spatial_features": x}))
        (d["cls_preds"].square().mean() + d["box_preds"].square().mean() + d["dir_cls_preds"].square().mean()).backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3


for cl in (False, True):
    print("channels_last=%s find=off: %.2f ms fwd+bwd" % (cl, run(cl, False)), flush=True)
if "--find" in sys.argv:
    for cl in (False, True):
        print("channels_last=%s find=on: %.2f ms fwd+bwd" % (cl, run(cl, True)), flush=True)
