"""Device time of the fused training BatchNorm (+ReLU) on BEV-map-sized matrices: forward and backward launches,
effective bandwidth.  usage: python tools/bn_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd.spconv import core as sp  # noqa: E402

dev = torch.device("cuda")
for n, c in ((140800, 64), (35200, 128), (140800, 128), (58000, 16), (24000, 64)):
    bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).train()
    x = torch.randn(n, c, device=dev, requires_grad=True)
    g = torch.randn(n, c, device=dev)
    y = sp.fused_train_bn(bn, x, True)
    y.backward(g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        y = sp.fused_train_bn(bn, x, True)
    torch.cuda.synchronize()
    tf = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50):
        y = sp.fused_train_bn(bn, x, True)
        y.backward(g)
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / 50 - tf
    mb = n * c * 4 / 1e6
    print("(%d, %d) %.1f MB: forward %.1f us (%.2f TB/s for 3 passes), backward %.1f us (%.2f TB/s for 5 passes)"
          % (n, c, mb, tf * 1e6, 3 * mb / tf / 1e6, tb * 1e6, 5 * mb / tb / 1e6))
