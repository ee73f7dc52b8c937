"""Fixed cost of the fused BatchNorm forward launches: graphs of 40 forward calls per size, device time per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd.spconv import core as sp  # noqa: E402

dev = torch.device("cuda")
REP = 40
for n, c in ((61858, 16), (70583, 32), (48147, 64), (29989, 64), (35200, 128), (140800, 64), (140800, 128)):
    bn = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev).train()
    x = torch.randn(n, c, device=dev)
    with torch.no_grad():
        for _ in range(3):
            sp.fused_train_bn(bn, x, True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        for _ in range(REP):
            y = sp.fused_train_bn(bn, x, True)
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / (5 * REP)
    print("(%6d, %3d) %5.1f MB: %.1f us per forward (2 launches)" % (n, c, n * c * 4 / 1e6, us))
