"""Device time of detector.topk_desc (glx_topk_desc) against torch.topk at the proposal-layer size; K = 64 shows
the streaming (select + compaction) part alone.  usage: python tools/topk_time.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import detector as det  # noqa: E402

dev = torch.device("cuda")
s = torch.sigmoid(torch.randn(4, 70400, device=dev) * 0.3 - 4.6)
for k in (64, 9000):
    for name, fn in (("kernel", lambda: det.topk_desc(s, k)), ("torch", lambda: torch.topk(s, k, dim=1))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        print("K %5d %-6s %.1f us" % (k, name, (time.perf_counter() - t0) / 50 * 1e6))
