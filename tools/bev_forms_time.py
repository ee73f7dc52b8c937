"""The 3x3 forward kernel's in-step forms alone (64 -> 64 @ 4 x 200 x 176 and 128 -> 128 @ 4 x 100 x 88): plain, + statistics epilogue,
+ transform on load, both, and the input-gradient form with the BatchNorm-backward epilogue.  python tools/bev_forms_time.py [iters]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import conv2d as c2  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.manual_seed(0)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for c, h, w in ((64, 200, 176), (128, 100, 88)):
    x = torch.randn(4, c, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(c, c, 3, 3, device=dev) / (3 * c ** 0.5)
    pf, pb = c2.packs(wt)
    bn = torch.nn.BatchNorm2d(c).to(dev).train()
    coef = torch.cat([torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1])
    mean, invstd = torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5
    t0 = time.time()
    while time.time() - t0 < 0.3:
        c2._run(x, pf, c)
        torch.cuda.synchronize()
    res = dict(plain=timed(lambda: c2._run(x, pf, c)),
               stats=timed(lambda: c2._run(x, pf, c, bn)),
               pre=timed(lambda: c2._run(x, pf, c, pre=(coef, True))),
               stats_pre=timed(lambda: c2._run(x, pf, c, bn, pre=(coef, True))),
               dgrad_bn_bwd=timed(lambda: c2._run(x, pb, c, bwd=(x, coef, mean, invstd, bn.weight))))
    print("%d -> %d @ 4 x %d x %d: " % (c, c, h, w) + "  ".join("%s %.1f us" % kv for kv in res.items()), flush=True)
