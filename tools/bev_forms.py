"""The second conv form with its fusions, alone (GPU box): plain / statistics epilogue / transform on load / both / BWD epilogue,
on the two block-layer shapes -- what each fusion costs on top of the plain kernel."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import conv2d as c2  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, cout, h, w in ((64, 64, 200, 176), (128, 128, 100, 88)):
    x = torch.randn(4, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    y_prev = torch.randn(4, cout, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    pf, pb = c2.packs(wt)
    bn = torch.nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).to(dev).train()
    coef = torch.cat([torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1])
    coef_o = torch.cat([torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1])
    mean, invstd, gamma = torch.randn(cout, device=dev) * 0.1, torch.rand(cout, device=dev) + 0.5, torch.rand(cout, device=dev) + 0.5
    t0 = time.time()
    while time.time() - t0 < 0.3:
        c2._run(x, pf, cout)
        torch.cuda.synchronize()
    res = dict(plain=t(lambda: c2._run(x, pf, cout)), stats=t(lambda: c2._run(x, pf, cout, bn=bn)),
               pre=t(lambda: c2._run(x, pf, cout, pre=(coef, True))), stats_pre=t(lambda: c2._run(x, pf, cout, bn=bn, pre=(coef, True))),
               bwd=t(lambda: c2._run(x, pb, cin, bwd=(y_prev, coef_o, mean, invstd, gamma))) if cin == cout else None)
    print("%d->%d@%dx%d: " % (cin, cout, h, w) + "  ".join("%s %.1f us" % (k, v) for k, v in res.items() if v is not None), flush=True)
