"""Timing-only ablations of k_conv3x3 (results are wrong by construction): where a step's time goes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glenet_amd import conv2d as c2, _lib

dev = torch.device("cuda", 0)
lib = _lib.load()


def t(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (b, cin, cout, h, w) in ((4, 64, 64, 200, 176), (4, 256, 64, 200, 176), (4, 128, 128, 100, 88)):
    x = torch.randn(b, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / 30
    fwd, bwd = c2.packs(wt)
    out = []
    for grid, abl in ((0, 0), (0, 1), (0, 2), (0, 4), (0, 5), (0, 6), (256, 0), (768, 0)):
        lib.glx_conv3x3_set_grid(grid, abl)
        out.append("g%d/a%d %.1f" % (grid, abl, t(lambda: c2._run(x, fwd, cout))))
    lib.glx_conv3x3_set_grid(0, 0)
    print((b, cin, cout, h, w), " ".join(out), flush=True)
