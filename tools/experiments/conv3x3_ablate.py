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
    # first form (weight image in LDS) with its ablations, then the second form by rows per tile
    for grid, abl in ((0, 0x100), (0, 0x101), (0, 0x102), (0, 0x104), (0, 0x105), (0, 0x106), (256, 0x100), (768, 0x100),
                      (0, 0x200), (0, 0x8200), (0, 0x7200), (0, 0x6200)):
        lib.glx_conv3x3_set_grid(grid, abl)
        out.append("g%d/%x %.1f" % (grid, abl, t(lambda: c2._run(x, fwd, cout))))
    lib.glx_conv3x3_set_grid(0, 0x200)
    print((b, cin, cout, h, w), " ".join(out), flush=True)
    # the clock the chip holds inside the kernel: shader-clock ticks / 100 MHz ticks per block, after 0.5 s of back-to-back launches
    import ctypes, time
    stamps = torch.zeros(2 * 512, dtype=torch.int64, device=dev)
    t0 = time.time()
    while time.time() - t0 < 0.5:
        for _ in range(50):
            c2._run(x, fwd, cout)
        torch.cuda.synchronize()
    for abl in (0x100, 0x105, 0x200):
        lib.glx_conv3x3_set_grid(0, abl)
        lib.glx_conv3x3_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
        for _ in range(20):
            c2._run(x, fwd, cout)
        torch.cuda.synchronize()
        lib.glx_conv3x3_set_stamps(None)
        if abl == 0x200:
            continue                       # (the second form carries no stamps)
        st = stamps.view(-1, 2).cpu().double()
        st = st[st[:, 1] > 0]
        ghz = (st[:, 0] / st[:, 1] * 0.1)
        print("   form/ablate %x: in-kernel clock median %.2f GHz (p10 %.2f, p90 %.2f); block lifetime median %.1f us"
              % (abl, ghz.median(), ghz.quantile(0.1), ghz.quantile(0.9), (st[:, 1] / 100).median()), flush=True)
    lib.glx_conv3x3_set_grid(0, 0x200)
