"""Numerics + timing of glx_conv3x3 against an fp64 convolution and MIOpen's fp32 kernels on the BEV shapes.
  python tools/conv3x3_check.py [--time]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from glenet_amd import conv2d as c2

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def one(b, cin, cout, h, w, time=False):
    x = torch.randn(b, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5))
    fwd, bwd = c2.packs(wt)
    y = c2._run(x, fwd, cout)
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    lib = F.conv2d(x, wt, None, 1, 1)
    scale = ref.abs().max().item()
    e_own, e_lib = (y.double() - ref).abs().max().item() / scale, (lib.double() - ref).abs().max().item() / scale
    gy = torch.randn(b, cout, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    gx = c2._run(gy, bwd, cin)
    gref = torch.ops.aten.convolution_backward(gy.double(), x.double(), wt.double(), None, (1, 1), (1, 1), (1, 1), False,
                                               (0, 0), 1, [True, False, False])[0]
    glib = torch.ops.aten.convolution_backward(gy, x, wt, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                               [True, False, False])[0]
    gs = gref.abs().max().item()
    g_own, g_lib = (gx.double() - gref).abs().max().item() / gs, (glib.double() - gref).abs().max().item() / gs
    gw = c2.wgrad(x, gy, wt)
    wref = torch.ops.aten.convolution_backward(gy.double(), x.double(), wt.double(), None, (1, 1), (1, 1), (1, 1), False,
                                               (0, 0), 1, [False, True, False])[1]
    wlib = torch.ops.aten.convolution_backward(gy, x, wt, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                               [False, True, False])[1]
    ws_ = wref.abs().max().item()
    w_own, w_lib = (gw.double() - wref).abs().max().item() / ws_, (wlib.double() - wref).abs().max().item() / ws_
    print("   wgrad err own %.2e lib %.2e" % (w_own, w_lib), flush=True)
    line = "(%d,%d->%d,%dx%d) fwd err own %.2e lib %.2e | dgrad err own %.2e lib %.2e" % (b, cin, cout, h, w, e_own, e_lib,
                                                                                      g_own, g_lib)
    if time:
        def t(fn, n=20):
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
        flop = 2.0 * b * h * w * 9 * cin * cout
        t_own, t_lib = t(lambda: c2._run(x, fwd, cout)), t(lambda: F.conv2d(x, wt, None, 1, 1))
        t_gown = t(lambda: c2._run(gy, bwd, cin))
        t_glib = t(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                               [True, False, False]))
        bn = torch.nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).to(dev).train()
        t_bn = t(lambda: c2._run(x, fwd, cout, bn))
        line += " | fwd+stats %.1f us" % t_bn
        t_wown = t(lambda: c2.wgrad(x, gy, wt))
        t_wlib = t(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                               [False, True, False]))
        line += " | wgrad %.1f us lib %.1f us" % (t_wown, t_wlib)
        line += " | fwd %.1f us (%.0f TF) lib %.1f us | dgrad %.1f us lib %.1f us" % (
            t_own, flop / t_own / 1e6, t_lib, t_gown, t_glib)
    print(line, flush=True)
    assert e_own < 4 * max(e_lib, 1e-7) + 1e-6 and g_own < 4 * max(g_lib, 1e-7) + 1e-6 and w_own < 4 * max(w_lib, 1e-7) + 2e-6


if __name__ == "__main__":
    tm = "--time" in sys.argv
    one(1, 64, 64, 8, 16)
    one(2, 64, 128, 19, 37)
    one(1, 128, 64, 100, 88)
    if tm:
        one(4, 64, 64, 200, 176, True)
        one(4, 256, 64, 200, 176, True)
        one(4, 128, 128, 100, 88, True)
