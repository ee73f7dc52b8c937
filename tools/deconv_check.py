"""Timing of the own transposed convolutions against MIOpen's on the BEV deblock shapes (4 frames)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from glenet_amd import conv2d as c2

dev = torch.device("cuda", 0)


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


for (b, cin, cout, u, h, w) in ((4, 64, 128, 1, 200, 176), (4, 128, 128, 2, 100, 88)):
    x = torch.randn(b, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = torch.nn.Parameter(torch.randn(cin, cout, u, u, device=dev) / cin ** 0.5)
    gy = torch.randn(b, cout, h * u, w * u, device=dev).contiguous(memory_format=torch.channels_last)
    args = ((u, u), (0, 0), (1, 1), True, (0, 0), 1)
    own_f = t(lambda: c2.deconv(x.detach(), wt.detach()))
    lib_f = t(lambda: F.conv_transpose2d(x.detach(), wt.detach(), None, stride=u))

    def own_b():
        x.grad = None; wt.grad = None
        c2.deconv(x, wt).backward(gy)

    def lib_b():
        x.grad = None; wt.grad = None
        F.conv_transpose2d(x, wt, None, stride=u).backward(gy)
    print((b, cin, cout, u, h, w), "forward us own %.1f lib %.1f | forward + backward own %.1f lib %.1f" % (own_f, lib_f, t(own_b), t(lib_b)), flush=True)
    # the three own launches apart
    import ctypes
    from glenet_amd import _lib
    fwd, bwd = c2._deconv_packs(wt)
    xd = x.detach()
    gx = torch.empty_like(xd)
    y = torch.empty_like(gy)
    n = _lib.query("glx_deconv_wgrad_workspace_bytes", cin, cout, u)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    gw = torch.empty_like(wt)
    s = gw.stride()
    ll = ctypes.c_longlong
    print("   own: forward %.1f  input grad %.1f  weight grad %.1f  pack %.1f us" % (
        t(lambda: _lib.call("glx_deconv_forward", xd, b, h, w, cin, fwd, cout, u, y)),
        t(lambda: _lib.call("glx_deconv_input_grad", gy, b, h, w, cin, bwd, cout, u, gx)),
        t(lambda: _lib.call("glx_deconv_wgrad", xd, gy, b, h, w, cin, cout, u, gw, ll(s[0]), ll(s[1]), ll(s[2]), ll(s[3]), ws, _lib.size_arg(n))),
        t(lambda: c2._deconv_packs(wt))), flush=True)
    print("   lib: input grad %.1f  weight grad %.1f us" % (
        t(lambda: torch.ops.aten.convolution_backward(gy, xd, wt.detach(), None, (u, u), (0, 0), (1, 1), True, (0, 0), 1, [True, False, False])),
        t(lambda: torch.ops.aten.convolution_backward(gy, xd, wt.detach(), None, (u, u), (0, 0), (1, 1), True, (0, 0), 1, [False, True, False]))), flush=True)
