"""The own 3x3 convolution kernels of the BEV backbone alone, on the block layers' shapes, random data (csrc/glx_conv2d.hip):
forward, input gradient, weight gradient.  The program tools/profile_r05_bound.sh puts under rocprofv3 --pmc; without a
profiler it prints the event-timed averages.  argv: iterations per kernel (default 20)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import conv2d as c2  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.manual_seed(0)
cases = []
for cin, cout, h, w in ((64, 64, 200, 176), (128, 128, 100, 88)):
    xi = torch.randn(4, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(4, cout, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    if os.environ.get("GLX_BEV_CONSTANT_DATA"):      # operands without switching activity (is the kernel held by the power limiter?)
        xi.fill_(1.0); gy.fill_(1.0); wt.fill_(0.01)
    pf, pb = c2.packs(wt)
    cases.append((cin, cout, h, w, xi, gy, wt, pf, pb))
# the chip's clock settles under load: half a second of back-to-back launches first
t0 = time.time()
while time.time() - t0 < 0.5:
    for cin, cout, h, w, xi, gy, wt, pf, pb in cases:
        c2._run(xi, pf, cout)
    torch.cuda.synchronize()
for cin, cout, h, w, xi, gy, wt, pf, pb in cases:
    fl = 2.0 * 4 * h * w * 9 * cin * cout
    for name, fn in (("forward", lambda: c2._run(xi, pf, cout)), ("input_grad", lambda: c2._run(gy, pb, cin)),
                     ("weight_grad", lambda: c2.wgrad(xi, gy, wt))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        k = (3 if os.environ.get("GLX_WGRAD_FORM", "2") == "2" else 6) if name == "weight_grad" else (6 if c2.arithmetic() == "bf16x3" else 3)
        print("%d->%d@%dx%d %-12s %7.1f us  %6.1f TFLOP/s fp32-equivalent  %.3f of the 16-bit pipe (%d MFMAs per product, 2.5 PF)"
              % (cin, cout, h, w, name, us, fl / us / 1e6, k * fl / us / 1e6 / 2500.0, k), flush=True)
