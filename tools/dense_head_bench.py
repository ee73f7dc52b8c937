"""MFMA side of the path on the GPU box: BEV backbone + anchor head
forward at the GLENet-VR shape (B,256,200,176), and the CVAE sampler at batch 4096 x 512 points,
fp32 and bf16 autocast, as TFLOP/s and fraction of the dense MFMA peak of the dtype."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import dense_path as dp  # noqa: E402

dev = torch.device("cuda", 0)
PEAK = {"fp32": 157e12, "bf16": 2500e12}      # MI355X_MICROARCH.md dense peaks


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


torch.backends.cudnn.benchmark = True
B = 4
bev = dp.BEVBackbone(256).to(dev).eval()
head = dp.AnchorHead(256, num_class=1, num_anchors_per_location=2).to(dev).eval()
x = torch.randn(B, 256, 200, 176, device=dev)
flops = B * (dp.BEVBackbone.flops_per_frame(200, 176) + 2 * 200 * 176 * 256 * (2 + 14 + 4))
for name, ctx in (("fp32", torch.autocast("cuda", enabled=False)),
                  ("bf16", torch.autocast("cuda", dtype=torch.bfloat16))):
    with torch.no_grad(), ctx:
        dt = timeit(lambda: head(bev({"spatial_features": x})))
    print("BEV backbone + anchor head  %s: %7.3f ms / %d frames = %6.1f TFLOP/s = %.3f of the %s MFMA peak"
          % (name, dt * 1e3, B, flops / dt / 1e12, flops / dt / PEAK[name], name))
    xc = x.to(memory_format=torch.channels_last)
    bevc = bev.to(memory_format=torch.channels_last)
    with torch.no_grad(), ctx:
        dt = timeit(lambda: head(bevc({"spatial_features": xc})))
    print("   channels_last            %s: %7.3f ms = %6.1f TFLOP/s = %.3f" % (name, dt * 1e3, flops / dt / 1e12,
                                                                                 flops / dt / PEAK[name]))
    bev = bev.to(memory_format=torch.contiguous_format)

cv = dp.CVAE(4, 8).to(dev).eval()
pts = torch.randn(4096, 4, 512, device=dev)
# PointFeat flops: x_encoder (4-64-128-512) + decoder's small one, per point
per_obj = 512 * 2 * (4 * 64 + 64 * 128 + 128 * 512 + 4 * 8 + 8 * 8 + 8 * 8)
for name, ctx in (("fp32", torch.autocast("cuda", enabled=False)),
                  ("bf16", torch.autocast("cuda", dtype=torch.bfloat16))):
    with torch.no_grad(), ctx:
        dt = timeit(lambda: cv.sample(pts), n=10)
    print("CVAE sample, 4096 objects x 512 pts %s: %7.3f ms = %6.1f TFLOP/s = %.3f of peak; 30 samples/object = %.1f ms"
          % (name, dt * 1e3, 4096 * per_obj / dt / 1e12, 4096 * per_obj / dt / PEAK[name], 30 * dt * 1e3))
