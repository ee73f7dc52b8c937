"""Soak of the two kernels that synchronise through per-wave LDS counts instead of barriers (k_pointnet_feat_f16w, k_pointmax_fwd_f16w):
many launches at odd shapes, results compared with the streamed forms every time.  python3 tools/experiments/spinwait_soak.py [rounds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glenet_amd import _lib, dense_path as dp  # noqa: E402

dev = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lib = _lib.load()
torch.manual_seed(0)
m = dp.CVAE(4, 8).to(dev).eval()
fe = m.x_encoder.fe
w1, b1, _, b2, _, b3 = fe._packed()
w2h, e2, w3h, e3 = fe._packed_f16()
narrow, _ = m._sample_pack()
w = torch.randn(512, 128, device=dev)
wh, ew = dp.PointFeat._f16x2_image(w)
shapes = [(4096, 512), (300, 65), (259, 191), (1, 17), (37, 1000), (1024, 64), (5, 129)]
t0 = time.time()
n = 0
for r in range(rounds):
    for B, P in shapes:
        pts = torch.randn(B, 4, P, device=dev)
        h2 = torch.randn(B * P, 128, device=dev)
        out = {}
        for form in (1, 0):
            lib.glx_pointnet_feat_set_form(form)
            f512, f8 = torch.empty(B, 512, device=dev), torch.empty(B, 8, device=dev)
            _lib.call("glx_pointnet_feat_f16x2_pair", pts, B, 4, P, w1, b1, w2h, e2, b2, w3h, e3, b3, f512, narrow, f8)
            v = torch.empty(B, 512, device=dev)
            a = torch.empty(B, 512, device=dev, dtype=torch.int32)
            _lib.call("glx_pointmax_forward_f16x2", h2, B, P, wh, ew, v, a, None)
            out[form] = (f512, f8, v, a)
            n += 2
        lib.glx_pointnet_feat_set_form(1)
        torch.cuda.synchronize()
        assert torch.equal(out[1][2], out[0][2]) and torch.equal(out[1][3], out[0][3]), (r, B, P)
        d = float((out[1][0] - out[0][0]).abs().max())
        assert d <= 1e-3 * float(out[0][0].abs().max()) + 1e-6, (r, B, P, d)
print("%d launches of the register-resident forms in %.1f s, every one checked against the streamed form" % (n // 2, time.time() - t0))
