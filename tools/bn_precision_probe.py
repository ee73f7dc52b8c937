"""Vendor (MIOpen) against native training-mode BatchNorm on stacked (1, C, M) tensors, both against fp64 (GPU box)."""
import torch
import torch.nn.functional as F

dev = torch.device("cuda", 0)
torch.manual_seed(0)
for c, m, shift in ((32, 5003, 0.0), (32, 5003, 3.0), (64, 110592, 0.0), (16, 777 * 16, 1.0)):
    x = torch.randn(1, c, m, device=dev) * 0.6 + shift
    ref = F.batch_norm(x.double(), None, None, None, None, True, 0.1, 1e-5)
    with torch.backends.cudnn.flags(enabled=True):
        yv = F.batch_norm(x, None, None, None, None, True, 0.1, 1e-5)
    with torch.backends.cudnn.flags(enabled=False):
        yn = F.batch_norm(x, None, None, None, None, True, 0.1, 1e-5)
    print("C %3d M %6d mean %.1f: vendor err %.2e  native err %.2e (max |y| %.2f)"
          % (c, m, shift, float((yv.double() - ref).abs().max()), float((yn.double() - ref).abs().max()), float(ref.abs().max())), flush=True)
