"""BEV backbone + anchor head forward alone (fp32, GLENet-VR shape) -- the target of tools/pmc_bev.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import dense_path as dp  # noqa: E402

dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = True
bev = dp.BEVBackbone(256).to(dev).eval()
head = dp.AnchorHead(256, num_class=1, num_anchors_per_location=2).to(dev).eval()
x = torch.randn(4, 256, 200, 176, device=dev)
with torch.no_grad():
    for _ in range(int(os.environ.get("BEV_ITERS", "8"))):
        head(bev({"spatial_features": x}))
torch.cuda.synchronize()
print("bev ok")
