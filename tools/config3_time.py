"""configs[3] alone: python tools/config3_time.py -> the bench's config3 object as one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    if os.environ.get("GLX_POINTNET_FORM"):                   # 0: W3 through the LDS ring, 1 (default): W3 in registers
        from glenet_amd import _lib
        _lib.load().glx_pointnet_feat_set_form(int(os.environ["GLX_POINTNET_FORM"]))
    print(json.dumps(bench.bench_config3(torch.device("cuda:0"))))
