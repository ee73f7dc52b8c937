"""configs[3] alone: python tools/config3_time.py -> the bench's config3 object as one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(bench.bench_config3(torch.device("cuda:0"))))
