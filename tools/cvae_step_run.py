"""The recorded CVAE training step of BASELINE configs[3] (4096 objects x 512 points), 12 replays: for rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import cvae_train as ct, dense_path as dp, synth  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1)
B = int(os.environ.get("B", 4096))
pts, box8, box7 = (torch.from_numpy(a).to(dev) for a in synth.cvae_objects(B, 2000, 512, with_labels=True))
step = ct.CVAETrainStep(dp.CVAE(4, 8).to(dev), B, 512, lr=ct.OPTIM_CFG["LR"] / 10)
step.load(pts, box8, box7)
step.capture()
for _ in range(12):
    step.step()
torch.cuda.synchronize()
print("loss", float(step.loss))
