"""configs[3]'s recorded training step with one class switch off / on, alternating on one box:
python3 tools/experiments/cvae_ab.py Class.ATTR [pairs]   e.g.  LatentEncoder.TWO_HEADS_ONE_PRODUCT"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glenet_amd import cvae_train as ct, dense_path as dp, synth  # noqa: E402

cls_name, attr = sys.argv[1].split(".")
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cls = getattr(dp, cls_name, None) or getattr(ct, cls_name)
dev = torch.device("cuda", 0)
pts, box8, box7 = (torch.from_numpy(a).to(dev) for a in synth.cvae_objects(4096, 2000, 512, with_labels=True))
for rep in range(pairs):
    for val in (False, True):
        setattr(cls, attr, val)
        torch.manual_seed(1)
        step = ct.CVAETrainStep(dp.CVAE(4, 8).to(dev), 4096, 512, lr=ct.OPTIM_CFG["LR"] / 10)
        step.load(pts, box8, box7)
        step.capture()
        for _ in range(5):
            step.step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            step.step()
        e1.record()
        torch.cuda.synchronize()
        print("%s = %s: %.3f ms per step" % (sys.argv[1], val, e0.elapsed_time(e1) / 30), flush=True)
        del step
