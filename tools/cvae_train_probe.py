"""GPU probe: loss trajectory of the full-size CVAE training step (4096 x 512) at the schedule's first and peak lr."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import cvae_train as ct, dense_path as dp, synth  # noqa: E402

dev = torch.device("cuda", 0)
pts, box8, box7 = (torch.from_numpy(a).to(dev) for a in synth.cvae_objects(4096, 2000, 512, with_labels=True))
for lr in (3e-4, 3e-3):
    torch.manual_seed(1)
    model = dp.CVAE(4, 8).to(dev)
    step = ct.CVAETrainStep(model, 4096, 512, lr=lr)
    step.load(pts, box8, box7)
    out = []
    for i in range(8):
        step.enqueue()
        out.append([round(float(t), 4) for t in step.terms] + [round(float(step.optimizer.grad_norm), 3)])
    print("lr", lr, out, flush=True)

# the same at the schedule's first lr as ONE recorded graph (what bench.py times), after an eval-mode sampler pass
torch.manual_seed(1)
model = dp.CVAE(4, 8).to(dev).eval()
with torch.no_grad():
    model.sample(pts)
step = ct.CVAETrainStep(model, 4096, 512, lr=3e-4)
step.load(pts, box8, box7)
step.capture()
out = []
for i in range(14):
    step.step()
    out.append([round(float(t), 4) for t in step.terms] + [round(float(step.optimizer.grad_norm), 3)])
print("graph lr 3e-4", out, flush=True)
