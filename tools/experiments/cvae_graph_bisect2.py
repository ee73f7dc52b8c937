"""Which intermediate of the row-form PointNet extractors first differs between the first and the second replay of the
recorded CVAE training step (lr = 0, fixed eps: every replay must compute the same thing).  Every stage's output is cloned
inside the recorded step; after each replay the clones are compared with those of replay 0."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GLX_CVAE_ROWS_MAX", "4000000")
from glenet_amd import cvae_train as ct, dense_path as dp, synth  # noqa: E402
from glenet_amd.spconv import core  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1)
B = int(os.environ.get("B", 2048))
pts, box8, box7 = (torch.from_numpy(a).to(dev) for a in synth.cvae_objects(B, 2000, 512, with_labels=True))
model = dp.CVAE(4, 8).to(dev)
stash = []


def forward_rows(self, x):
    b, cin, p = x.shape
    rows = x.transpose(1, 2).reshape(b * p, cin)
    tag = "C%d" % self.conv3.out_channels
    stash.append((tag + " rows", rows.detach().clone()))
    h = self._rows_linear(rows, self.conv1)
    stash.append((tag + " lin1", h.detach().clone()))
    h = core.fused_train_bn(self.bn1, h, True, None)
    stash.append((tag + " bn1", h.detach().clone()))
    h = self._rows_linear(h, self.conv2)
    stash.append((tag + " lin2", h.detach().clone()))
    h = core.fused_train_bn(self.bn2, h, True, None)
    stash.append((tag + " bn2", h.detach().clone()))
    h = self._rows_linear(h, self.conv3)
    stash.append((tag + " lin3", h.detach().clone()))
    h = core.fused_train_bn(self.bn3, h, False, None)
    stash.append((tag + " bn3", h.detach().clone()))
    out = h.view(b, p, -1).amax(dim=1)
    stash.append((tag + " amax", out.detach().clone()))
    return out


dp.PointFeat._forward_train_rows = forward_rows
step = ct.CVAETrainStep(model, B, 512, lr=0.0)
step.load(pts, box8, box7, torch.randn((B, 8), device=dev))
real_enqueue = step.enqueue


def enqueue():
    stash.clear()
    return real_enqueue()


step.enqueue = enqueue
step.capture()
first = None
for rep in range(4):
    step.step()
    torch.cuda.synchronize()
    cur = [(n, t.clone()) for n, t in stash]
    if first is None:
        first = cur
        print("replay 0: loss %.6f, %d stages recorded" % (float(step.loss), len(cur)), flush=True)
        continue
    bad = [(n, "%.2e" % float(torch.nan_to_num(a - b, nan=float("inf")).abs().max())) for (n, a), (_, b) in zip(cur, first)
           if not torch.equal(a, b)]
    print("replay %d: loss %.6f, stages that differ from replay 0: %s" % (rep, float(step.loss), bad[:8]), flush=True)
