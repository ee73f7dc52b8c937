"""The recorded CVAE training step's NaN narrowed down: with lr = 0 and a fixed eps every step from the
same state computes the same gradients, so replay k of the recorded step is compared with the EAGER step's flat gradient --
which replay first differs, in which parameters, by how much.  GLX_CVAE_ROWS_MAX=4000000 selects the row form at full size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import cvae_train as ct, dense_path as dp, synth  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1)
B = int(os.environ.get("B", 4096))
pts, box8, box7 = (torch.from_numpy(a).to(dev) for a in synth.cvae_objects(B, 2000, 512, with_labels=True))
model = dp.CVAE(4, 8).to(dev)
step = ct.CVAETrainStep(model, B, 512, lr=0.0)
step.load(pts, box8, box7, torch.randn((B, 8), device=dev))
names = [n for n, p in model.named_parameters() if p.requires_grad]
sizes = [p.numel() for n, p in model.named_parameters() if p.requires_grad]
for _ in range(2):
    step.enqueue()
torch.cuda.synchronize()
want = step.optimizer.flat_grad.clone()
for rep in range(int(os.environ.get("EAGER_STEPS", 4))):        # eager steps from the same state: the same gradients?
    step.optimizer.flat_grad.fill_(float("nan"))
    step.enqueue()
    torch.cuda.synchronize()
    got = step.optimizer.flat_grad
    print("eager step %d: loss %.6f, NaN entries %d, max |difference| / max |gradient| %.2e" % (
        rep, float(step.loss), int(torch.isnan(got).sum()),
        float(torch.nan_to_num(got - want, nan=float("inf")).abs().max() / want.abs().max())), flush=True)
loss0 = float(step.loss)
print("eager: loss %.6f grad norm %.4f finite %s" % (loss0, float(want.norm()), bool(torch.isfinite(want).all())), flush=True)
step.capture()
for rep in range(8):
    step.optimizer.flat_grad.fill_(float("nan"))
    step.step()
    torch.cuda.synchronize()
    got = step.optimizer.flat_grad
    nan = int(torch.isnan(got).sum())
    d = (got - want).abs()
    worst, off = [], 0
    for n, k in zip(names, sizes):
        e = d[off:off + k]
        m = float(torch.nan_to_num(e, nan=float("inf")).max())
        if m > 1e-3 * (float(want[off:off + k].abs().max()) + 1e-12):
            worst.append((n, "%.2e" % m))
        off += k
    print("replay %d: loss %.6f, NaN gradient entries %d, parameters off by > 1e-3 of their scale: %d %s" % (
        rep, float(step.loss), nan, len(worst), worst[:6]), flush=True)
