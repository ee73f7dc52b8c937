"""Which torch statements the CVAE training step (configs[3]) still launches, eager, one step under torch.profiler:
python3 tools/experiments/cvae_ops_census.py -> launches per op (CPU-side aten names with their device kernels' count and time)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from glenet_amd import cvae_train as ct, dense_path as dp, synth  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1)
B = 4096
pts, box8, box7 = (torch.from_numpy(a).to(dev) for a in synth.cvae_objects(B, 2000, 512, with_labels=True))
step = ct.CVAETrainStep(dp.CVAE(4, 8).to(dev), B, 512, lr=ct.OPTIM_CFG["LR"] / 10)
step.load(pts, box8, box7)
for _ in range(2):
    step.enqueue()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step.enqueue()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")]
# leaf aten ops that launched kernels, by name and shape
from collections import defaultdict  # noqa: E402
cnt, tim, where = defaultdict(int), defaultdict(float), {}
for e in ev:
    ks = e.kernels
    if not ks:
        continue
    kids = [c for c in e.cpu_children if c.name.startswith("aten::") and c.kernels]
    if kids:
        continue
    key = e.name
    cnt[key] += len(ks)
    tim[key] += sum(k.duration for k in ks)
    if key not in where and e.stack:
        where[key] = [f for f in e.stack if "glenet_amd" in f][:3]
tot = sum(cnt.values())
print("device launches from aten ops in one eager step: %d, %.2f ms" % (tot, sum(tim.values()) / 1e3))
for k in sorted(cnt, key=lambda k: -cnt[k])[:40]:
    print("%-40s launches %4d  %8.1f us   %s" % (k, cnt[k], tim[k], " <- ".join(w.split("/")[-1] for w in where.get(k, []))))
# by source line
byline = defaultdict(int)
for e in ev:
    if e.kernels and e.stack:
        kids = [c for c in e.cpu_children if c.name.startswith("aten::") and c.kernels]
        if kids:
            continue
        fr = [f for f in e.stack if "glenet_amd" in f]
        byline[fr[0].split("/")[-1] if fr else "(autograd / optimizer)"] += len(e.kernels)
print("\nby source line (first glenet_amd frame):")
for k in sorted(byline, key=lambda k: -byline[k])[:45]:
    print("%5d  %s" % (byline[k], k))
