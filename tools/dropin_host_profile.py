"""Host-side profile (cProfile) of the eager exact-shape training step with the fast paths on (what bench.py's
`dropin_accelerated_step_ms` times): where the Python time of a step goes.  GPU box only."""
import cProfile, os, pstats, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import glenet_vr as gvr, synth  # noqa: E402

dev = torch.device("cuda", 0)
K = synth.KITTI
frames = [synth.kitti_frame(i) for i in range(4)]
pts = torch.from_numpy(np.concatenate([f[0] for f in frames])).to(dev)
bidx = torch.from_numpy(np.concatenate([np.full(len(f[0]), i, np.int32) for i, f in enumerate(frames)])).to(dev)
gt = torch.zeros(4, 16, 8, device=dev)
unc = torch.full((4, 16, 7), 0.05, device=dev)
for i, f in enumerate(frames):
    gt[i, :len(f[1]), :7] = torch.from_numpy(f[1]).to(dev)
    gt[i, :len(f[1]), 7] = 1
seed = torch.tensor([0.2, -0.15, 0.05, 0.1, -0.05, 0.03, 0.08], device=dev)
torch.manual_seed(0)
m = gvr.GLENetVR(K).to(dev).train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)


def step():
    m.zero_grad(set_to_none=True)
    loss, _ = m.training_step(pts, bidx, 4, gt, unc, seed_rois_with_gt=seed)
    loss.backward()
    opt.step()
    m.last = None


for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print("eager step: %.2f ms" % ((time.perf_counter() - t) / 10 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)

if os.environ.get("ITEM_CALLERS"):
    pr2 = cProfile.Profile()
    pr2.enable()
    step()
    torch.cuda.synchronize()
    pr2.disable()
    pstats.Stats(pr2).print_callers("item")
