"""How long is the DATA stage of the recorded training step (load + voxelize + MeanVFE + all rule tables) on its own,
as a HIP graph?  That is what a cross-step prefetch could take off the critical path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from glenet_amd import backbone as gb, glenet_vr as gvr, synth, _lib

dev = torch.device("cuda", 0)
K = synth.KITTI
B = 4
frames = [synth.kitti_frame(1000 + i)[0] for i in range(B)]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
model = gvr.GLENetVR(K).to(dev).train()
pipe = gvr.StaticTrainStep(model, B, pts.shape[0] + 2000, max_gt=32, lr=1e-4)
pipe.calibrate(pts, bidx)
pipe.load(pts, bidx)


def data_stage():
    with torch.no_grad():
        bd = gb.voxelize_batch(pipe.points, pipe.batch_idx, B, K, train=pipe.train_cap, static=True)
        plan = pipe.model.plan(bd["voxel_coords"], B, index=bd["voxel_index"], capacities=pipe.capacities)
        bd = pipe.vfe(bd)
    return bd, plan


side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    for _ in range(3):
        data_stage()
torch.cuda.synchronize()
g = _lib.new_graph()
with torch.cuda.graph(g, stream=side):
    keep = data_stage()
torch.cuda.synchronize()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(100):
    g.replay()
e.record()
torch.cuda.synchronize()
print("data stage (voxelize + MeanVFE + every rule table incl. the BEV first layer's), one graph: %.3f ms" % (s.elapsed_time(e) / 100))
