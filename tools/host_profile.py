"""cProfile of the bench step on the GPU box (host overhead hunt)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import backbone as gb, synth  # noqa: E402

K = synth.KITTI
dev = torch.device("cuda", 0)
frames = [synth.kitti_frame(i)[0] for i in range(4)]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
torch.manual_seed(0)
grid = gb.gv.grid_size_of(K["point_cloud_range"], K["voxel_size"])
model = gb.VoxelBackBone8x(4, grid).to(dev).eval()
vfe, hc = gb.MeanVFE(), gb.HeightCompression()


def step():
    with torch.no_grad():
        bd = gb.voxelize_batch(pts, bidx, 4, K, train=True)
        bd = vfe(bd)
        bd = model(bd)
        bd = hc(bd)
    return bd


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print("ms/step (no hooks): %.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
