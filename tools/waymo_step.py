"""Config-5 shaped side measurement (GPU box): 2 Waymo-shaped frames/GPU (180 k points, 5 features,
0.1 x 0.1 x 0.15 m voxels), VoxelResBackBone8x (17 SubM + 4 strided convs), shape-static graph."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import backbone as gb, synth  # noqa: E402

W = synth.WAYMO
dev = torch.device("cuda", 0)
B = 2
frames = [synth.waymo_frame(i)[0] for i in range(B)]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
torch.manual_seed(0)
grid = gb.gv.grid_size_of(W["point_cloud_range"], W["voxel_size"])
model = gb.VoxelResBackBone8x(W["num_features"], grid).to(dev).eval()
pipes = []
for _ in range(2):
    p = gb.StaticFramePipeline(model, W, B, pts.shape[0], W["num_features"], train_voxel_cap=False)
    p.calibrate(pts, bidx)
    p.load(pts, bidx)
    p.capture()
    pipes.append(p)
streams = [torch.cuda.Stream(dev) for _ in pipes]


def run(n):
    for i in range(n):
        with torch.cuda.stream(streams[i % 2]):
            pipes[i % 2].load(pts, bidx)
            pipes[i % 2].replay()


run(10)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 100
run(n)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
for p in pipes:
    p.check()
st = pipes[0].out["encoded_spconv_tensor"]
print("waymo-shaped: %d points, %d voxels in, %d out; %.3f ms/step, %.1f frames/s (2 frames/step)"
      % (pts.shape[0], int(pipes[0].out["voxel_index"].count.item()), int(st.count.item()), dt * 1e3, B / dt))
