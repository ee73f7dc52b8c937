"""Time forward and forward+backward of the sparse backbone on the bench batch (GPU box)."""
import os
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
model = gb.VoxelBackBone8x(4, grid).to(dev).train()
vfe, hc = gb.MeanVFE(), gb.HeightCompression()


def step(backward):
    bd = gb.voxelize_batch(pts, bidx, 4, K)
    bd = vfe(bd)
    bd = hc(model(bd))
    if backward:
        model.zero_grad(set_to_none=True)
        bd["spatial_features"].square().mean().backward()


for name, bw in (("fwd (train mode, BN batch stats)", False), ("fwd+bwd", True),
                 ("fwd again", False), ("fwd+bwd again", True)):
    for _ in range(3):
        step(bw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        step(bw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-36s %8.3f ms/step  %8.1f frames/s" % (name, dt * 1e3, 4 / dt))

# ---- the same step on the shape-static path: no read-backs, then replayed as one HIP graph
pipe = gb.StaticTrainPipeline(model, K, 4, pts.shape[0], 4)
pipe.calibrate(pts, bidx)
pipe.load(pts, bidx)
for name, graph in (("fwd+bwd shape-static, eager", False), ("fwd+bwd shape-static, HIP graph", True)):
    if graph:
        pipe.capture()
    fn = pipe.replay if graph else pipe.enqueue
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    pipe.check()
    print("%-36s %8.3f ms/step  %8.1f frames/s" % (name, dt * 1e3, 4 / dt))

# ---- with the optimizer inside the graph (Adam, capturable): a complete backbone training step
import copy  # noqa: E402
model2 = copy.deepcopy(model)
opt = torch.optim.Adam(model2.parameters(), lr=1e-4, capturable=True, foreach=True)  # fused=True does not update under replay (ROCm 7.2 / torch 2.10)
pipe2 = gb.StaticTrainPipeline(model2, K, 4, pts.shape[0], 4, optimizer=opt)
pipe2.calibrate(pts, bidx)
pipe2.load(pts, bidx)
pipe2.capture()
for _ in range(3):
    pipe2.replay()
torch.cuda.synchronize()
l0 = float(pipe2.loss.detach())
t0 = time.perf_counter()
for _ in range(20):
    pipe2.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
pipe2.check()
print("%-36s %8.3f ms/step  %8.1f frames/s   (loss %.5f -> %.5f over 20 steps)" % (
    "fwd+bwd+Adam, HIP graph", dt * 1e3, 4 / dt, l0, float(pipe2.loss.detach())))
