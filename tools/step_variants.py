"""Time variants of the single-stream step on the GPU box (where does the time go?)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import backbone as gb, synth, voxelize as gv  # noqa: E402

K = synth.KITTI
dev = torch.device("cuda", 0)
frames = [synth.kitti_frame(i)[0] for i in range(4)]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
torch.manual_seed(0)
grid = gb.gv.grid_size_of(K["point_cloud_range"], K["voxel_size"])
model = gb.VoxelBackBone8x(4, grid).to(dev).eval()
vfe, hc = gb.MeanVFE(), gb.HeightCompression()


def vox_old():
    v, c, n, offs = gv.hard_voxelize(pts, K["voxel_size"], K["point_cloud_range"], 5, 16000, batch_idx=bidx, batch_size=4)
    return dict(voxels=v, voxel_coords=c, voxel_num_points=n, batch_size=4)


def step(share_index, plan, dense=True):
    with torch.no_grad():
        bd = gb.voxelize_batch(pts, bidx, 4, K) if share_index else vox_old()
        bd = vfe(bd)
        if plan:
            bd["rule_plan"] = model.plan(bd["voxel_coords"], 4, index=bd.get("voxel_index"))
        bd = model(bd)
        if dense:
            bd = hc(bd)
    return bd


def timeit(name, fn, n=60):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print("%-40s %.3f ms/step" % (name, (time.perf_counter() - t0) / n * 1e3))


timeit("own index, lazy rules", lambda: step(False, False))
timeit("shared index, lazy rules", lambda: step(True, False))
timeit("shared index, planned rules", lambda: step(True, True))
timeit("own index, planned rules", lambda: step(False, True))
timeit("voxelize only (shared)", lambda: gb.voxelize_batch(pts, bidx, 4, K))
timeit("voxelize only (own)", vox_old)
bd0 = vfe(gb.voxelize_batch(pts, bidx, 4, K))
timeit("plan only", lambda: model.plan(bd0["voxel_coords"], 4, index=bd0["voxel_index"]))
plan = model.plan(bd0["voxel_coords"], 4, index=bd0["voxel_index"])


def convs_only():
    with torch.no_grad():
        b = dict(bd0)
        b["rule_plan"] = plan
        hc(model(b))


timeit("convs + dense only (rules given)", convs_only)
