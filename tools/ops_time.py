"""Kernel-level times of the remaining pcdet.ops entry points at the sizes SURVEY §8 lists for them
(rows a18-a20 and the PV-RCNN set-abstraction helpers): ball query, FPS, 3-NN + interpolation,
points-in-boxes, RoI-aware pooling, RoI-point pooling.  Device time with HIP events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import synth  # noqa: E402
from glenet_amd.pcdet_ops.pointnet2.pointnet2_stack import pointnet2_utils as pu  # noqa: E402
from glenet_amd.pcdet_ops.roiaware_pool3d import roiaware_pool3d_utils as ra  # noqa: E402
from glenet_amd.pcdet_ops.roipoint_pool3d import roipoint_pool3d_utils as rp  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(7)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


B, N, M = 4, 16384, 2048
xyz = torch.from_numpy(rng.uniform([0, -40, -3], [70.4, 40, 1], (B * N, 3)).astype(np.float32)).to(dev)
cnt = torch.full((B,), N, dtype=torch.int32, device=dev)
t = timeit(lambda: pu.stack_farthest_point_sample(xyz, cnt, M))
keep = pu.stack_farthest_point_sample(xyz, cnt, M).long()
print("stack FPS %d x %d -> %d: %9.1f us" % (B, N, M, t))
new_xyz = xyz[keep].contiguous()
ncnt = torch.full((B,), M, dtype=torch.int32, device=dev)
for r, ns in ((0.8, 16), (1.6, 32)):
    t = timeit(lambda: pu.ball_query(r, ns, xyz, cnt, new_xyz, ncnt))
    print("ball query r=%.1f nsample=%d, %d x (%d queries x %d points): %9.1f us" % (r, ns, B, M, N, t))
t = timeit(lambda: pu.three_nn(xyz, cnt, new_xyz, ncnt))
dist, idx = pu.three_nn(xyz, cnt, new_xyz, ncnt)
print("three_nn %d unknown x %d known per frame, %d frames: %9.1f us" % (N, M, B, t))
feat = torch.randn(B * M, 128, device=dev)
w = torch.softmax(-dist, dim=1).contiguous()
t = timeit(lambda: pu.three_interpolate(feat, idx, w))
print("three_interpolate (%d,128) -> (%d,128): %9.1f us" % (B * M, B * N, t))

pts = torch.from_numpy(rng.uniform([0, -40, -3], [70.4, 40, 1], (B, 20000, 3)).astype(np.float32)).to(dev)
boxes = torch.from_numpy(np.stack([synth.random_boxes(rng, 60, xy_range=35.0, near_dup=0.0) for _ in range(B)])).to(dev)
boxes[:, :, 0] += 35.0
t = timeit(lambda: ra.points_in_boxes_gpu(pts, boxes))
print("points_in_boxes_gpu %d x (20000 points x 60 boxes): %9.1f us" % (B, t))

rois = boxes[0, :, :7].repeat(3, 1)[:128].contiguous()
p1 = pts[0, :16384].contiguous()
pf = torch.randn(16384, 128, device=dev)
pool = ra.RoIAwarePool3d(out_size=14, max_pts_each_voxel=128)
for method in ("max", "avg"):
    t = timeit(lambda: pool(rois, p1, pf, pool_method=method), n=5)
    print("roiaware_pool3d %s: 128 RoIs x 16384 points x 128 channels -> 14^3: %9.1f us" % (method, t))

pp = rp.RoIPointPool3d(num_sampled_points=512, pool_extra_width=[0.0, 0.0, 0.0])
pfe = torch.randn(B, 16384, 128, device=dev)
b128 = boxes[:, :, :7].repeat(1, 3, 1)[:, :128].contiguous()
t = timeit(lambda: pp(pts[:, :16384].contiguous(), pfe, b128), n=5)
print("roipoint_pool3d %d x (16384 points x 128 boxes) -> 512 samples x 131: %9.1f us" % (B, t))
