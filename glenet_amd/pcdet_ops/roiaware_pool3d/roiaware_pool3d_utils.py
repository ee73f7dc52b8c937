"""Mirror of pcdet/ops/roiaware_pool3d/roiaware_pool3d_utils.py."""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import roiaware_pool3d_cuda

_METHODS = {'max': 0, 'avg': 1}


def _as_tensor(x):
    return (torch.from_numpy(x).float(), True) if isinstance(x, np.ndarray) else (x, False)


def points_in_boxes_cpu(points, boxes):
    """points (P,3), boxes (N,7) -> (N,P) int 0/1 (roiaware_pool3d_utils.py:9-26)."""
    assert boxes.shape[1] == 7 and points.shape[1] == 3
    points, was_np = _as_tensor(points)
    boxes, _ = _as_tensor(boxes)
    flags = points.new_zeros((boxes.shape[0], points.shape[0]), dtype=torch.int)
    roiaware_pool3d_cuda.points_in_boxes_cpu(boxes.float().contiguous(), points.float().contiguous(), flags)
    return flags.numpy() if was_np else flags


def points_in_boxes_gpu(points, boxes):
    """points (B,M,3), boxes (B,T,7) -> (B,M) index of the containing box, background -1
    (roiaware_pool3d_utils.py:29-43)."""
    assert boxes.shape[0] == points.shape[0] and boxes.shape[2] == 7 and points.shape[2] == 3
    owner = points.new_full(points.shape[:2], -1, dtype=torch.int)
    roiaware_pool3d_cuda.points_in_boxes_gpu(boxes.contiguous(), points.contiguous(), owner)
    return owner


class RoIAwarePool3dFunction(Function):
    @staticmethod
    def forward(ctx, rois, pts, pts_feature, out_size, max_pts_each_voxel, pool_method):
        assert rois.shape[1] == 7 and pts.shape[1] == 3
        ox, oy, oz = (out_size,) * 3 if isinstance(out_size, int) else tuple(out_size)
        shape = (rois.shape[0], ox, oy, oz)
        c = pts_feature.shape[-1]
        method = _METHODS[pool_method]
        # the kernels write every element of `pooled` and (max pooling) `argmax`: no zero fill of those
        # 2 x 180 MB at PartA2's sizes; avg pooling leaves argmax at the reference's zeros
        pooled = pts_feature.new_empty(shape + (c,))
        argmax = (pts_feature.new_empty if method == 0 else pts_feature.new_zeros)(shape + (c,), dtype=torch.int)
        lists = pts_feature.new_zeros(shape + (max_pts_each_voxel,), dtype=torch.int)
        roiaware_pool3d_cuda.forward(rois.contiguous(), pts.contiguous(), pts_feature.contiguous(),
                                     argmax, lists, pooled, method)
        ctx.roiaware_pool3d_for_backward = (lists, argmax, method, pts.shape[0], c)
        return pooled

    @staticmethod
    def backward(ctx, grad_out):
        lists, argmax, method, num_pts, c = ctx.roiaware_pool3d_for_backward
        grad_in = grad_out.new_zeros((num_pts, c))
        roiaware_pool3d_cuda.backward(lists, argmax, grad_out.contiguous(), grad_in, method)
        return None, None, grad_in, None, None, None


class RoIAwarePool3d(nn.Module):
    def __init__(self, out_size, max_pts_each_voxel=128):
        super().__init__()
        self.out_size = out_size
        self.max_pts_each_voxel = max_pts_each_voxel

    def forward(self, rois, pts, pts_feature, pool_method='max'):
        assert pool_method in _METHODS
        return RoIAwarePool3dFunction.apply(rois, pts, pts_feature, self.out_size,
                                            self.max_pts_each_voxel, pool_method)
