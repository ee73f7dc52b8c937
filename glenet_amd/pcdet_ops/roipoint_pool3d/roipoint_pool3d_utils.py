"""Mirror of pcdet/ops/roipoint_pool3d/roipoint_pool3d_utils.py."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import roipoint_pool3d_cuda


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    """pcdet/utils/box_utils.py:145-158: grow dx, dy, dz by extra_width."""
    big = boxes3d.clone()
    big[:, 3:6] += boxes3d.new_tensor(extra_width).reshape(1, -1)    # a triple (the configs') or one width for all
    return big


class RoIPointPool3dFunction(Function):
    @staticmethod
    def forward(ctx, points, point_features, boxes3d, pool_extra_width, num_sampled_points=512):
        """points (B,N,3), point_features (B,N,C), boxes3d (B,M,7) ->
        pooled (B,M,S,3+C), empty flag (B,M) (roipoint_pool3d_utils.py:31-60)."""
        assert points.dim() == 3 and points.shape[2] == 3
        b, m, c = points.shape[0], boxes3d.shape[1], point_features.shape[2]
        grown = enlarge_box3d(boxes3d.view(-1, 7), pool_extra_width).view(b, -1, 7)
        pooled = point_features.new_zeros((b, m, num_sampled_points, 3 + c))
        empty = point_features.new_zeros((b, m)).int()
        roipoint_pool3d_cuda.forward(points.contiguous(), grown.contiguous(),
                                     point_features.contiguous(), pooled, empty)
        return pooled, empty

    @staticmethod
    def backward(ctx, grad_out):
        raise NotImplementedError   # as in the reference (:62-63)


class RoIPointPool3d(nn.Module):
    def __init__(self, num_sampled_points=512, pool_extra_width=1.0):
        super().__init__()
        self.num_sampled_points = num_sampled_points
        self.pool_extra_width = pool_extra_width

    def forward(self, points, point_features, boxes3d):
        return RoIPointPool3dFunction.apply(points, point_features, boxes3d, self.pool_extra_width,
                                            self.num_sampled_points)
