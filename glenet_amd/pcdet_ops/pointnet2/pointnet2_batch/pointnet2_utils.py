"""Mirror of pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py (names, argument order, return arity):
the autograd operators of the batch-layout PointNet++ family on the HIP kernels.  B equal frames, xyz
(B, N, 3), features channel-major (B, C, N), indices int32 local to the frame."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_batch_cuda as pointnet2


def _i32(shape, like, zero=False):
    return (torch.zeros if zero else torch.empty)(shape, dtype=torch.int32, device=like.device)


def _f32(shape, like, zero=False):
    return (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=like.device)


class FarthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B, N, 3), N > npoint -> (B, npoint) int32 (pointnet2_utils.py:10-32)."""
        assert xyz.is_contiguous()
        b, n, _ = xyz.size()
        out = _i32((b, npoint), xyz)
        temp = torch.full((b, n), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.farthest_point_sampling_wrapper(b, n, npoint, xyz, temp, out)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B, C, N), idx (B, npoint) -> (B, C, npoint) (pointnet2_utils.py:38-70)."""
        assert features.is_contiguous() and idx.is_contiguous()
        b, npoint = idx.size()
        _, c, n = features.size()
        out = _f32((b, c, npoint), features)
        pointnet2.gather_points_wrapper(b, c, n, npoint, features, idx, out)
        ctx.for_backwards = (idx, c, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, c, n = ctx.for_backwards
        b, npoint = idx.size()
        grad_features = _f32((b, c, n), grad_out, zero=True)
        pointnet2.gather_points_grad_wrapper(b, c, n, npoint, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B, N, 3), known (B, M, 3) -> dist (B, N, 3) (l2, not squared), idx (B, N, 3)
        (pointnet2_utils.py:76-101)."""
        assert unknown.is_contiguous() and known.is_contiguous()
        b, n, _ = unknown.size()
        m = known.size(1)
        dist2 = _f32((b, n, 3), unknown)
        idx = _i32((b, n, 3), unknown)
        pointnet2.three_nn_wrapper(b, n, m, unknown, known, dist2, idx)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B, C, M), idx / weight (B, n, 3) -> (B, C, n) (pointnet2_utils.py:107-148)."""
        assert features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
        b, c, m = features.size()
        n = idx.size(1)
        ctx.three_interpolate_for_backward = (idx, weight, m)
        out = _f32((b, c, n), features)
        pointnet2.three_interpolate_wrapper(b, c, m, n, features, idx, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, m = ctx.three_interpolate_for_backward
        b, c, n = grad_out.size()
        grad_features = _f32((b, c, m), grad_out, zero=True)
        pointnet2.three_interpolate_grad_wrapper(b, c, n, m, grad_out.contiguous(), idx, weight, grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B, C, N), idx (B, npoint, nsample) -> (B, C, npoint, nsample) (pointnet2_utils.py:154-192)."""
        assert features.is_contiguous() and idx.is_contiguous()
        b, npoint, nsample = idx.size()
        _, c, n = features.size()
        out = _f32((b, c, npoint, nsample), features)
        pointnet2.group_points_wrapper(b, c, n, npoint, nsample, features, idx, out)
        ctx.for_backwards = (idx, n)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, n = ctx.for_backwards
        b, c, npoint, nsample = grad_out.size()
        grad_features = _f32((b, c, n), grad_out, zero=True)
        pointnet2.group_points_grad_wrapper(b, c, n, npoint, nsample, grad_out.contiguous(), idx, grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """xyz (B, N, 3), new_xyz (B, npoint, 3) -> idx (B, npoint, nsample); a ball without points keeps the
        zero row (pointnet2_utils.py:198-222)."""
        assert new_xyz.is_contiguous() and xyz.is_contiguous()
        b, n, _ = xyz.size()
        npoint = new_xyz.size(1)
        idx = _i32((b, npoint, nsample), xyz, zero=True)
        pointnet2.ball_query_wrapper(b, n, npoint, radius, nsample, new_xyz, xyz, idx)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """Ball query + grouping of coordinates (relative to the centroid) and features
    (pointnet2_utils.py:228-265) -> (B, 3 + C, npoint, nsample)."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz
        grouped_features = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features


class GroupAll(nn.Module):
    """One group holding every point (pointnet2_utils.py:268-290) -> (B, 3 + C, 1, N)."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
