"""Mirror of the on-path part of pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py:
BallQuery, GroupingOperation, QueryAndGroup."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_stack_cuda as pointnet2


def _int(t):
    return t if t.dtype == torch.int32 else t.int()


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
        """xyz (N,3) stacked with xyz_batch_cnt (B,), new_xyz (M,3) with new_xyz_batch_cnt (B,) ->
        idx (M,nsample) int32 local to the frame, empty_ball_mask (M,) (pointnet2_utils.py:8-49)."""
        for t in (new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt):
            assert t.is_contiguous()
        idx = torch.zeros((new_xyz.shape[0], nsample), dtype=torch.int32, device=xyz.device)
        pointnet2.ball_query_wrapper(xyz_batch_cnt.shape[0], new_xyz.shape[0], radius, nsample, new_xyz,
                                     _int(new_xyz_batch_cnt), xyz, _int(xyz_batch_cnt), idx)
        empty = idx[:, 0] == -1
        idx[empty] = 0
        ctx.mark_non_differentiable(idx)
        ctx.mark_non_differentiable(empty)
        return idx, empty

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None


ball_query = BallQuery.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, features_batch_cnt, idx, idx_batch_cnt):
        """features (N,C), idx (M,nsample) -> (M,C,nsample) (pointnet2_utils.py:52-109)."""
        for t in (features, features_batch_cnt, idx, idx_batch_cnt):
            assert t.is_contiguous()
        assert features.shape[0] == features_batch_cnt.sum(), \
            'features: %s, features_batch_cnt: %s' % (str(features.shape), str(features_batch_cnt))
        assert idx.shape[0] == idx_batch_cnt.sum(), \
            'idx: %s, idx_batch_cnt: %s' % (str(idx.shape), str(idx_batch_cnt))
        m, nsample = idx.size()
        n, c = features.size()
        b = idx_batch_cnt.shape[0]
        out = torch.empty((m, c, nsample), dtype=torch.float32, device=features.device)
        pointnet2.group_points_wrapper(b, m, c, nsample, features, _int(features_batch_cnt), idx,
                                       _int(idx_batch_cnt), out)
        ctx.for_backwards = (b, n, idx, _int(features_batch_cnt), _int(idx_batch_cnt))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        b, n, idx, features_batch_cnt, idx_batch_cnt = ctx.for_backwards
        m, c, nsample = grad_out.size()
        grad_features = torch.zeros((n, c), dtype=torch.float32, device=grad_out.device)
        pointnet2.group_points_grad_wrapper(b, m, c, n, nsample, grad_out.contiguous(), idx,
                                            idx_batch_cnt, features_batch_cnt, grad_features)
        return grad_features, None, None, None


grouping_operation = GroupingOperation.apply


class QueryAndGroup(nn.Module):
    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None):
        """-> new_features (M, 3+C | C | 3, nsample), idx (pointnet2_utils.py:125-160)."""
        assert xyz.shape[0] == xyz_batch_cnt.sum() and new_xyz.shape[0] == new_xyz_batch_cnt.sum()
        idx, empty = ball_query(self.radius, self.nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
        rel = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt) - new_xyz.unsqueeze(-1)
        rel[empty] = 0
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return rel, idx
        grouped = grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        grouped[empty] = 0
        return (torch.cat([rel, grouped], dim=1) if self.use_xyz else grouped), idx
