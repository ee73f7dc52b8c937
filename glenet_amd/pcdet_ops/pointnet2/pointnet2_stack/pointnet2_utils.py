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


GATHER_MIN_REFS_PER_ROW = 4     # above this many (grid point, slot) references per feature row: gather backward


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
        if m * nsample >= GATHER_MIN_REFS_PER_ROW * max(n, 1):
            # many references per feature row (RoI-grid pooling): the atomic scatter serialises on the
            # hot rows, the gather form does not (csrc/glx_points.hip, k_gp_gather)
            grad_features = torch.empty((n, c), dtype=torch.float32, device=grad_out.device)
            pointnet2.group_points_grad_gather_wrapper(b, m, c, n, nsample, grad_out.contiguous(), idx,
                                                       idx_batch_cnt, features_batch_cnt, grad_features)
            return grad_features, None, None, None
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


class StackFarthestPointSampling(Function):
    """pointnet2_utils.py:190-222: xyz (N1+N2+..,3), xyz_batch_cnt, npoint (int | list | tensor)
    -> (sum npoint,) int32 indices into the stacked xyz."""

    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, npoint):
        assert xyz.is_contiguous() and xyz.shape[1] == 3
        B = len(xyz_batch_cnt)
        if not isinstance(npoint, torch.Tensor):
            npoint = torch.tensor(npoint if isinstance(npoint, list) else [npoint] * B, device=xyz.device)
        npoint = npoint.int().contiguous()
        temp = torch.full((xyz.shape[0],), 1e10, dtype=torch.float32, device=xyz.device)
        cnt = _int(xyz_batch_cnt)
        # one read-back for both host-side numbers: the output length and the largest frame (which picks
        # the register-resident kernel)
        total, largest = torch.stack([npoint.sum(), cnt.max().to(npoint.dtype)]).tolist()
        out = torch.empty(int(total), dtype=torch.int32, device=xyz.device)
        pointnet2.stack_farthest_point_sampling_wrapper(xyz, temp, cnt, out, npoint, max_points=int(largest))
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None


stack_farthest_point_sample = StackFarthestPointSampling.apply


class FarthestPointSampling(Function):
    """pointnet2_utils.py:162-187: xyz (B,N,3) -> (B,npoint) int32 per-frame indices."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        assert xyz.is_contiguous()
        B, N, _ = xyz.shape
        out = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.farthest_point_sampling_wrapper(B, N, npoint, xyz, temp, out)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


class ThreeNN(Function):
    """pointnet2_utils.py:227-258 -> (dist (N,3) euclidean, idx (N,3) int32 global)."""

    @staticmethod
    def forward(ctx, unknown, unknown_batch_cnt, known, known_batch_cnt):
        assert unknown.dim() == 2 and unknown.shape[1] == 3 and known.dim() == 2 and known.shape[1] == 3
        assert len(unknown_batch_cnt) == len(known_batch_cnt)
        dist2 = unknown.new_zeros(unknown.shape)
        idx = torch.zeros(unknown.shape, dtype=torch.int32, device=unknown.device)
        pointnet2.three_nn_wrapper(unknown.contiguous(), _int(unknown_batch_cnt), known.contiguous(),
                                   _int(known_batch_cnt), dist2, idx)
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """pointnet2_utils.py:264-300: features (M,C), idx (N,3), weight (N,3) -> (N,C)."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        assert idx.shape[0] == weight.shape[0] and idx.shape[1] == weight.shape[1] == 3
        ctx.save = (idx, weight, features.shape[0])
        out = features.new_zeros((idx.shape[0], features.shape[1]))
        pointnet2.three_interpolate_wrapper(features.contiguous(), _int(idx), weight.contiguous(), out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, M = ctx.save
        g = grad_out.new_zeros((M, grad_out.shape[1]))
        pointnet2.three_interpolate_grad_wrapper(grad_out.contiguous(), _int(idx), weight.contiguous(), g)
        return g, None, None


three_interpolate = ThreeInterpolate.apply


class ThreeNNForVectorPoolByTwoStep(Function):
    """pointnet2_utils.py:306-352: local neighbour lists (with the host retry loop that grows the buffer), then
    the three nearest list members of every grid centre -> (dist (M,G,3), idx (M,G,3) int32, avg_length tensor)."""

    @staticmethod
    def forward(ctx, support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt,
                max_neighbour_distance, nsample, neighbor_type, avg_length_of_neighbor_idxs, num_total_grids,
                neighbor_distance_multiplier):
        num_new_xyz = new_xyz.shape[0]
        new_xyz_grid_dist2 = new_xyz_grid_centers.new_zeros(new_xyz_grid_centers.shape)
        new_xyz_grid_idxs = new_xyz_grid_centers.new_zeros(new_xyz_grid_centers.shape).int().fill_(-1)
        while True:
            num_max_sum_points = avg_length_of_neighbor_idxs * num_new_xyz
            stack_neighbor_idxs = new_xyz_grid_idxs.new_zeros(max(num_max_sum_points, 1))
            start_len = new_xyz_grid_idxs.new_zeros(num_new_xyz, 2).int()
            cumsum = new_xyz_grid_idxs.new_zeros(1)
            pointnet2.query_stacked_local_neighbor_idxs_wrapper_stack(
                support_xyz.contiguous(), _int(xyz_batch_cnt).contiguous(), new_xyz.contiguous(),
                _int(new_xyz_batch_cnt).contiguous(), stack_neighbor_idxs, start_len, cumsum,
                avg_length_of_neighbor_idxs, max_neighbour_distance * neighbor_distance_multiplier, nsample,
                neighbor_type)
            total = int(cumsum[0].item())
            avg_length_of_neighbor_idxs = total // num_new_xyz + int(total % num_new_xyz > 0)
            if total <= num_max_sum_points:
                break
        stack_neighbor_idxs = stack_neighbor_idxs[:max(total, 1)]
        pointnet2.query_three_nn_by_stacked_local_idxs_wrapper_stack(
            support_xyz.contiguous(), new_xyz.contiguous(), new_xyz_grid_centers.contiguous(), new_xyz_grid_idxs,
            new_xyz_grid_dist2, stack_neighbor_idxs.contiguous(), start_len, num_new_xyz, num_total_grids)
        return torch.sqrt(new_xyz_grid_dist2), new_xyz_grid_idxs, torch.tensor(avg_length_of_neighbor_idxs)

    @staticmethod
    def backward(ctx, *a):
        return (None,) * 11


three_nn_for_vector_pool_by_two_step = ThreeNNForVectorPoolByTwoStep.apply


class VectorPoolWithVoxelQuery(Function):
    """pointnet2_utils.py:358-448: sub-voxel average (pooling_type 0) / first-point (1) pooling of the support
    features around every new point -> (new_features (M, G * c_each), new_local_xyz (M, 3G),
    num_mean_points_per_grid, point_cnt_of_grid); gradient w.r.t. support_features."""

    @staticmethod
    def forward(ctx, support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, num_grid_x, num_grid_y,
                num_grid_z, max_neighbour_distance, num_c_out_each_grid, use_xyz, num_mean_points_per_grid=100,
                nsample=-1, neighbor_type=0, pooling_type=0):
        for t in (support_xyz, support_features, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
            assert t.is_contiguous()
        num_total_grids = num_grid_x * num_grid_y * num_grid_z
        num_c_out = num_c_out_each_grid * num_total_grids
        N, num_c_in = support_features.shape
        M = new_xyz.shape[0]
        assert num_c_in % num_c_out_each_grid == 0, \
            'the input channels (%d) should be an integral multiple of num_c_out_each_grid(%d)' % (num_c_in, num_c_out_each_grid)
        xbc, nbc = _int(xyz_batch_cnt), _int(new_xyz_batch_cnt)
        while True:
            new_features = support_features.new_zeros((M, num_c_out))
            new_local_xyz = support_features.new_zeros((M, 3 * num_total_grids))
            point_cnt_of_grid = xbc.new_zeros((M, num_total_grids))
            num_max_sum_points = num_mean_points_per_grid * M
            grouped_idxs = xbc.new_zeros((max(num_max_sum_points, 1), 3))
            num_cum_sum = pointnet2.vector_pool_wrapper(
                support_xyz, xbc, support_features, new_xyz, nbc, new_features, new_local_xyz, point_cnt_of_grid,
                grouped_idxs, num_grid_x, num_grid_y, num_grid_z, max_neighbour_distance, use_xyz, num_max_sum_points,
                nsample, neighbor_type, pooling_type)
            num_mean_points_per_grid = num_cum_sum // M + int(num_cum_sum % M > 0)
            if num_cum_sum <= num_max_sum_points:
                break
        grouped_idxs = grouped_idxs[:num_cum_sum]
        normalizer = torch.clamp_min(point_cnt_of_grid[:, :, None].float(), min=1e-6)
        new_features = (new_features.view(-1, num_total_grids, num_c_out_each_grid) / normalizer).view(-1, num_c_out)
        if use_xyz:
            new_local_xyz = (new_local_xyz.view(-1, num_total_grids, 3) / normalizer).view(-1, num_total_grids * 3)
        num_mean_points_per_grid = torch.Tensor([num_mean_points_per_grid]).int()
        nsample = torch.Tensor([nsample]).int()
        ctx.vector_pool_for_backward = (point_cnt_of_grid, grouped_idxs, N, num_c_in)
        ctx.mark_non_differentiable(new_local_xyz, num_mean_points_per_grid, nsample, point_cnt_of_grid)
        return new_features, new_local_xyz, num_mean_points_per_grid, point_cnt_of_grid

    @staticmethod
    def backward(ctx, grad_new_features, grad_local_xyz, grad_num_cum_sum, grad_point_cnt_of_grid):
        point_cnt_of_grid, grouped_idxs, N, num_c_in = ctx.vector_pool_for_backward
        grad_support_features = grad_new_features.new_zeros((N, num_c_in))
        if grouped_idxs.shape[0] > 0:
            pointnet2.vector_pool_grad_wrapper(grad_new_features.contiguous(), point_cnt_of_grid,
                                               grouped_idxs.contiguous(), grad_support_features)
        return (None, None, grad_support_features) + (None,) * 12


vector_pool_with_voxel_query_op = VectorPoolWithVoxelQuery.apply
