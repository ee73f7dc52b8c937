"""Mirror of pcdet/ops/pointnet2/pointnet2_stack/voxel_query_utils.py."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import pointnet2_stack_cuda as pointnet2
from . import pointnet2_utils


class VoxelQuery(Function):
    @staticmethod
    def forward(ctx, max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
        """new_coords (M,4) int32 [b,z,y,x], point_indices (B,Z,Y,X) int32 dense voxel->point map
        -> idx (M,nsample) GLOBAL row indices, empty_ball_mask (voxel_query_utils.py:10-43)."""
        for t in (new_xyz, xyz, new_coords, point_indices):
            assert t.is_contiguous()
        m = new_coords.shape[0]
        _, z, y, x = point_indices.shape
        idx = torch.zeros((m, nsample), dtype=torch.int32, device=xyz.device)
        zr, yr, xr = max_range
        pointnet2.voxel_query_wrapper(m, z, y, x, nsample, radius, zr, yr, xr, new_xyz, xyz, new_coords,
                                      point_indices, idx)
        empty = idx[:, 0] == -1
        idx[empty] = 0
        return idx, empty

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


voxel_query = VoxelQuery.apply


def voxel_query_sparse(max_range, radius, nsample, xyz, new_xyz, new_coords, sparse_tensor):
    """voxel_query against the SparseConvTensor's own cell index: identical results to the dense
    map of common_utils.generate_voxel2pinds (pcdet/utils/common_utils.py:226-243) without the
    (B,Z,Y,X) int32 buffer (189 MB at KITTI x_conv2, refilled every step in the reference)."""
    index = sparse_tensor._ensure_index()
    z, y, x = sparse_tensor.spatial_shape
    m = new_coords.shape[0]
    idx = torch.zeros((m, nsample), dtype=torch.int32, device=xyz.device)
    zr, yr, xr = max_range
    pointnet2.voxel_query_index_wrapper(m, z, y, x, nsample, radius, zr, yr, xr, new_xyz.contiguous(),
                                        xyz.contiguous(), new_coords.contiguous(), index.bitmap,
                                        index.prefix, index.rank_to_row, idx)
    empty = idx[:, 0] == -1
    idx[empty] = 0
    return idx, empty


def voxel_query_raw(max_range, radius, nsample, xyz, new_xyz, new_coords, source):
    """The query kernel's own output: (M, nsample) GLOBAL row indices, idx[m,0] == -1 for an empty
    ball (the other slots of such a row are unspecified).  source: dense (B,Z,Y,X) map or a
    SparseConvTensor."""
    m = new_coords.shape[0]
    idx = torch.empty((m, nsample), dtype=torch.int32, device=xyz.device)
    zr, yr, xr = max_range
    if torch.is_tensor(source):
        z, y, x = source.shape[1:4]
        pointnet2.voxel_query_wrapper(m, z, y, x, nsample, radius, zr, yr, xr, new_xyz, xyz,
                                      new_coords, source.contiguous(), idx)
    else:
        index = source._ensure_index()
        z, y, x = source.spatial_shape
        pointnet2.voxel_query_index_wrapper(m, z, y, x, nsample, radius, zr, yr, xr, new_xyz, xyz,
                                            new_coords, index.bitmap, index.prefix, index.rank_to_row,
                                            idx)
    return idx


def rebase_to_frames(idx, xyz_batch_cnt, empty_mask):
    """Global -> per-frame indices (voxel_query_utils.py:85-91; equal M per frame, as there)."""
    b = xyz_batch_cnt.shape[0]
    starts = torch.cumsum(xyz_batch_cnt, 0) - xyz_batch_cnt
    local = idx.view(b, -1, idx.shape[1]) - starts.view(b, 1, 1).to(idx.dtype)
    local = local.view(-1, idx.shape[1])
    local[empty_mask] = 0
    return local.contiguous()


class VoxelQueryAndGrouping(nn.Module):
    def __init__(self, max_range, radius, nsample):
        super().__init__()
        self.max_range, self.radius, self.nsample = max_range, radius, nsample

    def forward(self, new_coords, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features,
                voxel2point_indices):
        """-> grouped_features (M,C,ns), grouped_xyz (M,3,ns), empty_ball_mask (M,)
        (voxel_query_utils.py:61-100).  voxel2point_indices: the dense (B,Z,Y,X) map, or a
        SparseConvTensor (then its cell index is queried instead)."""
        assert xyz.shape[0] == xyz_batch_cnt.sum() and new_coords.shape[0] == new_xyz_batch_cnt.sum()
        if torch.is_tensor(voxel2point_indices):
            idx, empty = voxel_query(self.max_range, self.radius, self.nsample, xyz, new_xyz, new_coords,
                                     voxel2point_indices)
        else:
            idx, empty = voxel_query_sparse(self.max_range, self.radius, self.nsample, xyz, new_xyz,
                                            new_coords, voxel2point_indices)
        idx = rebase_to_frames(idx, xyz_batch_cnt, empty)
        grouped_xyz = pointnet2_utils.grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        grouped_features = pointnet2_utils.grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)
        return grouped_features, grouped_xyz, empty
