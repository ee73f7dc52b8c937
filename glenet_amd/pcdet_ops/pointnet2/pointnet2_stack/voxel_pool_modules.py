"""Mirror of pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py: the RoI-grid pooling set
abstraction of Voxel-RCNN (called from voxelrcnn_head.py:106-191).  Same constructor keywords,
submodule names (groupers / mlps_in / mlps_pos / mlps_out, so checkpoints load) and forward
signature; the query + grouping run on glenet_amd kernels."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import voxel_query_utils


def _conv_bn(cin, cout, dims, relu=False):
    conv = (nn.Conv1d if dims == 1 else nn.Conv2d)(cin, cout, kernel_size=1, bias=False)
    bn = (nn.BatchNorm1d if dims == 1 else nn.BatchNorm2d)(cout)
    return nn.Sequential(conv, bn, nn.ReLU()) if relu else nn.Sequential(conv, bn)


class NeighborVoxelSAModuleMSG(nn.Module):
    def __init__(self, *, query_ranges, radii, nsamples, mlps, use_xyz=True, pool_method='max_pool'):
        super().__init__()
        assert len(query_ranges) == len(nsamples) == len(mlps)
        self.groupers, self.mlps_in = nn.ModuleList(), nn.ModuleList()
        self.mlps_pos, self.mlps_out = nn.ModuleList(), nn.ModuleList()
        for rng, radius, nsample, (c_in, c_mid, c_out) in zip(query_ranges, radii, nsamples, mlps):
            self.groupers.append(voxel_query_utils.VoxelQueryAndGrouping(rng, radius, nsample))
            self.mlps_in.append(_conv_bn(c_in, c_mid, 1))
            self.mlps_pos.append(_conv_bn(3, c_mid, 2))
            self.mlps_out.append(_conv_bn(c_mid, c_out, 1, relu=True))
        self.relu = nn.ReLU()
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, new_coords, features,
                voxel2point_indices):
        """xyz (N,3) voxel centres, new_xyz (M,3) grid points, new_coords (M,4) [b,x,y,z] voxel
        coords of the grid points, features (N,C), voxel2point_indices: dense (B,Z,Y,X) map or the
        SparseConvTensor itself -> (M, sum of mlps[k][-1])."""
        coords_bzyx = new_coords[:, [0, 3, 2, 1]].contiguous()      # voxel_pool_modules.py:84
        outs = []
        for grouper, mlp_in, mlp_pos, mlp_out in zip(self.groupers, self.mlps_in, self.mlps_pos,
                                                     self.mlps_out):
            feats = mlp_in(features.t().unsqueeze(0)).squeeze(0).t().contiguous()     # (N, c_mid)
            g_feat, g_xyz, empty = grouper(coords_bzyx, xyz, xyz_batch_cnt, new_xyz,
                                           new_xyz_batch_cnt, feats, voxel2point_indices)
            g_feat[empty] = 0
            rel = g_xyz - new_xyz.unsqueeze(-1)
            rel[empty] = 0
            pos = mlp_pos(rel.permute(1, 0, 2).unsqueeze(0))                           # (1,c,M,ns)
            x = self.relu(g_feat.permute(1, 0, 2).unsqueeze(0) + pos)
            if self.pool_method == 'max_pool':
                x = F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)
            elif self.pool_method == 'avg_pool':
                x = F.avg_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)
            else:
                raise NotImplementedError
            outs.append(mlp_out(x).squeeze(0).t())
        return torch.cat(outs, dim=1)
