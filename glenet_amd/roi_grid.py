"""RoI-grid pooling of Voxel-RCNN (the "RoI-grid point pooling" of north_star): geometry glue +
the multi-scale pool, our counterpart of VoxelRCNNHead.roi_grid_pool and its helpers
(pcdet/models/roi_heads/voxelrcnn_head.py:106-215, pcdet/utils/common_utils.py:41-82,226-243).

Differences by design: the dense (B,Z,Y,X) int32 voxel->point map the reference rebuilds per scale
and step (generate_voxel2pinds, 189 MB at x_conv2) is not built -- the query walks the sparse
tensor's cell index (glx_voxel_query_index); per-batch counts are computed without a Python loop.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from .pcdet_ops.pointnet2.pointnet2_stack import voxel_pool_modules


def get_voxel_centers(voxel_coords, downsample_times, voxel_size, point_cloud_range):
    """voxel_coords (N,3) [z,y,x] -> centres (N,3) xyz = (idx + 0.5) * voxel * stride + range_min."""
    assert voxel_coords.shape[1] == 3
    xyz = voxel_coords[:, [2, 1, 0]].float()
    size = torch.tensor(voxel_size, device=xyz.device).float() * downsample_times
    origin = torch.tensor(point_cloud_range[0:3], device=xyz.device).float()
    return (xyz + 0.5) * size + origin


def _voxel_centers_capturable(voxel_coords, downsample_times, voxel_size, point_cloud_range):
    """get_voxel_centers without host->device copies (no index lists, no torch.tensor(...) of the python
    constants), so that it can be recorded into a HIP graph; same float32 arithmetic step by step: the
    scaled voxel size is the float32 product the tensor expression forms, applied as a scalar."""
    import numpy as np
    c = voxel_coords
    cols = []
    for i, col in enumerate((2, 1, 0)):
        size = float(np.float32(voxel_size[i]) * np.float32(downsample_times))
        cols.append((c[:, col].float() + 0.5) * size + float(np.float32(point_cloud_range[i])))
    return torch.stack(cols, dim=1)


def rotate_points_along_z(points, angle):
    """points (B,N,3+), angle (B,) -> rotated about z by the matrix [[c, s],[-s, c]] applied on the
    right (common_utils.py:41-63)."""
    c, s = torch.cos(angle), torch.sin(angle)
    x, y = points[..., 0], points[..., 1]
    out = points.clone()
    out[..., 0] = x * c[:, None] - y * s[:, None]
    out[..., 1] = x * s[:, None] + y * c[:, None]
    return out


def dense_grid_points(rois, grid_size):
    """rois (R,7+) -> (R, G^3, 3) grid points in the box frame: (idx + 0.5)/G * size - size/2 with
    idx enumerated x-major (nonzero() order of a (G,G,G) array, voxelrcnn_head.py:206-215)."""
    g = torch.arange(grid_size, device=rois.device, dtype=rois.dtype)
    idx = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1).reshape(1, -1, 3)
    size = rois[:, None, 3:6]
    return (idx + 0.5) / grid_size * size - size / 2


def global_grid_points_of_roi(rois, grid_size):
    rois = rois.reshape(-1, rois.shape[-1])
    local = dense_grid_points(rois, grid_size)
    glob = rotate_points_along_z(local, rois[:, 6]) + rois[:, None, 0:3]
    return glob, local


class RoIGridPool(nn.Module):
    """Multi-scale RoI-grid pooling.  pool_cfg: {src_name: dict(mlps, query_ranges, radii, nsamples)}
    in FEATURES_SOURCE order; backbone_channels: {src_name: C}."""

    def __init__(self, backbone_channels, pool_cfg, grid_size, voxel_size, point_cloud_range):
        super().__init__()
        self.grid_size, self.voxel_size, self.point_cloud_range = grid_size, voxel_size, point_cloud_range
        self.sources = list(pool_cfg.keys())
        self.roi_grid_pool_layers = nn.ModuleList()
        self.num_features = 0
        for name in self.sources:
            c = pool_cfg[name]
            mlps = [[backbone_channels[name]] + list(m) for m in c["mlps"]]
            self.roi_grid_pool_layers.append(voxel_pool_modules.NeighborVoxelSAModuleMSG(
                query_ranges=c["query_ranges"], nsamples=c["nsamples"], radii=c["radii"], mlps=mlps,
                pool_method=c.get("pool_method", "max_pool")))
            self.num_features += sum(m[-1] for m in mlps)

    def grid_coords(self, roi_grid_xyz):
        """Voxel coordinates of grid points at stride 1: float floor division on f32, as the
        reference does (`//` on tensors, voxelrcnn_head.py:130-134)."""
        r, v = self.point_cloud_range, self.voxel_size
        return torch.stack([(roi_grid_xyz[..., i] - r[i]) // v[i] for i in range(3)], dim=-1)

    def forward(self, rois, multi_scale_3d_features, multi_scale_3d_strides, batch_size):
        """rois (B, R, 7+) -> (B*R, G^3, sum C_out)."""
        B = batch_size
        if self._fusable(rois, multi_scale_3d_features):
            return self._forward_fused(rois, multi_scale_3d_features, multi_scale_3d_strides, B)
        if self._trainable_rows(rois, multi_scale_3d_features):
            return self._forward_rows(rois, multi_scale_3d_features, multi_scale_3d_strides, B)
        grid_xyz, _ = global_grid_points_of_roi(rois, self.grid_size)           # (B*R, G^3, 3)
        grid_xyz = grid_xyz.reshape(B, -1, 3)
        coords1 = self.grid_coords(grid_xyz)                                      # (B, R*G^3, 3) float
        m = grid_xyz.shape[1]
        bcol = torch.arange(B, device=rois.device, dtype=coords1.dtype).view(B, 1, 1).expand(B, m, 1)
        new_cnt = torch.full((B,), m, dtype=torch.int32, device=rois.device)
        pooled = []
        for layer, name in zip(self.roi_grid_pool_layers, self.sources):
            st = multi_scale_3d_features[name]
            stride = multi_scale_3d_strides[name]
            n = st.indices.shape[0] if st.count is None else int(st.count.item())
            ind, feats = st.indices[:n], st.features[:n]
            xyz = get_voxel_centers(ind[:, 1:4], stride, self.voxel_size, self.point_cloud_range)
            xyz_cnt = torch.bincount(ind[:, 0].long(), minlength=B).int()
            coords = torch.cat([bcol, coords1 // stride], dim=-1).int()              # [b, x, y, z]
            out = layer(xyz=xyz.contiguous(), xyz_batch_cnt=xyz_cnt,
                        new_xyz=grid_xyz.reshape(-1, 3).contiguous(), new_xyz_batch_cnt=new_cnt,
                        new_coords=coords.reshape(-1, 4).contiguous(), features=feats.contiguous(),
                        voxel2point_indices=st)
            pooled.append(out.view(-1, self.grid_size ** 3, out.shape[-1]))
        return torch.cat(pooled, dim=-1)

    # ---- training path (autograd), free of host synchronisation: what a shape-static training step
    # (glenet_amd.glenet_vr.StaticTrainStep) records into its HIP graph.  Same arithmetic as the generic
    # path above (NeighborVoxelSAModuleMSG._forward_rows); the differences are the grid-point kernel, the
    # query through the cell index with centres rebuilt from the indices (no xyz / count / coordinate
    # tensors, no bincount) and the live-row handling of mlps_in: a shape-static sparse tensor carries
    # undefined rows past `count`, so the statistics of its BatchNorm run over the live rows only
    # (fused kernels, csrc/glx_bn.hip) and the rows past them are selected away (torch.where, not a
    # multiplication: they may hold NaN) on both sides of the 1x1 conv, forward and backward.
    USE_ROWS = True
    USE_POS_POOL = True

    def _trainable_rows(self, rois, tensors):
        if not (self.USE_ROWS and rois.is_cuda and rois.dtype == torch.float32):
            return False
        for layer, name in zip(self.roi_grid_pool_layers, self.sources):
            st = tensors[name]
            if layer.pool_method != "max_pool" or (st.count is not None and st._index is None):
                return False
        return True

    @staticmethod
    def _mlp_in_rows(layer, seq, st):
        from .spconv import core
        conv, bn = seq[0], seq[1]
        w = conv.weight.reshape(conv.out_channels, conv.in_channels)
        x = st.features
        live = None
        # rows past `count`: a tensor that left the fused BatchNorm kernels carries zeros there (and gets zero
        # gradients back, see glx_bn_relu_*), anything else may hold NaN and is masked on both sides of the conv
        clean = st.count is None or getattr(st, "clean_rows", False)
        if clean and voxel_pool_modules.rows_conv_bn_supported(seq, x):
            # the product with the BatchNorm statistics in its epilogue, the BatchNorm backward applied on load (csrc/glx_rows.hip)
            return voxel_pool_modules.rows_conv_bn(seq, x, st.count)
        if st.count is not None and not getattr(st, "clean_rows", False):
            live = (torch.arange(x.shape[0], device=x.device) < st.count).view(-1, 1)
            x = torch.where(live, x, x.new_zeros(()))
        y = layer._linear_rows(x, w, conv.bias)
        if core.can_fuse_train_bn(bn, y):
            if live is not None:
                y = torch.where(live, y, y.new_zeros(()))        # its backward cleans the gradient rows
            y = core.fused_train_bn(bn, y, False, st.count)
            return F.relu(y) if len(seq) > 2 else y
        if st.count is not None and bn.training:
            raise NotImplementedError("shape-static training needs the fused BatchNorm kernels")
        return layer._bn_rows(seq, y)

    def _forward_rows(self, rois, tensors, strides, B):
        import ctypes
        from . import _lib
        GroupRows, ReluAddMax = voxel_pool_modules.GroupRows, voxel_pool_modules.ReluAddMax
        f3 = ctypes.c_float * 3
        rmin, vsz = f3(*self.point_cloud_range[0:3]), f3(*self.voxel_size)
        rois2 = rois.detach().reshape(-1, rois.shape[-1]).contiguous()
        n, g3 = rois2.shape[0], self.grid_size ** 3
        m = n * g3
        dev = rois.device
        grid_xyz = torch.empty((m, 3), dtype=torch.float32, device=dev)
        coords = torch.empty((m, 4), dtype=torch.int32, device=dev)
        _lib.call("glx_roi_grid_points", rois2, n, rois2.shape[1], n // B, self.grid_size, rmin, vsz,
                  grid_xyz, coords)
        outs = []
        # the scales' neighbour queries need the grid points and the tensors' cell indices only: ONE launch for all of them
        # (blockIdx.y = scale) instead of a 43-46 us launch per scale down the chain
        queries = {}
        if self.GROUPED_QUERY:
            qs = []
            for k, (layer, name) in enumerate(zip(self.roi_grid_pool_layers, self.sources)):
                st = tensors[name]
                index = st._ensure_index()
                z, y, x = st.spatial_shape
                ind = st.indices.contiguous()
                for j, grouper in enumerate(layer.groupers):
                    zr, yr, xr = grouper.max_range
                    idx = torch.empty((m, grouper.nsample), dtype=torch.int32, device=dev)
                    qs.append((k, j, idx, ind, _lib.RoiQuery(z, y, x, grouper.nsample, zr, yr, xr, int(strides[name]),
                                                             float(grouper.radius), ind.data_ptr(), index.bitmap.data_ptr(),
                                                             index.prefix.data_ptr(), _lib._p(index.rank_to_row), idx.data_ptr())))
            if 1 <= len(qs) <= 4 and all(2 * q[4].x_range + 1 <= 32 for q in qs):
                arr = (_lib.RoiQuery * len(qs))(*[q[4] for q in qs])
                _lib.call("glx_roi_grid_query_multi", len(qs), arr, m, grid_xyz, coords, rmin, vsz)
                queries = {(q[0], q[1]): q[2] for q in qs}
        for k, (layer, name) in enumerate(zip(self.roi_grid_pool_layers, self.sources)):
            st = tensors[name]
            index = st._ensure_index()
            z, y, x = st.spatial_shape
            ind = st.indices.contiguous()
            stride = int(strides[name])
            xyz = torch.empty((ind.shape[0], 3), dtype=torch.float32, device=dev)           # get_voxel_centers
            _lib.call("glx_voxel_centers", ind, ind.shape[0], stride, rmin, vsz, xyz)
            for j, (grouper, mlp_in, mlp_pos, mlp_out) in enumerate(zip(layer.groupers, layer.mlps_in, layer.mlps_pos,
                                                                        layer.mlps_out)):
                feats = self._mlp_in_rows(layer, mlp_in, st)   # (N, c_mid)
                ns = grouper.nsample
                idx = queries.get((k, j))
                if idx is None:
                    idx = torch.empty((m, ns), dtype=torch.int32, device=dev)
                    zr, yr, xr = grouper.max_range
                    _lib.call("glx_roi_grid_query", m, z, y, x, ns, float(grouper.radius), zr, yr, xr, grid_xyz,
                              coords, stride, ind, rmin, vsz, index.bitmap, index.prefix, index.rank_to_row, idx)
                if self.USE_POS_POOL and voxel_pool_modules.pos_pool_out_supported(feats, mlp_pos, mlp_out):
                    # ... and the output MLP's convolution + BatchNorm statistics in the same launch
                    outs.append(voxel_pool_modules.pos_pool_out(feats, mlp_pos, mlp_out, idx, xyz, grid_xyz))
                    continue
                if self.USE_POS_POOL and voxel_pool_modules.pos_pool_supported(feats, mlp_pos):
                    # position MLP + add + ReLU + max-pool fused: no (M, ns, C) tensor (csrc/glx_roipool.hip)
                    pooled = voxel_pool_modules.pos_pool(feats, mlp_pos, idx, xyz, grid_xyz)
                else:
                    g_feat = GroupRows.apply(feats, idx)                           # (M, ns, c_mid)
                    with torch.no_grad():
                        keep = (idx[:, :1] >= 0).to(feats.dtype).view(m, 1, 1)
                        rel = (GroupRows.apply(xyz, idx) - grid_xyz.view(m, 1, 3)) * keep
                    pos = layer._conv_bn_rows(mlp_pos, rel.view(m * ns, 3))       # (M*ns, c_mid)
                    pooled = ReluAddMax.apply(g_feat, pos.view(m, ns, -1))        # (M, c_mid)
                outs.append(layer._conv_bn_rows(mlp_out, pooled))                 # (M, c_out)
        return torch.cat(outs, dim=1).view(n, g3, -1)

    # the scales' voxel queries in one launch: built, bit-identical, measured on the step 6.145 / 6.094 ms against 6.086 /
    # 6.078 per scale (the three launches already overlap the towers of the previous scale): off
    GROUPED_QUERY = False

    # ---- inference fast path: 1 + 3 launches per scale (csrc/glx_points.hip) -- grid points and
    # their voxel coordinates in one kernel, then per scale mlp_in (one GEMM), the voxel query and
    # the fused aggregation, both rebuilding voxel centres from the sparse tensor's indices and
    # writing straight into the concatenated output.  No centres, counts, coordinate or grouped
    # tensors, no host synchronisation.
    USE_FUSED = True

    def _fusable(self, rois, tensors):
        if not (self.USE_FUSED and not self.training and not torch.is_grad_enabled() and rois.is_cuda
                and rois.dtype == torch.float32):
            return False
        for layer, name in zip(self.roi_grid_pool_layers, self.sources):
            st = tensors[name]
            if not layer._fusable(st.features) or (st.count is not None and st._index is None):
                return False
        return True

    def _forward_fused(self, rois, tensors, strides, B):
        import ctypes
        from . import _lib
        f3 = ctypes.c_float * 3
        rmin, vsz = f3(*self.point_cloud_range[0:3]), f3(*self.voxel_size)
        rois2 = rois.reshape(-1, rois.shape[-1]).contiguous()
        n, g3 = rois2.shape[0], self.grid_size ** 3
        m = n * g3
        dev = rois.device
        grid_xyz = torch.empty((m, 3), dtype=torch.float32, device=dev)
        coords = torch.empty((m, 4), dtype=torch.int32, device=dev)
        _lib.call("glx_roi_grid_points", rois2, n, rois2.shape[1], n // B, self.grid_size, rmin, vsz,
                  grid_xyz, coords)
        out = torch.empty((m, self.num_features), dtype=torch.float32, device=dev)
        col = 0
        for layer, name in zip(self.roi_grid_pool_layers, self.sources):
            st = tensors[name]
            index = st._ensure_index()
            z, y, x = st.spatial_shape
            ind, feats_src = st.indices.contiguous(), st.features.contiguous()
            _lib.check_cuda(ind, feats_src)
            for grouper, ((w_in, b_in), (w_pos, b_pos), (w_out, b_out)) in zip(layer.groupers,
                                                                               layer._folded()):
                feats = torch.addmm(b_in, feats_src, w_in.t())
                idx = torch.empty((m, grouper.nsample), dtype=torch.int32, device=dev)
                zr, yr, xr = grouper.max_range
                _lib.call("glx_roi_grid_query", m, z, y, x, grouper.nsample, float(grouper.radius), zr,
                          yr, xr, grid_xyz, coords, int(strides[name]), ind, rmin, vsz, index.bitmap,
                          index.prefix, index.rank_to_row, idx)
                co, cm = w_out.shape
                _lib.call("glx_roi_grid_agg", feats, ind, int(strides[name]), rmin, vsz, grid_xyz, idx,
                          m, grouper.nsample, cm, co, w_pos, b_pos, w_out, b_out, out[:, col:],
                          self.num_features)
                col += co
        return out.view(n, g3, self.num_features)
