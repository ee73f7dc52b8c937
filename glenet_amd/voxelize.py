"""Device voxelizers (torch tensors in/out, all arithmetic in libglenet_hip.so).

hard_voxelize      semantics of VoxelGeneratorWrapper.generate
                   (pcdet/datasets/processor/data_processor.py:15-60, spconv generators)
dynamic_voxelize_mean  semantics of DynamicMeanVFE.forward
                   (pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:37-76)
mean_vfe           MeanVFE.forward (pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31)
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import call, query, size_arg, workspace


def grid_size_of(point_cloud_range, voxel_size):
    """round((max - min) / voxel) as data_processor.py:119-120 does (numpy float64 of the
    python lists)."""
    r = np.asarray(point_cloud_range, dtype=np.float64)
    v = np.asarray(voxel_size, dtype=np.float64)
    return [int(g) for g in np.round((r[3:6] - r[0:3]) / v).astype(np.int64)]


def _f32arr(vals):
    a = np.asarray(vals, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _split_batch(points, batch_idx, batch_size):
    if batch_idx is None:
        return None, 1 if batch_size is None else batch_size
    bi = batch_idx.int().contiguous()
    return bi, int(batch_size)


def hard_voxelize(points, voxel_size, point_cloud_range, max_points, max_voxels, batch_idx=None,
                  batch_size=None, index_depth=None, static=False):
    """points (P, C) float32 device tensor, xyz first; batch_idx (P,) frame ids (stacked frames,
    non-decreasing) or None for one frame.

    Returns voxels (Nv, max_points, C), coords (Nv, 4) int32 [b, z, y, x], num_points (Nv,) int32,
    voxel_offset (B+1,) int32 (rows of frame b are voxel_offset[b]:voxel_offset[b+1]).
    With index_depth (gz or gz+1) a fifth value is returned: the cell index (glenet_amd.spconv
    CellIndex) of these voxels on the grid (B, index_depth, gy, gx), or None when max_voxels
    dropped cells (then the sparse tensor builds its own).
    static=True (needs index_depth): no host synchronisation -- all B*max_voxels rows are
    returned, the live row count stays on the device as index.count (== voxel_offset[B]); rows
    beyond it are undefined.  spconv.core.check_static() validates the frame afterwards.
    """
    points = points.contiguous().float()
    _lib.check_cuda(points)
    dev = points.device
    P, C = points.shape
    bi, B = _split_batch(points, batch_idx, batch_size)
    gx, gy, gz = grid_size_of(point_cloud_range, voxel_size)
    rng, rng_p = _f32arr(point_cloud_range)
    vs, vs_p = _f32arr(voxel_size)
    cap = B * max_voxels
    voxels = torch.empty((cap, max_points, C), dtype=torch.float32, device=dev)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num = torch.empty((cap,), dtype=torch.int32, device=dev)
    offs = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    wsb = query("glx_voxelize_hard_workspace_bytes", P, B, gx, gy, gz, max_points, max_voxels)
    ws = workspace.get(wsb, dev)
    if index_depth is None:
        call("glx_voxelize_hard", points, bi, P, C, B, rng_p, vs_p, gx, gy, gz, max_points,
             max_voxels, voxels, coords, num, offs, 0, None, None, None, None, None, ws,
             size_arg(ws.numel()))
        nv = offs.tolist()[-1]  # host sync: voxel count sizes the outputs
        return voxels[:nv], coords[:nv], num[:nv], offs
    from .spconv.core import CellIndex
    grid = (B, int(index_depth), gy, gx)
    bitmap, flags, prefix = CellIndex.alloc(grid, dev)
    r2row = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
    meta = torch.empty(B + 2, dtype=torch.int32, device=dev)      # voxel_offset (B+1), n_unique
    call("glx_voxelize_hard", points, bi, P, C, B, rng_p, vs_p, gx, gy, gz, max_points, max_voxels,
         voxels, coords, num, meta[:B + 1], int(index_depth), bitmap, flags, prefix, r2row,
         meta[B + 1:], ws, size_arg(ws.numel()))
    if static:
        index = CellIndex(grid, bitmap, flags, prefix, r2row, None, cap, count=meta[B:B + 1],
                          unique=meta[B + 1:B + 2])
        return voxels, coords, num, meta[:B + 1], index
    meta_h = meta.tolist()  # the one host sync: voxel count + unique-cell count
    nv, n_unique = meta_h[B], meta_h[B + 1]
    index = CellIndex(grid, bitmap, flags, prefix, r2row[:nv], None, nv) if n_unique == nv else None
    return voxels[:nv], coords[:nv], num[:nv], meta[:B + 1], index


def dynamic_voxelize_mean(points, voxel_size, point_cloud_range, batch_idx=None, batch_size=None):
    """Returns features (Nv, C) = per-voxel mean of all point columns, coords (Nv,4) [b,z,y,x],
    ordered by ascending key b*XYZ + x*YZ + y*Z + z (torch.unique order of the reference)."""
    points = points.contiguous().float()
    _lib.check_cuda(points)
    dev = points.device
    P, C = points.shape
    bi, B = _split_batch(points, batch_idx, batch_size)
    gx, gy, gz = grid_size_of(point_cloud_range, voxel_size)
    rng, rng_p = _f32arr(point_cloud_range)
    vs, vs_p = _f32arr(voxel_size)
    feats = torch.empty((max(P, 1), C), dtype=torch.float32, device=dev)
    coords = torch.empty((max(P, 1), 4), dtype=torch.int32, device=dev)
    nv_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = query("glx_voxelize_dynamic_workspace_bytes", P, B, gx, gy, gz)
    ws = workspace.get(wsb, dev)
    call("glx_voxelize_dynamic_mean", points, bi, P, C, B, rng_p, vs_p, gx, gy, gz, feats, coords,
         nv_dev, ws, size_arg(ws.numel()))
    nv = int(nv_dev.item())
    return feats[:nv], coords[:nv]


def mean_vfe(voxels, num_points, count=None):
    """count: device int32[1] live rows (shape-static mode), rows beyond it are left undefined."""
    voxels = voxels.contiguous().float()
    num_points = num_points.int().contiguous()
    _lib.check_cuda(voxels, num_points)
    nv, mp, c = voxels.shape
    out = torch.empty((nv, c), dtype=torch.float32, device=voxels.device)
    call("glx_mean_vfe", voxels, num_points, nv, mp, c, out, count)
    return out
