"""Drop-in for `pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda`
(pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:172-177)."""
import torch

from ... import _lib
from ..._lib import call


def forward(rois, pts, pts_feature, argmax, pts_idx_of_voxels, pooled_features, pool_method):
    _lib.check_cuda(rois, pts, pts_feature, argmax, pts_idx_of_voxels, pooled_features)
    n, ox, oy, oz, c = pooled_features.shape
    call("glx_roiaware_pool3d_forward", rois, n, pts, pts.shape[0], pts_feature, c, ox, oy, oz,
         pts_idx_of_voxels.shape[4], int(pool_method), argmax, pts_idx_of_voxels, pooled_features)
    return 1


def backward(pts_idx_of_voxels, argmax, grad_out, grad_in, pool_method):
    _lib.check_cuda(pts_idx_of_voxels, argmax, grad_out, grad_in)
    n, ox, oy, oz, maxpts = pts_idx_of_voxels.shape
    call("glx_roiaware_pool3d_backward", pts_idx_of_voxels, argmax, grad_out, n, ox, oy, oz,
         grad_out.shape[4], maxpts, int(pool_method), grad_in)
    return 1


def points_in_boxes_gpu(boxes, pts, box_idx_of_points):
    _lib.check_cuda(boxes, pts, box_idx_of_points)
    b, t, _ = boxes.shape
    call("glx_points_in_boxes", boxes, pts, b, t, pts.shape[1], box_idx_of_points)
    return 1


def points_in_boxes_cpu(boxes, pts, pts_indices):
    """roiaware_pool3d.cpp:143-168: host tensors in and out, host arithmetic (libglenet_host.so; MARGIN 1e-2),
    no GPU runtime call -- callable from forked DataLoader workers (box_utils.py:86, augmentor_utils.py)."""
    from ... import _host
    if boxes.is_cuda or pts.is_cuda or pts_indices.is_cuda:
        raise _lib.GlxError("points_in_boxes_cpu takes host tensors (use points_in_boxes_gpu for device tensors)")
    out = _host.points_in_boxes(boxes.detach().float().contiguous().numpy(), pts.detach().float().contiguous().numpy())
    pts_indices.copy_(torch.from_numpy(out).view_as(pts_indices))
    return 1
