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
    """Host tensors in/out like the reference's CPU routine (MARGIN 1e-2); computed on the GPU,
    so not for forked DataLoader workers."""
    dev = torch.device("cuda", torch.cuda.current_device())
    b, p = boxes.float().contiguous().to(dev), pts.float().contiguous().to(dev)
    out = torch.zeros((b.shape[0], p.shape[0]), dtype=torch.int32, device=dev)
    call("glx_points_in_boxes_mask", b, b.shape[0], p, p.shape[0], 1e-2, out)
    pts_indices.copy_(out.cpu())
    return 1
