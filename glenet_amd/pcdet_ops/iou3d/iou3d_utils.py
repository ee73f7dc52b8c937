"""Mirror of the reachable part of pcdet/ops/iou3d/iou3d_utils.py: the pairwise-aligned 3-D IoU
used by the IoU-aware anchor heads (anchor_head_kl_label.py:255,428)."""
import torch

from .._boxgeom import iou3d_from_bev
from . import iou3d_cuda


def boxes3d_to_bev_torch(boxes3d, box_mode='wlh', rect=False):
    """(N,7|5) centre boxes -> (N,5) [x1, y1, x2, y2, ry] (iou3d_utils.py:79-106).  box_mode
    names the order of the three size columns; in LiDAR frame x spans w and y spans l, in the
    rectified camera frame (rect) the plane is (x, z) and x spans l."""
    first = {5: 2, 7: 3}.get(boxes3d.shape[-1])
    if first is None:
        raise NotImplementedError
    half_w = boxes3d[:, box_mode.index('w') + first] / 2.
    half_l = boxes3d[:, box_mode.index('l') + first] / 2.
    if rect:
        u, v, du, dv = boxes3d[:, 0], boxes3d[:, 2], half_l, half_w
    else:
        u, v, du, dv = boxes3d[:, 0], boxes3d[:, 1], half_w, half_l
    return torch.stack([u - du, v - dv, u + du, v + dv, boxes3d[:, -1]], dim=1)


def boxes_aligned_iou3d_gpu(boxes_a, boxes_b, box_mode='wlh', rect=False, need_bev=False):
    """iou_3d[i] of boxes_a[i] with boxes_b[i], shape (N,1) (iou3d_utils.py:332-387); unions are
    clamped at 1e-7."""
    assert boxes_a.shape[0] == boxes_b.shape[0]
    if rect:
        raise NotImplementedError
    w_col, l_col, h_col = (box_mode.index(ch) + 3 for ch in 'wlh')
    bev = torch.zeros((boxes_a.shape[0], 1), dtype=torch.float32, device=boxes_a.device)
    iou3d_cuda.boxes_aligned_overlap_bev_gpu(boxes3d_to_bev_torch(boxes_a, box_mode, rect).contiguous(),
                                             boxes3d_to_bev_torch(boxes_b, box_mode, rect).contiguous(), bev)
    iou3d = iou3d_from_bev(bev, boxes_a, boxes_b, pairwise=False, eps=1e-7, h_col=h_col)
    if not need_bev:
        return iou3d
    area_a = (boxes_a[:, w_col] * boxes_a[:, l_col]).view(-1, 1)
    area_b = (boxes_b[:, w_col] * boxes_b[:, l_col]).view(-1, 1)
    return iou3d, bev / (area_a + area_b - bev).clamp(min=1e-7)
