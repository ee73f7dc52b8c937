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


def boxes3d_to_bev_3d_torch(boxes3d, box_mode='wlh', rect=False):
    """(N,7) centre boxes -> (N,7) [x1, y1, z1, x2, y2, z2, ry] (iou3d_utils.py:109-131); LiDAR frame: z is the
    box centre; rectified camera frame: the plane is (x, z) and y is the box bottom."""
    w_col, l_col, h_col = (box_mode.index(ch) + 3 for ch in 'wlh')
    half_w, half_l, height = boxes3d[:, w_col] / 2., boxes3d[:, l_col] / 2., boxes3d[:, h_col]
    if rect:
        cu, cv, cw = boxes3d[:, 0], boxes3d[:, 2], boxes3d[:, 1]
        lo = [cu - half_l, cv - half_w, cw - height]
        hi = [cu + half_l, cv + half_w, cw]
    else:
        cu, cv, cw = boxes3d[:, 0], boxes3d[:, 1], boxes3d[:, 2]
        lo = [cu - half_w, cv - half_l, cw - height / 2.]
        hi = [cu + half_w, cv + half_l, cw + height / 2.]
    return torch.stack(lo + hi + [boxes3d[:, 6]], dim=1)


def _nms(entry, boxes, scores, thresh):
    order = scores.sort(0, descending=True)[1]
    keep = torch.zeros(boxes.shape[0], dtype=torch.int64)             # host tensor, as iou3d_utils.py:402 builds it
    n = entry(boxes[order].contiguous(), keep, thresh)
    return order[keep[:n].to(order.device)].contiguous()


def nms_gpu(boxes, scores, thresh, box_mode='wlh'):
    """Rotated-BEV NMS on (N,7) centre boxes (iou3d_utils.py:389-406; the reference converts with rect=True)."""
    return _nms(iou3d_cuda.nms_gpu, boxes3d_to_bev_torch(boxes, box_mode, rect=True), scores, thresh)


def nms_3d_gpu(boxes, scores, thresh, box_mode='wlh'):
    """3-D IoU NMS on (N,7) centre boxes (iou3d_utils.py:408-424)."""
    return _nms(iou3d_cuda.nms_3d_gpu, boxes3d_to_bev_3d_torch(boxes, box_mode, rect=False), scores, thresh)


def nms_normal_gpu(boxes, scores, thresh):
    """Axis-aligned NMS on (N,5) [x1,y1,x2,y2,ry] (iou3d_utils.py:426-442)."""
    return _nms(iou3d_cuda.nms_normal_gpu, boxes, scores, thresh)
