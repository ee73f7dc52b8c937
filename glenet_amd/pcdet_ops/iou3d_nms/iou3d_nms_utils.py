"""Mirror of pcdet/ops/iou3d_nms/iou3d_nms_utils.py: same names, argument order, return arity
(every NMS entry swallows **kwargs because the caller passes the whole nms_config,
pcdet/models/model_utils/model_nms_utils.py:41-52)."""
import numpy as np
import torch

from ... import _lib
from ..._lib import call
from .._boxgeom import iou3d_from_bev
from . import iou3d_nms_cuda


def _to_torch(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def limit_period(val, offset=0.5, period=np.pi):
    """pcdet/utils/common_utils.py:21-24."""
    val, is_numpy = _to_torch(val)
    ans = val - torch.floor(val / period + offset) * period
    return ans.numpy() if is_numpy else ans


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:52-68 (host tensors / ndarrays in and out)."""
    boxes_a, is_numpy = _to_torch(boxes_a)
    boxes_b, _ = _to_torch(boxes_b)
    assert not (boxes_a.is_cuda or boxes_b.is_cuda), 'Only support CPU tensors'
    assert boxes_a.shape[1] == 7 and boxes_b.shape[1] == 7
    ans_iou = boxes_a.new_zeros(torch.Size((boxes_a.shape[0], boxes_b.shape[0])))
    iou3d_nms_cuda.boxes_iou_bev_cpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou.numpy() if is_numpy else ans_iou


def boxes_iou_bev(boxes_a, boxes_b):
    """iou3d_nms_utils.py:71-85."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    ans_iou = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7) x (M,7) -> (N,M) 3-D IoU: BEV overlap area from the kernel, then height overlap and
    union volume clamped at 1e-6 (iou3d_nms_utils.py:88-121)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    bev = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_overlap_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), bev)
    return iou3d_from_bev(bev, boxes_a, boxes_b, pairwise=True, eps=1e-6)


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """iou3d_nms_utils.py:182-197 -> (selected indices into `boxes`, None).  The suppression
    matrix never leaves the device; one 4-byte read-back sizes the result."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous()
    keep, num = iou3d_nms_cuda.nms_device(boxes, thresh, normal=False)
    return order[keep[:int(num.item())]].contiguous(), None


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """iou3d_nms_utils.py:276-290."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    boxes = boxes[order].contiguous()
    keep, num = iou3d_nms_cuda.nms_device(boxes, thresh, normal=True)
    return order[keep[:int(num.item())]].contiguous(), None


def nms_func_device(boxes, scores, iou_threshold, score_threshold=0, variance=None):
    """nms_func (iou3d_nms_utils.py:227-273) on device tensors; returns (scores, boxes) copies."""
    boxes = boxes.float().contiguous().clone()
    scores = scores.float().contiguous().clone()
    _lib.check_cuda(boxes, scores)
    n = boxes.shape[0]
    ious_t = boxes_iou_bev(boxes, boxes).t().contiguous()      # the voting block reads rows of the transpose
    var, stride = None, 0
    if variance is not None:
        var = variance.float().contiguous()
        stride = var.shape[1]
    scratch = torch.empty((n, 8), dtype=torch.float32, device=boxes.device) if var is not None else None
    call("glx_nms_vote", boxes, scores, var, stride, ious_t, n, float(iou_threshold), float(score_threshold),
         scratch)
    return scores, boxes


def new_nms_gpu(boxes, scores, iou_threshold, pre_maxsize=None, score_threshold=0, variance=None, **kwargs):
    """iou3d_nms_utils.py:200-224: returns (keep ndarray, None, new_boxes ndarray) like the
    reference, which runs this on the host in numpy; here the IoU matrix and the voting loop run
    on the device and only the results are copied back."""
    boxes = boxes.float().clone()
    boxes[:, 6] = limit_period(boxes[:, 6], offset=0.5, period=np.pi * 2)
    new_scores, new_boxes = nms_func_device(boxes, scores, iou_threshold, score_threshold, variance)
    new_scores, new_boxes = new_scores.cpu().numpy(), new_boxes.cpu().numpy()
    keep = (new_scores > 0).nonzero()[0]
    keep = keep[new_scores[keep].argsort()[::-1]]
    return keep, None, new_boxes
