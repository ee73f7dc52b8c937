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


def softnms(boxes, scores, iou_threshold, soft_sigma, score_threshold, soft_mode="gaussian", variance=None):
    """Soft-NMS with optional variance voting (iou3d_nms_utils.py:313-356), updating `boxes` and `scores` in place
    as the reference does.  The reference recomputes one IoU column per iteration on the device and reads the
    arg-max back each time; every IoU it forms is between ORIGINAL boxes (a box is rewritten by the vote only when
    it leaves the candidate set), so here the (N, N) matrix is computed once on the device and the inherently
    sequential sweep runs on the host in float32."""
    assert soft_mode in ("linear", "gaussian")
    n = boxes.shape[0]
    if n == 0:
        return scores, boxes
    iou = boxes_iou_bev(boxes[:, :7].float().contiguous(), boxes[:, :7].float().contiguous()).cpu().numpy()
    b = boxes.detach().float().cpu().numpy().copy()
    s = scores.detach().float().cpu().numpy().copy()
    var = variance.detach().float().cpu().numpy() if variance is not None else None
    orig = b.copy()
    undone = s >= np.float32(score_threshold)
    while undone.sum() > 1:
        cand = undone.nonzero()[0]
        idx = int(cand[s[cand].argmax()])
        undone[idx] = False
        others = undone.nonzero()[0]
        ious = iou[others, idx]
        if var is not None:
            m = ious > np.float32(iou_threshold)
            klbox = np.concatenate([orig[others[m], :6], orig[idx:idx + 1, :6]], 0)
            klvar = np.concatenate([var[others[m], :6], var[idx:idx + 1, :6]], 0)
            w = np.exp(np.float32(-1.0) * (np.float32(1.0) - ious[m]) ** 2 / np.float32(0.05)).astype(np.float32)
            w = np.concatenate([w, np.ones(1, np.float32)])[:, None] / klvar
            w = w / w.sum(0)
            b[idx, :6] = (w * klbox).sum(0)
        if soft_mode == "linear":
            scale = np.where(ious >= np.float32(soft_sigma), np.float32(1.0) - ious, np.float32(1.0)).astype(np.float32)
        else:
            scale = np.exp(-ious ** 2 / np.float32(soft_sigma)).astype(np.float32)
        s[others] *= scale
        undone[s < np.float32(score_threshold)] = False
    boxes.copy_(torch.from_numpy(b).to(boxes.device, boxes.dtype))
    scores.copy_(torch.from_numpy(s).to(scores.device, scores.dtype))
    return scores, boxes


def softnms_gpu(boxes, scores, iou_threshold, score_threshold=0.1, soft_mode='gaussian', variance=None,
                soft_sigma=0.3, **kwargs):
    """iou3d_nms_utils.py:292-302 -> (keep, None, new_boxes): indices whose decayed score stays above the
    threshold, by descending score."""
    assert soft_mode in ("linear", "gaussian")
    assert boxes.shape[-1] == 7
    new_scores, new_boxes = softnms(boxes, scores, iou_threshold, soft_sigma, score_threshold, soft_mode,
                                    variance=variance)
    keep = (new_scores > score_threshold).nonzero(as_tuple=False).view(-1)
    keep = keep[new_scores[keep].argsort(descending=True)]
    return keep, None, new_boxes
