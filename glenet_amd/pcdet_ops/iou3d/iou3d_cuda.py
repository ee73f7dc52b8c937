"""Drop-in for `pcdet.ops.iou3d.iou3d_cuda` (pcdet/ops/iou3d/src/iou3d.cpp:270-281): the BEV
entry points on [x1,y1,x2,y2,ry] boxes.  Only boxes_aligned_overlap_bev_gpu is reached by
GLENet (AnchorHeadKLLabelIoU); overlap / IoU share the kernel."""
import torch

from ... import _lib
from ..._lib import call


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    _lib.check_cuda(boxes_a, boxes_b, ans_overlap)
    call("glx_iou3d_boxes_overlap_bev", boxes_a, boxes_a.shape[0], boxes_b, boxes_b.shape[0], 0, ans_overlap)
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    _lib.check_cuda(boxes_a, boxes_b, ans_iou)
    call("glx_iou3d_boxes_overlap_bev", boxes_a, boxes_a.shape[0], boxes_b, boxes_b.shape[0], 1, ans_iou)
    return 1


def boxes_aligned_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    _lib.check_cuda(boxes_a, boxes_b, ans_overlap)
    assert boxes_a.shape[0] == boxes_b.shape[0]
    call("glx_iou3d_boxes_aligned_overlap_bev", boxes_a, boxes_b, boxes_a.shape[0], ans_overlap)
    return 1


def _host_bev(boxes_a, boxes_b, ans, iou):
    from ... import _host
    if boxes_a.is_cuda or boxes_b.is_cuda or ans.is_cuda:
        raise _lib.GlxError("the *_cpu entry points take host tensors")
    out = _host.iou3d_boxes_bev(boxes_a.detach().float().contiguous().numpy(),
                                boxes_b.detach().float().contiguous().numpy(), iou=iou)
    ans.copy_(torch.from_numpy(out).view_as(ans))
    return 1


def boxes_overlap_bev_cpu(boxes_a, boxes_b, ans_overlap):
    """iou3d_cpu.cpp:232-256 of the iou3d library: host tensors, host arithmetic (libglenet_host.so)."""
    return _host_bev(boxes_a, boxes_b, ans_overlap, False)


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """iou3d_cpu.cpp:259-282 of the iou3d library."""
    return _host_bev(boxes_a, boxes_b, ans_iou, True)
