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
