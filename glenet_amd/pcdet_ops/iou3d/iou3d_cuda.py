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


# ------------------------------------------------------------------ the rest of iou3d.cpp:270-281
# Dead code for every GLENet configuration, kept so that `pcdet.ops.iou3d.iou3d_cuda` is complete.
# The pairwise quantities come from the pinned device kernel (glx_iou3d_boxes_overlap_bev, the library's rotated
# rectangle overlap on [x1,y1,x2,y2,ry]); the greedy sweep over the suppression matrix runs on the host, as the
# reference's own entry points do (iou3d.cpp:120-266: mask copied to the host, sequential loop there), and `keep`
# is the HOST int64 tensor the callers pass (iou3d_utils.py:402-404).
_EPS = 1e-8                                                          # iou3d_kernel.cu:13


def _iou3d_matrix(boxes_a, boxes_b):
    """iou_3d of iou3d_kernel.cu:256-268 for (N,7) x (M,7) [x1,y1,z1,x2,y2,z2,ry]: BEV overlap of the
    (x1,y1,x2,y2,ry) rectangles x height overlap, over the union of the volumes."""
    cols = [0, 1, 3, 4, 6]
    bev = torch.empty((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    call("glx_iou3d_boxes_overlap_bev", boxes_a[:, cols].contiguous(), boxes_a.shape[0],
         boxes_b[:, cols].contiguous(), boxes_b.shape[0], 0, bev)
    va = ((boxes_a[:, 3] - boxes_a[:, 0]) * (boxes_a[:, 4] - boxes_a[:, 1]) * (boxes_a[:, 5] - boxes_a[:, 2]))[:, None]
    vb = ((boxes_b[:, 3] - boxes_b[:, 0]) * (boxes_b[:, 4] - boxes_b[:, 1]) * (boxes_b[:, 5] - boxes_b[:, 2]))[None, :]
    dh = torch.minimum(boxes_a[:, None, 5], boxes_b[None, :, 5]) - torch.maximum(boxes_a[:, None, 2], boxes_b[None, :, 2])
    flat = dh <= _EPS                                                # fmaxf(dh, EPS) == EPS -> 0
    vo = bev * dh.clamp(min=_EPS)
    iou = vo / (va + vb - vo).clamp(min=_EPS)
    return torch.where(flat, torch.zeros_like(iou), iou)


def boxes_iou3d_gpu(boxes_a, boxes_b, ans_iou):
    """(N,7) x (M,7) [x1,y1,z1,x2,y2,z2,ry] -> ans_iou (N,M) (iou3d.cpp:98-118)."""
    _lib.check_cuda(boxes_a, boxes_b, ans_iou)
    ans_iou.copy_(_iou3d_matrix(boxes_a.float(), boxes_b.float()).view_as(ans_iou))
    return 1


def boxes_iou3d_cpu(boxes_a, boxes_b, ans_iou):
    """Host twin (iou3d_cpu.cpp:305-337): BEV overlaps from libglenet_host.so, the rest in host tensor ops."""
    from ... import _host
    if boxes_a.is_cuda or boxes_b.is_cuda or ans_iou.is_cuda:
        raise _lib.GlxError("the *_cpu entry points take host tensors")
    a, b = boxes_a.detach().float().contiguous(), boxes_b.detach().float().contiguous()
    cols = [0, 1, 3, 4, 6]
    bev = torch.from_numpy(_host.iou3d_boxes_bev(a[:, cols].contiguous().numpy(), b[:, cols].contiguous().numpy(),
                                                 iou=False))
    va = ((a[:, 3] - a[:, 0]) * (a[:, 4] - a[:, 1]) * (a[:, 5] - a[:, 2]))[:, None]
    vb = ((b[:, 3] - b[:, 0]) * (b[:, 4] - b[:, 1]) * (b[:, 5] - b[:, 2]))[None, :]
    dh = torch.minimum(a[:, None, 5], b[None, :, 5]) - torch.maximum(a[:, None, 2], b[None, :, 2])
    vo = bev * dh.clamp(min=_EPS)
    iou = vo / (va + vb - vo).clamp(min=_EPS)
    ans_iou.copy_(torch.where(dh <= _EPS, torch.zeros_like(iou), iou).view_as(ans_iou))
    return 1


def _greedy(suppress, keep):
    """iou3d.cpp:139-153: walk the boxes in the given (score) order; a box that no earlier kept box suppresses is
    kept and suppresses the later ones its row marks.  suppress (N,N) bool on any device; keep: host int64."""
    if keep.is_cuda:
        raise _lib.GlxError("keep is the host int64 tensor of the reference's interface")
    m = suppress.cpu().numpy()
    n = m.shape[0]
    removed = [False] * n
    out = 0
    for i in range(n):
        if removed[i]:
            continue
        keep[out] = i
        out += 1
        row = m[i]
        for j in row[i + 1:].nonzero()[0]:
            removed[i + 1 + int(j)] = True
    return out


def nms_gpu(boxes, keep, nms_overlap_thresh):
    """boxes (N,5) [x1,y1,x2,y2,ry] in score order -> number kept, indices in keep[:n] (iou3d.cpp:120-166)."""
    _lib.check_cuda(boxes)
    n = boxes.shape[0]
    iou = torch.empty((n, n), dtype=torch.float32, device=boxes.device)
    if n:
        call("glx_iou3d_boxes_overlap_bev", boxes, n, boxes, n, 1, iou)
    return _greedy(iou > float(nms_overlap_thresh), keep)


def nms_3d_gpu(boxes, keep, nms_overlap_thresh):
    """boxes (N,7) [x1,y1,z1,x2,y2,z2,ry] in score order, 3-D IoU (iou3d.cpp:168-214)."""
    _lib.check_cuda(boxes)
    b = boxes.float()
    return _greedy(_iou3d_matrix(b, b) > float(nms_overlap_thresh), keep)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    """boxes (N,5) [x1,y1,x2,y2,ry], the angle ignored: axis-aligned IoU (iou3d_kernel.cu:411-422; iou3d.cpp:216-262)."""
    _lib.check_cuda(boxes)
    b = boxes.float()
    w = (torch.minimum(b[:, None, 2], b[None, :, 2]) - torch.maximum(b[:, None, 0], b[None, :, 0])).clamp(min=0)
    h = (torch.minimum(b[:, None, 3], b[None, :, 3]) - torch.maximum(b[:, None, 1], b[None, :, 1])).clamp(min=0)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    inter = w * h
    iou = inter / (area[:, None] + area[None, :] - inter).clamp(min=_EPS)
    return _greedy(iou > float(nms_overlap_thresh), keep)
