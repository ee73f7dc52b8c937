"""Drop-in for the pybind module `pcdet.ops.iou3d_nms.iou3d_nms_cuda`
(pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17): same function names, positional arguments,
caller-allocated outputs, int return.  Arithmetic: libglenet_hip.so."""
import torch

from ... import _lib
from ..._lib import call, query, size_arg, workspace


def _f(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), "expects contiguous float32"
    return t


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    _lib.check_cuda(boxes_a, boxes_b, ans_overlap)
    call("glx_boxes_overlap_bev", _f(boxes_a), boxes_a.shape[0], _f(boxes_b), boxes_b.shape[0], 0,
         _f(ans_overlap))
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    _lib.check_cuda(boxes_a, boxes_b, ans_iou)
    call("glx_boxes_overlap_bev", _f(boxes_a), boxes_a.shape[0], _f(boxes_b), boxes_b.shape[0], 1,
         _f(ans_iou))
    return 1


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """The reference's CPU entry point (iou3d_cpu.cpp:232-252): host tensors in and out, host arithmetic
    (libglenet_host.so, plain C++): stateless and free of any GPU runtime call, so the forked DataLoader
    workers that call it through boxes_bev_iou_cpu (database_sampler.py:246-247) may do so."""
    from ... import _host
    if boxes_a.is_cuda or boxes_b.is_cuda or ans_iou.is_cuda:
        raise _lib.GlxError("boxes_iou_bev_cpu takes host tensors (use boxes_iou_bev_gpu for device tensors)")
    out = _host.boxes_iou_bev(boxes_a.detach().float().contiguous().numpy(),
                              boxes_b.detach().float().contiguous().numpy())
    ans_iou.copy_(torch.from_numpy(out).view_as(ans_iou))
    return 1


def nms_device(boxes, thresh, normal=False):
    """boxes (N,7) sorted by score -> (keep int64 (N,) on device, num_out int32 (1,) on device)."""
    _lib.check_cuda(boxes)
    n = boxes.shape[0]
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=boxes.device)
    num = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    ws = workspace.get(query("glx_nms_workspace_bytes", n), boxes.device)
    call("glx_nms", _f(boxes), n, float(thresh), 1 if normal else 0, keep, num, ws,
         size_arg(ws.numel()))
    return keep, num


def nms_device_batch(boxes, thresh, normal=False, max_keep=0):
    """boxes (F,N,7), every frame sorted by score -> (keep int64 (F,N), num_out int32 (F,)), both
    on the device; one launch sequence for all frames.  max_keep > 0: a frame's sweep stops once
    that many boxes are kept (keep[f, :min(num, max_keep)] is what the full sweep would give)."""
    _lib.check_cuda(boxes)
    f, n = boxes.shape[0], boxes.shape[1]
    keep = torch.empty((f, max(n, 1)), dtype=torch.int64, device=boxes.device)
    num = torch.zeros(f, dtype=torch.int32, device=boxes.device)
    if f == 0:
        return keep, num
    ws = workspace.get(f * query("glx_nms_workspace_bytes", n), boxes.device)
    call("glx_nms_batch", _f(boxes), f, n, float(thresh), 1 if normal else 0, int(max_keep), keep, num,
         ws, size_arg(ws.numel()))
    return keep, num


def _nms_into_cpu_keep(boxes, keep, thresh, normal):
    k, num = nms_device(boxes, thresh, normal)
    n = int(num.item())
    keep[:n] = k[:n].cpu()
    return n


def nms_gpu(boxes, keep, nms_overlap_thresh):
    """keep: CPU int64 tensor (N), as in iou3d_nms.cpp:90-136; returns the number kept."""
    return _nms_into_cpu_keep(boxes, keep, nms_overlap_thresh, False)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    return _nms_into_cpu_keep(boxes, keep, nms_overlap_thresh, True)
