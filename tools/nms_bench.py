"""Rotated NMS timing at the sizes of GLENet_VR.yaml (GPU box): 9000 boxes thr 0.8 (train),
2048 thr 0.7 (test), 4096 thr 0.1 (single-stage GLENet-S/C feeding nms_func)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import synth  # noqa: E402
from glenet_amd.pcdet_ops.iou3d_nms import iou3d_nms_cuda, iou3d_nms_utils  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(3000)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for n, thr in ((9000, 0.8), (2048, 0.7), (4096, 0.1)):
    boxes = torch.from_numpy(synth.random_boxes(rng, n, xy_range=35.0, near_dup=0.7)).to(dev)
    scores = torch.rand(n, device=dev)
    order = scores.argsort(descending=True)
    bs = boxes[order].contiguous()
    t = timeit(lambda: iou3d_nms_cuda.nms_device(bs, thr))
    keep, _ = iou3d_nms_utils.nms_gpu(boxes, scores, thr)
    print("nms %5d boxes thr %.1f: %8.1f us (device keep list, no read-back), %d kept" % (n, thr, t, len(keep)))
    if n <= 4096:
        t2 = timeit(lambda: iou3d_nms_utils.boxes_iou_bev(bs, bs))
        print("    pairwise BEV IoU %d x %d: %8.1f us" % (n, n, t2))

# GLENet's variance-voting NMS (nms_func / new_nms_gpu, iou3d_nms_utils.py:200-273) at the sizes the
# single-stage models feed it (GLENet_S/C: every anchor above SCORE_THRESH, up to NMS_PRE_MAXSIZE 4096;
# two-stage: <= 100 RoIs).  Device time of IoU matrix + voting loop, no read-back.
for n in (100, 1000, 4096):
    boxes = torch.from_numpy(synth.random_boxes(rng, n, xy_range=35.0, near_dup=0.7)).to(dev)
    scores = torch.rand(n, device=dev) * 0.9 + 0.1
    var = torch.rand(n, 7, device=dev) * 0.2 + 0.01
    t = timeit(lambda: iou3d_nms_utils.nms_func_device(boxes, scores, 0.1, 0.0, var), n=5)
    s, _ = iou3d_nms_utils.nms_func_device(boxes, scores, 0.1, 0.0, var)
    print("variance-voting nms %5d boxes thr 0.1: %9.1f us (IoU matrix + voting block), %d kept" % (n, t, int((s > 0).sum())))
