"""CPU oracle of GLENet's detection hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; glenet_amd/ never does (the product path has no CPU fallback).

numpy in / numpy out.  Heavy loops live in glenet_oracle.c (built by oracle/build.py into
oracle/_build/liboracle.so); small glue that the reference keeps in Python is restated
here in numpy, each function citing the reference file:line it follows.
"""
import ctypes
import os

import numpy as np

from . import build as _build

_lib = None
_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int32)
_L = ctypes.POINTER(ctypes.c_int64)


def lib():
    global _lib
    if _lib is None:
        path = _build.LIB
        if not os.path.exists(path) or (
                os.path.exists(_build.SRC) and os.path.getmtime(path) < os.path.getmtime(_build.SRC)):
            _build.build()
        _lib = ctypes.CDLL(path)
    return _lib


def _f(a):
    return a.ctypes.data_as(_F)


def _i(a):
    return a.ctypes.data_as(_I)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ints(v):
    return (ctypes.c_int * len(v))(*[int(x) for x in v])


def _floats(v):
    return (ctypes.c_float * len(v))(*[float(x) for x in v])


# ------------------------------------------------------------------------------ voxelization
def grid_size_of(point_cloud_range, voxel_size):
    """data_processor.py:119-120."""
    r = np.asarray(point_cloud_range, dtype=np.float64)
    v = np.asarray(voxel_size, dtype=np.float64)
    return [int(g) for g in np.round((r[3:6] - r[0:3]) / v).astype(np.int64)]


def voxelize_hard(points, voxel_size, point_cloud_range, max_points, max_voxels):
    """One frame.  Returns voxels (Nv,max_points,C), coords (Nv,3) [z,y,x], num_points (Nv,)."""
    points = _f32(points)
    P, C = points.shape
    grid = grid_size_of(point_cloud_range, voxel_size)
    voxels = np.zeros((max_voxels, max_points, C), np.float32)
    coords = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    nv = lib().orc_voxelize_hard(_f(points), P, C, _floats(point_cloud_range), _floats(voxel_size),
                                 _ints(grid), max_points, max_voxels, _f(voxels), _i(coords), _i(num))
    assert nv >= 0
    return voxels[:nv].copy(), coords[:nv].copy(), num[:nv].copy()


def voxelize_hard_batch(frames, voxel_size, point_cloud_range, max_points, max_voxels):
    """Per-frame generator + collate_batch's batch-index padding (dataset.py:192-197)."""
    vs, cs, ns = [], [], []
    for b, pts in enumerate(frames):
        v, c, n = voxelize_hard(pts, voxel_size, point_cloud_range, max_points, max_voxels)
        vs.append(v)
        cs.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], axis=1))
        ns.append(n)
    return np.concatenate(vs), np.concatenate(cs), np.concatenate(ns)


def mean_vfe(voxels, num_points):
    voxels, num_points = _f32(voxels), _i32(num_points)
    nv, mp, c = voxels.shape
    out = np.zeros((nv, c), np.float32)
    lib().orc_mean_vfe(_f(voxels), _i(num_points), nv, mp, c, _f(out))
    return out


def voxelize_dynamic_mean(points, batch_idx, voxel_size, point_cloud_range):
    """DynamicMeanVFE.forward, dynamic_mean_vfe.py:53-72, in numpy (float32 arithmetic).
    points (P,C) xyz first; returns features (Nv,C), coords (Nv,4) [b,z,y,x]."""
    points = _f32(points)
    rng = np.asarray(point_cloud_range, np.float32)
    vsz = np.asarray(voxel_size, np.float32)
    grid = np.asarray(grid_size_of(point_cloud_range, voxel_size), np.int64)
    pc = np.floor((points[:, 0:3] - rng[0:3]) / vsz).astype(np.int32)           # :53
    mask = ((pc >= 0) & (pc < grid)).all(axis=1)                                  # :54
    points, pc = points[mask], pc[mask].astype(np.int64)
    b = np.asarray(batch_idx)[mask].astype(np.int64)
    sxyz, syz, sz = grid[0] * grid[1] * grid[2], grid[1] * grid[2], grid[2]
    merge = b * sxyz + pc[:, 0] * syz + pc[:, 1] * sz + pc[:, 2]                # :57-60
    unq, inv, cnt = np.unique(merge, return_inverse=True, return_counts=True)   # :63
    sums = np.zeros((len(unq), points.shape[1]), np.float32)
    np.add.at(sums, inv, points)                                                # scatter_mean :65
    mean = sums / cnt[:, None].astype(np.float32)
    coords = np.stack([unq // sxyz, (unq % sxyz) // syz, (unq % syz) // sz, unq % sz], 1)
    coords = coords[:, [0, 3, 2, 1]].astype(np.int32)                           # :68-72
    return mean.astype(np.float32), coords


# ------------------------------------------------------------------------------ sparse conv
def _triple(v):
    return [int(x) for x in v] if isinstance(v, (list, tuple)) else [int(v)] * 3


class Rules:
    """Classic per-offset pair lists (what spconv calls indice pairs)."""

    def __init__(self, pairs_in, pairs_out, n_pairs, out_indices, out_shape, n_in):
        self.pairs_in, self.pairs_out, self.n_pairs = pairs_in, pairs_out, n_pairs
        self.out_indices, self.out_shape, self.n_in = out_indices, out_shape, n_in

    @property
    def R(self):
        return int(self.n_pairs.sum())

    def nbr_table(self):
        """(N_out, K) input row per (output row, offset), -1 where absent."""
        K = len(self.n_pairs)
        t = np.full((len(self.out_indices), K), -1, np.int32)
        for k in range(K):
            n = self.n_pairs[k]
            t[self.pairs_out[k, :n], k] = self.pairs_in[k, :n]
        return t


def build_rules(indices, spatial_shape, ksize, stride=1, padding=0, subm=True, dilation=1):
    indices = _i32(indices)
    ksize, stride, padding, dilation = _triple(ksize), _triple(stride), _triple(padding), _triple(dilation)
    K = ksize[0] * ksize[1] * ksize[2]
    N_in = len(indices)
    shape = [int(s) for s in spatial_shape]
    if subm:
        out_idx, oshape = indices, shape
    else:
        oshape = [(s + 2 * p - d * (k - 1) - 1) // st + 1
                  for s, p, k, st, d in zip(shape, padding, ksize, stride, dilation)]
        buf = np.zeros((max(N_in, 1) * K, 4), np.int32)
        n_out = lib().orc_outset_strided_dil(_i(indices), N_in, _ints(shape), _ints(ksize),
                                             _ints(stride), _ints(padding), _ints(dilation), _ints(oshape), _i(buf))
        out_idx = buf[:n_out].copy()
    N_out = len(out_idx)
    ld = max(N_in, N_out, 1)
    pin = np.zeros((K, ld), np.int32)
    pout = np.zeros((K, ld), np.int32)
    npairs = np.zeros((K,), np.int32)
    lib().orc_build_rules_dil(_i(indices), N_in, _i(_i32(out_idx)), N_out, _ints(shape), _ints(ksize),
                              _ints(stride), _ints(padding), _ints(dilation), 1 if subm else 0, _i(pin), _i(pout),
                              _i(npairs))
    return Rules(pin, pout, npairs, out_idx, oshape, N_in)


def sconv_forward(features, weight, rules, bias=None):
    """features (N_in,Cin), weight (K,Cin,Cout) -> (N_out,Cout)."""
    features, weight = _f32(features), _f32(weight)
    K, cin, cout = weight.shape
    n_out = len(rules.out_indices)
    out = np.zeros((n_out, cout), np.float32)
    b = _f(_f32(bias)) if bias is not None else None
    lib().orc_sconv_forward(_f(features), _f(weight), b, _i(rules.pairs_in), _i(rules.pairs_out),
                            _i(rules.n_pairs), K, rules.pairs_in.shape[1], n_out, cin, cout, _f(out))
    return out


def sconv_backward(features, weight, grad_out, rules):
    features, weight, grad_out = _f32(features), _f32(weight), _f32(grad_out)
    K, cin, cout = weight.shape
    din = np.zeros_like(features)
    dw = np.zeros_like(weight)
    lib().orc_sconv_backward(_f(features), _f(weight), _f(grad_out), _i(rules.pairs_in),
                             _i(rules.pairs_out), _i(rules.n_pairs), K, rules.pairs_in.shape[1],
                             len(features), cin, cout, _f(din), _f(dw))
    return din, dw


def dense(features, indices, batch_size, spatial_shape):
    features, indices = _f32(features), _i32(indices)
    n, c = features.shape
    d, h, w = [int(s) for s in spatial_shape]
    out = np.zeros((batch_size, c, d, h, w), np.float32)
    lib().orc_dense(_f(features), _i(indices), n, c, batch_size, d, h, w, _f(out))
    return out


# ------------------------------------------------------------------------------ IoU / NMS
def boxes_overlap_bev(a, b):
    a, b = _f32(a), _f32(b)
    out = np.zeros((len(a), len(b)), np.float32)
    lib().orc_boxes_overlap_bev(_f(a), len(a), _f(b), len(b), _f(out))
    return out


def boxes_iou_bev(a, b):
    """boxes_bev_iou_cpu, iou3d_nms_utils.py:52-68."""
    a, b = _f32(a), _f32(b)
    out = np.zeros((len(a), len(b)), np.float32)
    lib().orc_boxes_iou_bev(_f(a), len(a), _f(b), len(b), _f(out))
    return out


def boxes_iou3d(a, b):
    """boxes_iou3d_gpu glue, iou3d_nms_utils.py:88-121 (float32 like torch)."""
    a, b = _f32(a), _f32(b)
    a_max = (a[:, 2] + a[:, 5] / 2).reshape(-1, 1)
    a_min = (a[:, 2] - a[:, 5] / 2).reshape(-1, 1)
    b_max = (b[:, 2] + b[:, 5] / 2).reshape(1, -1)
    b_min = (b[:, 2] - b[:, 5] / 2).reshape(1, -1)
    ov_bev = boxes_overlap_bev(a, b)
    ov_h = np.clip(np.minimum(a_max, b_max) - np.maximum(a_min, b_min), 0, None)
    ov3d = ov_bev * ov_h
    va = (a[:, 3] * a[:, 4] * a[:, 5]).reshape(-1, 1)
    vb = (b[:, 3] * b[:, 4] * b[:, 5]).reshape(1, -1)
    return (ov3d / np.clip(va + vb - ov3d, 1e-6, None)).astype(np.float32)


def iou_bev_pairs(boxes, pairs):
    """IoU(boxes[i], boxes[j]) (row box first, as nms_kernel evaluates it) for pairs (P,2) int32."""
    boxes, pairs = _f32(boxes), _i32(pairs)
    out = np.zeros(len(pairs), np.float32)
    lib().orc_iou_bev_pairs(_f(boxes), _i(pairs), ctypes.c_longlong(len(pairs)), _f(out))
    return out


def nms_from_pairs(n, pairs, suppress):
    """The host sweep of iou3d_nms.cpp:116-132 on a sparse suppression relation: pairs (P,2) with i < j,
    suppress (P,) bool = iou(i, j) > thresh.  -> kept indices (ascending)."""
    pairs = np.asarray(pairs)[np.asarray(suppress, bool)]
    order = np.argsort(pairs[:, 0], kind="stable")
    pairs = pairs[order]
    starts = np.searchsorted(pairs[:, 0], np.arange(n + 1))
    removed = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        removed[pairs[starts[i]:starts[i + 1], 1]] = True
    return np.asarray(keep, np.int64)


def nms_sorted(boxes_sorted, thresh, normal=False):
    """iou3d_nms_cuda.nms_gpu / nms_normal_gpu on boxes already sorted by score."""
    boxes_sorted = _f32(boxes_sorted)
    keep = np.zeros((max(len(boxes_sorted), 1),), np.int64)
    n = lib().orc_nms(_f(boxes_sorted), len(boxes_sorted), ctypes.c_float(thresh),
                      1 if normal else 0, keep.ctypes.data_as(_L))
    return keep[:n].copy()


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, normal=False):
    """nms_gpu / nms_normal_gpu wrappers, iou3d_nms_utils.py:182-197, 276-290.
    torch.sort(descending) is not stable; callers must use distinct scores."""
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    if pre_maxsize is not None and not normal:
        order = order[:pre_maxsize]
    keep = nms_sorted(np.asarray(boxes)[order], thresh, normal)
    return order[keep]


def limit_period(val, offset=0.5, period=np.pi):
    """common_utils.py:21-24 (float32)."""
    val = np.asarray(val, np.float32)
    return (val - np.floor(val / np.float32(period) + np.float32(offset)) * np.float32(period)).astype(np.float32)


def nms_func(boxes, scores, iou_threshold, score_threshold=0, variance=None):
    """GLENet variance-voting NMS, iou3d_nms_utils.py:227-273, restated line by line.
    boxes (N,7) and scores (N,) are float32 copies; returns (scores, boxes)."""
    boxes = np.array(boxes, np.float32, copy=True)
    scores = np.array(scores, np.float32, copy=True)
    undone = scores >= score_threshold                                        # :229
    ious_all = boxes_iou_bev(boxes, boxes)                                    # :235
    while undone.sum() > 0:                                                   # :237
        idx = scores[undone].argmax()
        idx = undone.nonzero()[0][idx]
        top_box = boxes[idx:idx + 1]
        _boxes = boxes[undone]
        ious = ious_all[undone, idx]
        if variance is not None:
            _var = variance[undone, :7]
            ioumask = ious > iou_threshold
            klbox = _boxes[ioumask]
            far = np.abs(klbox[:, 6] - top_box[:, 6]) >= np.pi * 3 / 2
            if top_box[:, 6] > 0:                                             # :251-254
                klbox[far, 6] += np.pi * 2
            else:
                klbox[far, 6] -= np.pi * 2
            kliou = ious[ioumask]
            klvar = _var[ioumask]
            std_iou_sigma = 0.05
            pi = (np.exp(-1 * (1 - kliou) ** 2 / std_iou_sigma)).reshape(-1, 1)
            pi = pi / klvar
            pi[np.abs(klbox[:, 6] - top_box[:, 6]) >= np.pi / 4, 6] = 0
            pi = pi / pi.sum(0)
            boxes[idx, :7] = (pi * klbox[:, :7]).sum(0)                       # :266
        undone[idx] = False
        scores[undone] *= (ious_all[undone, idx] < iou_threshold)             # :269
        undone[scores < score_threshold] = False
    return scores, boxes


def new_nms_gpu(boxes, scores, iou_threshold, score_threshold=0, variance=None):
    """iou3d_nms_utils.py:200-224."""
    boxes = np.array(boxes, np.float32, copy=True)
    boxes[:, 6] = limit_period(boxes[:, 6], offset=0.5, period=np.pi * 2)
    new_scores, new_boxes = nms_func(boxes, scores, iou_threshold, score_threshold, variance)
    keep = (new_scores > 0).nonzero()[0]
    keep = keep[new_scores[keep].argsort()[::-1]]
    return keep, new_boxes


def post_processing(cls_preds, box_preds, box_std_preds=None, labels=None, score_thresh=0.3, post_score_thresh=0.81,
                    nms_thresh=0.1, nms_pre_maxsize=4096, nms_post_maxsize=500, normalized=False):
    """ONE frame of Detector3DTemplate.post_processing with NMS_TYPE new_nms_gpu (pcdet/models/detectors/
    detector3d_template.py:196-316) -> model_nms_utils.class_agnostic_nms (:6-62) -> new_nms_gpu above, restated in numpy:
    cls_preds (R,C) logits, box_preds (R,7), box_std_preds (R,7) log-variances or None, labels (R,) 1-based or None.
    Returns (boxes (n,7), scores (n,), labels (n,), source index (n,)).  torch.topk's order among equal scores is
    unspecified in the reference; here lower index first (stable)."""
    cls = np.asarray(cls_preds, np.float32)
    if not normalized:                                                         # :213-214 torch.sigmoid in float32
        cls = (np.float32(1) / (np.float32(1) + np.exp(-cls, dtype=np.float32))).astype(np.float32)
    scores, arg = cls.max(1), cls.argmax(1)                                    # :258
    lab = np.asarray(labels) if labels is not None else arg + 1                # :259-263
    box = np.asarray(box_preds, np.float32)
    mask = scores >= np.float32(score_thresh) if score_thresh is not None else np.ones(len(scores), bool)   # nms_utils :12-15
    s, bx = scores[mask], box[mask]
    var = np.exp(np.asarray(box_std_preds, np.float32))[mask] if box_std_preds is not None else None        # :18-21
    if len(s) == 0:
        return np.zeros((0, 7), np.float32), s, lab[:0], np.zeros(0, np.int64)
    order = np.argsort(-s, kind="stable")[:min(nms_pre_maxsize, len(s))]      # :26 topk
    keep, new_boxes = new_nms_gpu(bx[order, :7], s[order], nms_thresh, 0, var[order] if var is not None else None)
    keep = keep[:nms_post_maxsize]                                             # :45-46
    sel = np.nonzero(mask)[0][order[keep]]                                     # :58-60
    out_boxes, out_scores, out_labels = new_boxes[keep], scores[sel], lab[sel]
    if post_score_thresh is not None:                                          # detector3d_template.py:295-300
        m = out_scores > np.float32(post_score_thresh)
        out_boxes, out_scores, out_labels, sel = out_boxes[m], out_scores[m], out_labels[m], sel[m]
    return out_boxes, out_scores, out_labels, sel


# ---- iou3d (older) library, [x1,y1,x2,y2,ry] boxes
def iou3d_boxes_overlap_bev(a, b):
    a, b = _f32(a), _f32(b)
    out = np.zeros((len(a), len(b)), np.float32)
    lib().orc_iou3d_boxes_overlap_bev(_f(a), len(a), _f(b), len(b), _f(out))
    return out


def iou3d_boxes_iou_bev(a, b):
    a, b = _f32(a), _f32(b)
    out = np.zeros((len(a), len(b)), np.float32)
    lib().orc_iou3d_boxes_iou_bev(_f(a), len(a), _f(b), len(b), _f(out))
    return out


def iou3d_boxes_aligned_overlap_bev(a, b):
    a, b = _f32(a), _f32(b)
    out = np.zeros((len(a), 1), np.float32)
    lib().orc_iou3d_boxes_aligned_overlap_bev(_f(a), _f(b), len(a), _f(out))
    return out


def iou3d_boxes_iou3d(a, b):
    """iou_3d of the older library for (N,7) x (M,7) [x1,y1,z1,x2,y2,z2,ry] boxes, iou3d_kernel.cu:256-268 /
    iou3d_cpu.cpp:305-337: BEV overlap of the (x1,y1,x2,y2,ry) rectangles times the height overlap (EPS 1e-8
    floor; a floor-sized overlap counts as none) over the union of the volumes."""
    a, b = _f32(a), _f32(b)
    eps = np.float32(1e-8)
    bev = iou3d_boxes_overlap_bev(a[:, [0, 1, 3, 4, 6]], b[:, [0, 1, 3, 4, 6]])
    va = ((a[:, 3] - a[:, 0]) * (a[:, 4] - a[:, 1]) * (a[:, 5] - a[:, 2])).reshape(-1, 1)
    vb = ((b[:, 3] - b[:, 0]) * (b[:, 4] - b[:, 1]) * (b[:, 5] - b[:, 2])).reshape(1, -1)
    dh = np.maximum(np.minimum(a[:, 5].reshape(-1, 1), b[:, 5].reshape(1, -1))
                    - np.maximum(a[:, 2].reshape(-1, 1), b[:, 2].reshape(1, -1)), eps)
    vo = bev * dh
    out = vo / np.maximum(va + vb - vo, eps)
    out[dh == eps] = 0
    return out.astype(np.float32)


def iou3d_nms_sorted(boxes_sorted, thresh, kind="bev"):
    """nms_gpu / nms_3d_gpu / nms_normal_gpu of the older library on boxes in score order (iou3d.cpp:120-262: the
    mask kernels mark IoU > thresh, the host loop keeps a box unless an earlier kept one marked it).
    kind: 'bev' (N,5) rotated, '3d' (N,7) [x1,y1,z1,x2,y2,z2,ry], 'normal' (N,5) axis-aligned
    (iou3d_kernel.cu:411-422)."""
    b = _f32(boxes_sorted)
    if kind == "bev":
        iou = iou3d_boxes_iou_bev(b, b)
    elif kind == "3d":
        iou = iou3d_boxes_iou3d(b, b)
    else:
        w = np.maximum(np.minimum(b[:, None, 2], b[None, :, 2]) - np.maximum(b[:, None, 0], b[None, :, 0]), 0)
        h = np.maximum(np.minimum(b[:, None, 3], b[None, :, 3]) - np.maximum(b[:, None, 1], b[None, :, 1]), 0)
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        iou = (w * h) / np.maximum(area[:, None] + area[None, :] - w * h, np.float32(1e-8))
    removed = np.zeros(len(b), bool)
    keep = []
    for i in range(len(b)):
        if removed[i]:
            continue
        keep.append(i)
        removed[i + 1:] |= iou[i, i + 1:] > np.float32(thresh)
    return np.asarray(keep, np.int64)


def softnms(boxes, scores, iou_threshold, soft_sigma, score_threshold, soft_mode="gaussian", variance=None):
    """iou3d_nms_utils.py:313-356 restated statement by statement (float32, host loops; small cases only): the IoU
    column of the current top box is recomputed every iteration from the CURRENT boxes, as the reference does."""
    boxes = np.array(boxes, np.float32, copy=True)
    scores = np.array(scores, np.float32, copy=True)
    variance = None if variance is None else np.asarray(variance, np.float32)
    undone = scores >= np.float32(score_threshold)                                   # :316
    while undone.sum() > 1:                                                          # :317
        idx = undone.nonzero()[0][scores[undone].argmax()]                           # :318-319
        top = boxes[idx:idx + 1].copy()
        undone[idx] = False                                                          # :321
        rest = boxes[undone]
        ious = boxes_iou_bev(rest[:, :7], top[:, :7]).reshape(-1)                    # :324
        if variance is not None:                                                     # :326-342
            m = ious > np.float32(iou_threshold)
            klbox = np.concatenate([rest[m], top], 0)
            klvar = np.concatenate([variance[undone][m][:, :6], variance[idx:idx + 1, :6]], 0)
            w = np.exp(np.float32(-1) * (np.float32(1) - ious[m]) ** 2 / np.float32(0.05)).astype(np.float32)
            w = np.concatenate([w, np.ones(1, np.float32)])[:, None]
            w = w / klvar
            w = w / w.sum(0)
            boxes[idx, :6] = (w * klbox[:, :6]).sum(0)
        if soft_mode == "linear":                                                    # :304-311
            scale = np.ones_like(ious)
            scale[ious >= np.float32(soft_sigma)] = 1 - ious[ious >= np.float32(soft_sigma)]
        else:
            scale = np.exp(-ious ** 2 / np.float32(soft_sigma)).astype(np.float32)
        scores[undone] *= scale                                                      # :352
        undone[scores < np.float32(score_threshold)] = False                         # :353
    return scores, boxes


def softnms_gpu(boxes, scores, iou_threshold, score_threshold=0.1, soft_mode="gaussian", variance=None, soft_sigma=0.3):
    """iou3d_nms_utils.py:292-302."""
    new_scores, new_boxes = softnms(boxes, scores, iou_threshold, soft_sigma, score_threshold, soft_mode, variance)
    keep = (new_scores > np.float32(score_threshold)).nonzero()[0]
    keep = keep[np.argsort(-new_scores[keep], kind="stable")]
    return keep, new_boxes


# ------------------------------------------------------------------------------ point ops
def points_in_boxes_cpu(points, boxes):
    """roiaware_pool3d_utils.points_in_boxes_cpu: (N boxes, P points) int32."""
    points, boxes = _f32(points), _f32(boxes)
    out = np.zeros((len(boxes), len(points)), np.int32)
    lib().orc_points_in_boxes_cpu(_f(boxes), len(boxes), _f(points), len(points), _i(out))
    return out


def points_in_boxes_gpu(points, boxes):
    """points (B,P,3), boxes (B,T,7) -> (B,P) first containing box or -1."""
    points, boxes = _f32(points), _f32(boxes)
    B, P, _ = points.shape
    out = np.zeros((B, P), np.int32)
    lib().orc_points_in_boxes_gpu(_f(boxes), B, boxes.shape[1], _f(points), P, _i(out))
    return out


def roiaware_pool3d_forward(rois, pts, feat, out_size, max_pts, method):
    rois, pts, feat = _f32(rois), _f32(pts), _f32(feat)
    ox, oy, oz = _triple(out_size)
    N, C = len(rois), feat.shape[1]
    pooled = np.zeros((N, ox, oy, oz, C), np.float32)
    argmax = np.zeros((N, ox, oy, oz, C), np.int32)
    pidx = np.zeros((N, ox, oy, oz, max_pts), np.int32)
    lib().orc_roiaware_pool3d_forward(_f(rois), N, _f(pts), len(pts), _f(feat), C, ox, oy, oz,
                                      max_pts, {"max": 0, "avg": 1}[method], _i(argmax), _i(pidx),
                                      _f(pooled))
    return pooled, argmax, pidx


def roiaware_pool3d_backward(pidx, argmax, grad_out, num_pts, method):
    pidx, argmax, grad_out = _i32(pidx), _i32(argmax), _f32(grad_out)
    N, ox, oy, oz, C = grad_out.shape
    gin = np.zeros((int(num_pts), C), np.float32)
    lib().orc_roiaware_pool3d_backward(_i(pidx), _i(argmax), _f(grad_out), N, ox, oy, oz, C,
                                       pidx.shape[-1], {"max": 0, "avg": 1}[method], _f(gin))
    return gin


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    """box_utils.enlarge_box3d, pcdet/utils/box_utils.py:127-139."""
    b = np.array(boxes3d, np.float32, copy=True)
    b[:, 3:6] += np.asarray(extra_width, np.float32)[None, :]
    return b


def roipoint_pool3d(points, feats, boxes3d, pool_extra_width, num_sampled):
    """RoIPointPool3dFunction.forward, roipoint_pool3d_utils.py:31-60."""
    points, feats, boxes3d = _f32(points), _f32(feats), _f32(boxes3d)
    B, Np, _ = points.shape
    M, C = boxes3d.shape[1], feats.shape[2]
    ew = pool_extra_width if isinstance(pool_extra_width, (list, tuple)) else [pool_extra_width] * 3
    big = enlarge_box3d(boxes3d.reshape(-1, 7), ew).reshape(B, M, 7)
    pooled = np.zeros((B, M, num_sampled, 3 + C), np.float32)
    empty = np.zeros((B, M), np.int32)
    lib().orc_roipoint_pool3d(_f(points), _f(_f32(big)), _f(feats), B, Np, M, C, num_sampled,
                              _f(pooled), _i(empty))
    return pooled, empty


def voxel_query(max_range, radius, nsample, xyz, new_xyz, new_coords, point_indices):
    """VoxelQuery.forward, voxel_query_utils.py:13-43: returns (idx, empty_ball_mask)."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    new_coords, point_indices = _i32(new_coords), _i32(point_indices)
    M = len(new_coords)
    B, Z, Y, X = point_indices.shape
    idx = np.zeros((M, nsample), np.int32)
    zr, yr, xr = max_range
    lib().orc_voxel_query(M, Z, Y, X, nsample, ctypes.c_float(radius), zr, yr, xr, _f(new_xyz),
                          _f(xyz), _i(new_coords), _i(point_indices), _i(idx))
    empty = idx[:, 0] == -1
    idx[empty] = 0
    return idx, empty


def ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
    """BallQuery.forward, pointnet2_utils.py:11-43."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    xbc, nbc = _i32(xyz_batch_cnt), _i32(new_xyz_batch_cnt)
    M = len(new_xyz)
    idx = np.zeros((M, nsample), np.int32)
    lib().orc_ball_query(len(xbc), M, ctypes.c_float(radius), nsample, _f(new_xyz), _i(nbc),
                         _f(xyz), _i(xbc), _i(idx))
    empty = idx[:, 0] == -1
    idx[empty] = 0
    return idx, empty


def group_points(features, features_batch_cnt, idx, idx_batch_cnt):
    features, idx = _f32(features), _i32(idx)
    fbc, ibc = _i32(features_batch_cnt), _i32(idx_batch_cnt)
    M, ns = idx.shape
    C = features.shape[1]
    out = np.zeros((M, C, ns), np.float32)
    lib().orc_group_points(len(ibc), M, C, ns, _f(features), _i(fbc), _i(idx), _i(ibc), _f(out))
    return out


def group_points_grad(grad_out, idx, idx_batch_cnt, features_batch_cnt, N):
    grad_out, idx = _f32(grad_out), _i32(idx)
    fbc, ibc = _i32(features_batch_cnt), _i32(idx_batch_cnt)
    M, C, ns = grad_out.shape
    g = np.zeros((N, C), np.float32)
    lib().orc_group_points_grad(len(ibc), M, C, int(N), ns, _f(grad_out), _i(idx), _i(ibc), _i(fbc), _f(g))
    return g


def generate_voxel2pinds(indices, batch_size, spatial_shape):
    """common_utils.generate_voxel2pinds, pcdet/utils/common_utils.py:226-243."""
    indices = np.asarray(indices).astype(np.int64)
    v2p = -np.ones((batch_size, *[int(s) for s in spatial_shape]), np.int32)
    v2p[indices[:, 0], indices[:, 1], indices[:, 2], indices[:, 3]] = np.arange(len(indices), dtype=np.int32)
    return v2p


def stack_farthest_point_sample(xyz, xyz_batch_cnt, npoint):
    """StackFarthestPointSampling.forward, pointnet2_utils.py:190-218 -> (sum npoint,) global indices."""
    xyz, xbc = _f32(xyz), _i32(xyz_batch_cnt)
    npt = _i32(np.full(len(xbc), npoint) if np.isscalar(npoint) else npoint)
    temp = np.full(len(xyz), 1e10, np.float32)
    out = np.zeros(int(npt.sum()), np.int32)
    lib().orc_stack_fps(len(xbc), _f(xyz), _i(xbc), _f(temp), _i(npt), _i(out))
    return out


def three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt):
    """ThreeNN.forward, pointnet2_utils.py:227-254 -> (dist (N,3) = sqrt of the kernel's dist2, idx (N,3))."""
    unknown, known = _f32(unknown), _f32(known)
    ubc, kbc = _i32(unknown_batch_cnt), _i32(known_batch_cnt)
    d2 = np.zeros((len(unknown), 3), np.float32)
    idx = np.zeros((len(unknown), 3), np.int32)
    lib().orc_three_nn(len(ubc), len(unknown), _f(unknown), _i(ubc), _f(known), _i(kbc), _f(d2), _i(idx))
    return np.sqrt(d2), idx


def three_interpolate(features, idx, weight):
    features, idx, weight = _f32(features), _i32(idx), _f32(weight)
    out = np.zeros((len(idx), features.shape[1]), np.float32)
    lib().orc_three_interpolate(len(idx), features.shape[1], _f(features), _i(idx), _f(weight), _f(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, M):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    g = np.zeros((int(M), grad_out.shape[1]), np.float32)
    lib().orc_three_interpolate_grad(len(idx), grad_out.shape[1], _f(grad_out), _i(idx), _f(weight), _f(g))
    return g


def libm_eval(fn, x, y=None):
    """sinf / cosf / atanf / atan2f of the host libm on float32 arrays (fn = "sin" | "cos" | "atan" | "atan2")."""
    code = {"sin": 0, "cos": 1, "atan": 2, "atan2": 3}[fn]
    x = _f32(x).ravel()
    y = _f32(y).ravel() if y is not None else x
    out = np.empty_like(x)
    lib().orc_libm_eval(code, _f(x), _f(y), ctypes.c_longlong(x.size), _f(out))
    return out


# ------------------------------------------------------------------------------ pointnet2_batch (batch layout)
def batch_ball_query(radius, nsample, xyz, new_xyz):
    """BallQuery.forward, pointnet2_batch/pointnet2_utils.py:198-222: xyz (B,N,3), new_xyz (B,m,3) ->
    idx (B,m,nsample) int32, zero rows for balls without points."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    B, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((B, m, nsample), np.int32)
    lib().orc_batch_ball_query(B, n, m, ctypes.c_float(radius), nsample, _f(new_xyz), _f(xyz), _i(idx))
    return idx


def batch_farthest_point_sample(xyz, npoint):
    """FarthestPointSampling.forward, pointnet2_batch/pointnet2_utils.py:10-32 -> (B, npoint) local indices."""
    xyz = _f32(xyz)
    B, n, _ = xyz.shape
    temp = np.full((B, n), 1e10, np.float32)
    out = np.zeros((B, npoint), np.int32)
    lib().orc_batch_fps(B, n, int(npoint), _f(xyz), _f(temp), _i(out))
    return out


def batch_three_nn(unknown, known):
    """ThreeNN.forward, pointnet2_batch/pointnet2_utils.py:76-101 -> (dist = sqrt(dist2) (B,n,3), idx (B,n,3))."""
    unknown, known = _f32(unknown), _f32(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.zeros((B, n, 3), np.float32)
    idx = np.zeros((B, n, 3), np.int32)
    lib().orc_batch_three_nn(B, n, m, _f(unknown), _f(known), _f(d2), _i(idx))
    return np.sqrt(d2), idx


def batch_group_points(points, idx):
    """group_points_kernel_fast, pointnet2_batch/src/group_points_gpu.cu:57-78:
    out[b, c, p, s] = points[b, c, idx[b, p, s]]; with a 2-D idx (B, m) it is gather_points_kernel_fast
    (sampling_gpu.cu:14-33)."""
    points, idx = _f32(points), _i32(idx)
    B, C, _ = points.shape
    flat = np.take_along_axis(points, np.broadcast_to(idx.reshape(B, 1, -1), (B, C, idx[0].size)).astype(np.int64), 2)
    return flat.reshape((B, C) + idx.shape[1:])


def batch_group_points_grad(grad_out, idx, n):
    """group_points_grad_kernel_fast (group_points_gpu.cu:14-32) / gather_points_grad_kernel_fast
    (sampling_gpu.cu:52-71): scatter-add into (B, C, n), sequential order (the GPU's atomics are unordered)."""
    grad_out, idx = _f32(grad_out), _i32(idx)
    B, C = grad_out.shape[:2]
    g = np.zeros((B, C, int(n)), np.float32)
    go = grad_out.reshape(B, C, -1)
    ii = idx.reshape(B, -1)
    for b in range(B):
        for c in range(C):
            np.add.at(g[b, c], ii[b], go[b, c])
    return g


def batch_three_interpolate(points, idx, weight):
    """three_interpolate_kernel_fast, pointnet2_batch/src/interpolate_gpu.cu:84-106: points (B,C,m),
    idx / weight (B,n,3) -> (B,C,n), the three products summed left to right in float."""
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    B, C, _ = points.shape
    out = np.zeros((B, C, idx.shape[1]), np.float32)
    for b in range(B):
        p = points[b]
        acc = weight[b, :, 0][None] * p[:, idx[b, :, 0]]
        acc = acc + weight[b, :, 1][None] * p[:, idx[b, :, 1]]
        out[b] = acc + weight[b, :, 2][None] * p[:, idx[b, :, 2]]
    return out


def batch_three_interpolate_grad(grad_out, idx, weight, m):
    """three_interpolate_grad_kernel_fast, interpolate_gpu.cu:130-153 -> (B, C, m)."""
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    B, C, _ = grad_out.shape
    g = np.zeros((B, C, int(m)), np.float32)
    for b in range(B):
        for c in range(C):
            for j in range(3):
                np.add.at(g[b, c], idx[b, :, j], grad_out[b, c] * weight[b, :, j])
    return g


# ------------------------------------------------------------------------------ vector pool (PV-RCNN++)
def query_stacked_local_neighbor_idxs(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, avg_length,
                                      max_dist, nsample, neighbor_type):
    """One call of query_stacked_local_neighbor_idxs_wrapper_stack -> (stack_neighbor_idxs (avg*M,), start_len (M,2),
    cumsum).  Segment order = ascending new-point index (see glenet_oracle.c)."""
    sx, nx = _f32(support_xyz), _f32(new_xyz)
    xbc, nbc = _i32(xyz_batch_cnt), _i32(new_xyz_batch_cnt)
    m = len(nx)
    stack = np.zeros(max(avg_length * m, 1), np.int32)
    start_len = np.zeros((m, 2), np.int32)
    cum = lib().orc_query_stacked_local_neighbor_idxs(_f(sx), _i(xbc), _f(nx), _i(nbc), len(xbc), m, _i(stack),
                                                      _i(start_len), int(avg_length), ctypes.c_float(max_dist),
                                                      int(nsample), int(neighbor_type))
    return stack, start_len, int(cum)


def three_nn_for_vector_pool_by_two_step(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_grid_centers, new_xyz_batch_cnt,
                                         max_neighbour_distance, nsample, neighbor_type, avg_length, num_total_grids,
                                         neighbor_distance_multiplier):
    """ThreeNNForVectorPoolByTwoStep.forward, pointnet2_utils.py:306-352 (the host retry loop included)
    -> (dist (M,G,3), idx (M,G,3), avg_length)."""
    m = len(new_xyz)
    while True:
        stack, start_len, cum = query_stacked_local_neighbor_idxs(
            support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, avg_length,
            max_neighbour_distance * neighbor_distance_multiplier, nsample, neighbor_type)
        max_sum = avg_length * m
        avg_length = cum // m + int(cum % m > 0)
        if cum <= max_sum:
            break
    centers = _f32(new_xyz_grid_centers)
    d2 = np.zeros(centers.shape, np.float32)
    idx = np.full(centers.shape, -1, np.int32)
    stack = np.ascontiguousarray(stack[:cum])
    lib().orc_query_three_nn_by_stacked_local_idxs(_f(_f32(support_xyz)), _f(centers), _i(idx), _f(d2), _i(stack),
                                                   _i(start_len), m, int(num_total_grids))
    with np.errstate(invalid="ignore"):
        return np.sqrt(d2), idx, avg_length


def vector_pool(support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, num_grid, max_dist,
                num_c_out_each_grid, use_xyz, num_mean_points_per_grid=100, nsample=-1, neighbor_type=0, pooling_type=0):
    """VectorPoolWithVoxelQuery.forward, pointnet2_utils.py:360-431 -> (new_features, new_local_xyz,
    num_mean_points_per_grid, point_cnt_of_grid, grouped_idxs)."""
    sx, sf, nx = _f32(support_xyz), _f32(support_features), _f32(new_xyz)
    xbc, nbc = _i32(xyz_batch_cnt), _i32(new_xyz_batch_cnt)
    gx, gy, gz = num_grid
    G = gx * gy * gz
    c_out = num_c_out_each_grid * G
    m, c_in = len(nx), sf.shape[1]
    while True:
        nf = np.zeros((m, c_out), np.float32)
        nl = np.zeros((m, 3 * G), np.float32)
        pc = np.zeros((m, G), np.int32)
        max_sum = num_mean_points_per_grid * m
        gi = np.zeros((max(max_sum, 1), 3), np.int32)
        cum = lib().orc_vector_pool(_f(sx), _f(sf), _i(xbc), _f(nx), _i(nbc), len(xbc), m, c_in, c_out, gx, gy, gz,
                                    ctypes.c_float(max_dist), 1 if use_xyz else 0, int(max_sum), int(nsample),
                                    int(neighbor_type), int(pooling_type), _f(nf), _f(nl), _i(pc), _i(gi))
        num_mean_points_per_grid = cum // m + int(cum % m > 0)
        if cum <= max_sum:
            break
    gi = gi[:cum]
    norm = np.clip(pc[:, :, None].astype(np.float32), 1e-6, None)
    nf = (nf.reshape(-1, G, num_c_out_each_grid) / norm).reshape(-1, c_out)
    if use_xyz:
        nl = (nl.reshape(-1, G, 3) / norm).reshape(-1, G * 3)
    return nf, nl, num_mean_points_per_grid, pc, gi


def vector_pool_grad(grad_new_features, point_cnt_of_grid, grouped_idxs, N, num_c_in):
    g = _f32(grad_new_features)
    pc, gi = _i32(point_cnt_of_grid), _i32(grouped_idxs)
    out = np.zeros((int(N), int(num_c_in)), np.float32)
    lib().orc_vector_pool_grad(_f(g), _i(pc), _i(gi), len(gi), int(num_c_in), g.shape[1], pc.shape[1], _f(out))
    return out


# ---------------------------------------------------------------------------------------------------------------
# Dense convolutions of the BEV backbone (pcdet/models/backbones_2d/base_bev_backbone.py:30-66): plain fp64 restatements of
# nn.Conv2d(k=3, padding=1, stride 1 or 2) and nn.ConvTranspose2d(k = stride = u) on channels-first arrays, and the
# split-bf16 arithmetic the device kernels compute fp32 products with (csrc/glx_bf16x3.h), restated on the host.
def conv2d_3x3(x, w, stride=1):
    """x (B, Cin, H, W), w (Cout, Cin, 3, 3), zero padding 1 -> (B, Cout, H', W') in fp64 (torch.nn.functional.conv2d's
    definition: y[b, o, i, j] = sum_{c, p, q} x[b, c, s i + p - 1, s j + q - 1] w[o, c, p, q])."""
    x, w = np.asarray(x, np.float64), np.asarray(w, np.float64)
    b, cin, h, wd = x.shape
    ho, wo = (h + 2 - 3) // stride + 1, (wd + 2 - 3) // stride + 1
    xp = np.zeros((b, cin, h + 2, wd + 2))
    xp[:, :, 1:-1, 1:-1] = x
    y = np.zeros((b, w.shape[0], ho, wo))
    for p in range(3):
        for q in range(3):
            patch = xp[:, :, p:p + stride * (ho - 1) + 1:stride, q:q + stride * (wo - 1) + 1:stride]
            y += np.einsum("bchw,oc->bohw", patch, w[:, :, p, q])
    return y


def conv_transpose2d(x, w, u):
    """x (B, Cin, H, W), w (Cin, Cout, u, u), stride u, no padding -> (B, Cout, u H, u W) in fp64:
    y[b, o, u i + p, u j + q] = sum_c x[b, c, i, j] w[c, o, p, q]."""
    x, w = np.asarray(x, np.float64), np.asarray(w, np.float64)
    b, cin, h, wd = x.shape
    y = np.zeros((b, w.shape[1], u * h, u * wd))
    for p in range(u):
        for q in range(u):
            y[:, :, p::u, q::u] = np.einsum("bchw,co->bohw", x, w[:, :, p, q])
    return y


def bf16_round(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32 (v_cvt_pk_bf16_f32's rounding)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def bf16x3_split(x):
    """The three bf16 pieces of an fp32 array (cv_split in csrc/glx_bf16x3.h): x1 = bf16(x), x2 = bf16(x - x1),
    x3 = bf16(x - x1 - x2), each returned as float32."""
    x = np.asarray(x, np.float32)
    a = bf16_round(x)
    r = (x - a).astype(np.float32)
    b = bf16_round(r)
    c = bf16_round((r - b).astype(np.float32))
    return a, b, c


def bf16x3_product(x, w):
    """x * w as the device computes it: the six piece products with i + j <= 4, each exact, summed in fp64 here (the
    device accumulates them in fp32 inside the matrix pipe)."""
    xs, ws = bf16x3_split(x), bf16x3_split(w)
    terms = ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0))
    return sum(ws[i].astype(np.float64) * xs[j].astype(np.float64) for i, j in terms)


def f16x2_block_exponent(m):
    """cv_block_exponent of csrc/glx_bf16x3.h: the power of two that puts a block's maximum |value| m into [2^14, 2^15)
    (127 for a block of zeros: 'no value yet'), clamped to +-110."""
    m = np.float32(m)
    if not m > 0:
        return 127
    field = int((np.array([m], np.float32).view(np.uint32)[0] >> 23) & 0xFF)        # the biased exponent, as the kernel reads it
    return max(-110, min(110, 14 - field + 127))


def f16x2_split(x, e):
    """The two fp16 pieces of x * 2^e (cv_split2 in csrc/glx_bf16x3.h; the scaling is exact): a = fp16(xs), b = fp16(xs - a),
    each returned as float32 in the SCALED domain."""
    xs = np.ldexp(np.asarray(x, np.float32), int(e)).astype(np.float32)
    a = xs.astype(np.float16).astype(np.float32)
    b = (xs - a).astype(np.float32).astype(np.float16).astype(np.float32)
    return a, b


def f16x2_product(x, w, ex, ew):
    """x * w as the f16x2 form of csrc/glx_conv2d.hip computes it: operands scaled by 2^ex / 2^ew, the three piece products
    b_w a_x, a_w b_x, a_w a_x (each exact in fp32), summed in fp64 here, the exponents taken out again."""
    ax, bx = f16x2_split(x, ex)
    aw, bw = f16x2_split(w, ew)
    s = bw.astype(np.float64) * ax + aw.astype(np.float64) * bx + aw.astype(np.float64) * ax
    return np.ldexp(s, -(int(ex) + int(ew)))


def sconv_forward_f16x2(features, weights, rules, bias=None):
    """sconv_forward with the products formed the way csrc/glx_sconv.hip's f16x2 block kernel forms them (k_sconv_gemm<..., F16>;
    reference semantics: spconv's SubMConv3d / SparseConv3d, spconv_backbone.py:30-75): the whole filter scaled by ONE power of two
    (max |w| into [2^14, 2^15)), every gathered input row by its OWN, two fp16 pieces per operand, the three piece products
    b_w a_x + a_w b_x + a_w a_x -- exact here (fp64 sums), where the kernel adds them in fp32.  Pure numpy: small cases only."""
    f = np.asarray(features, np.float32)
    w = np.asarray(weights, np.float32)
    K, cin, cout = w.shape
    ew = f16x2_block_exponent(np.abs(w).max())
    ew = 0 if ew == 127 else ew
    aw, bw = f16x2_split(w, ew)
    aw, bw = aw.astype(np.float64), bw.astype(np.float64)
    n_out = len(rules.out_indices)
    rows_e = np.array([f16x2_block_exponent(m) for m in np.abs(f).max(axis=1)], np.int64) if len(f) else np.zeros(0, np.int64)
    rows_e[rows_e == 127] = 0
    fs = np.ldexp(f, rows_e[:, None].astype(np.int32)).astype(np.float32)           # exact: a power of two
    ax = fs.astype(np.float16).astype(np.float32)
    bx = (fs - ax).astype(np.float32).astype(np.float16).astype(np.float64)
    ax = ax.astype(np.float64)
    out = np.zeros((n_out, cout), np.float64)
    for k in range(K):
        n = int(rules.n_pairs[k])
        i, o = rules.pairs_in[k, :n], rules.pairs_out[k, :n]
        s = ax[i] @ bw[k] + bx[i] @ aw[k] + ax[i] @ aw[k]
        np.add.at(out, o, np.ldexp(s, (-(rows_e[i] + ew))[:, None].astype(np.int32)))
    if bias is not None:
        out += np.asarray(bias, np.float64)
    return out
