"""Harness of the two-stage detector data flow (GLENet-VR = Voxel-RCNN, BASELINE config 3), our
counterpart of the Python callers listed in SURVEY.md section 8c.  It exists to drive the hot-path
kernels in the order and with the shapes the reference drives them; it is not a training system.

  generate_anchors      AnchorGenerator.generate_anchors   (dense_heads/target_assigner/anchor_generator.py:17-61)
  decode_boxes          ResidualCoder.decode_torch         (utils/box_coder_utils.py:46-77)
  predicted_boxes       AnchorHeadTemplate.generate_predicted_boxes (dense_heads/anchor_head_template.py:233-279)
  proposal_layer        RoIHeadTemplate.proposal_layer + class_agnostic_nms
                        (roi_heads/roi_head_template.py:52-128, model_utils/model_nms_utils.py:6-62)
  refine_boxes          RoIHeadTemplate.generate_predicted_boxes (roi_head_template.py:288-316)
  VoxelRCNNFlow         Detector3DTemplate module order for GLENet_VR.yaml: MeanVFE -> VoxelBackBone8x
                        -> HeightCompression -> BaseBEVBackbone -> AnchorHeadSingle -> VoxelRCNNHead
"""
import os

import numpy as np
import torch
from torch import nn

from . import backbone as gb
from . import dense_path as dp
from . import roi_grid as rg
from .pcdet_ops.iou3d_nms import iou3d_nms_utils


def generate_anchors(anchor_range, grid_size_xy, anchor_sizes, anchor_rotations, anchor_bottom_heights,
                     align_center=False, device="cpu"):
    """One anchor set -> (nz, ny, nx, n_size, n_rot, 7) [x,y,z,dx,dy,dz,ry], z lifted to the box
    centre; x/y positions cover the range end to end ((grid-1) intervals) unless align_center."""
    gx, gy = grid_size_xy
    r = anchor_range
    if align_center:
        sx, sy = (r[3] - r[0]) / gx, (r[4] - r[1]) / gy
        ox, oy = sx / 2, sy / 2
    else:
        sx, sy = (r[3] - r[0]) / (gx - 1), (r[4] - r[1]) / (gy - 1)
        ox = oy = 0
    xs = torch.arange(r[0] + ox, r[3] + 1e-5, step=sx, dtype=torch.float32, device=device)
    ys = torch.arange(r[1] + oy, r[4] + 1e-5, step=sy, dtype=torch.float32, device=device)
    zs = torch.tensor(anchor_bottom_heights, dtype=torch.float32, device=device)
    sizes = torch.tensor(anchor_sizes, dtype=torch.float32, device=device)
    rots = torch.tensor(anchor_rotations, dtype=torch.float32, device=device)
    nz, ny, nx, ns, nr = len(zs), len(ys), len(xs), len(sizes), len(rots)
    a = torch.empty((nz, ny, nx, ns, nr, 7), dtype=torch.float32, device=device)
    a[..., 0] = xs.view(1, 1, nx, 1, 1)
    a[..., 1] = ys.view(1, ny, 1, 1, 1)
    a[..., 2] = zs.view(nz, 1, 1, 1, 1)
    a[..., 3:6] = sizes.view(1, 1, 1, ns, 1, 3)
    a[..., 6] = rots.view(1, 1, 1, 1, nr)
    a[..., 2] += a[..., 5] / 2
    return a


def decode_boxes(enc, anchors):
    """Residual box code -> boxes: xy scaled by the anchor's BEV diagonal, z by its height, sizes
    exponential, heading additive; extra code channels additive."""
    xa, ya, za, dxa, dya, dza, ra = [anchors[..., i:i + 1] for i in range(7)]
    xt, yt, zt, dxt, dyt, dzt, rt = [enc[..., i:i + 1] for i in range(7)]
    diag = torch.sqrt(dxa ** 2 + dya ** 2)
    out = [xt * diag + xa, yt * diag + ya, zt * dza + za, torch.exp(dxt) * dxa, torch.exp(dyt) * dya,
           torch.exp(dzt) * dza, rt + ra]
    if enc.shape[-1] > 7:
        out.append(enc[..., 7:] + anchors[..., 7:])
    return torch.cat(out, dim=-1)


FUSED_PREDICTED_BOXES = True


def predicted_boxes(cls_preds, box_preds, dir_cls_preds, anchors, dir_offset=0.78539, dir_limit_offset=0.0,
                    num_dir_bins=2):
    """Head maps (B,H,W,A*c) + anchors (.., 7) -> batch_cls_preds (B,N,cls), batch_box_preds (B,N,7)."""
    B = cls_preds.shape[0]
    anc = anchors.reshape(1, -1, anchors.shape[-1])
    n = anc.shape[1]
    if (FUSED_PREDICTED_BOXES and box_preds.is_cuda and not torch.is_grad_enabled() and anc.shape[-1] == 7
            and box_preds.dtype == torch.float32 and box_preds.numel() == B * n * 7):
        # one launch instead of ~30 elementwise ones, same rounding (csrc/glx_loss.hip: k_predicted_boxes)
        import ctypes
        from . import _lib
        bp = box_preds.reshape(B, n, 7).contiguous()
        dirp = dir_cls_preds.reshape(B, n, -1).contiguous().float() if dir_cls_preds is not None else None
        boxes = torch.empty((B, n, 7), dtype=torch.float32, device=bp.device)
        _lib.call("glx_predicted_boxes", bp, dirp, anc.reshape(n, 7).contiguous().float(), B, n,
                  num_dir_bins if dirp is None else dirp.shape[-1], ctypes.c_float(dir_offset),
                  ctypes.c_float(dir_limit_offset), boxes)
        return cls_preds.reshape(B, n, -1).float(), boxes
    boxes = decode_boxes(box_preds.reshape(B, n, -1), anc.expand(B, n, anc.shape[-1]))
    if dir_cls_preds is not None:
        labels = dir_cls_preds.reshape(B, n, -1).max(dim=-1)[1]
        period = 2 * np.pi / num_dir_bins
        rot = dp.limit_period(boxes[..., 6] - dir_offset, dir_limit_offset, period)
        boxes[..., 6] = rot + dir_offset + period * labels.to(boxes.dtype)
    return cls_preds.reshape(B, n, -1).float(), boxes


def proposal_layer(batch_box_preds, batch_cls_preds, nms_pre_maxsize, nms_post_maxsize, nms_thresh,
                   normalized=False):
    """Per frame: class-max score, top-k, rotated NMS (device-side keep list), zero padding to
    nms_post_maxsize.  Returns rois (B,P,7+), roi_scores (B,P), roi_labels (B,P) 1-based."""
    B = batch_box_preds.shape[0]
    if not normalized:
        batch_cls_preds = torch.sigmoid(batch_cls_preds)
    if BATCHED_PROPOSALS and batch_box_preds.is_cuda and not torch.is_grad_enabled():
        return _proposal_layer_batched(batch_box_preds, batch_cls_preds, nms_pre_maxsize,
                                       nms_post_maxsize, nms_thresh)
    rois = batch_box_preds.new_zeros((B, nms_post_maxsize, batch_box_preds.shape[-1]))
    scores = batch_box_preds.new_zeros((B, nms_post_maxsize))
    labels = batch_box_preds.new_zeros((B, nms_post_maxsize), dtype=torch.long)
    for b in range(B):
        s, lab = batch_cls_preds[b].max(dim=1)
        top, order = torch.topk(s, k=min(nms_pre_maxsize, s.shape[0]))
        cand = batch_box_preds[b][order]
        keep, _ = iou3d_nms_utils.nms_gpu(cand[:, 0:7], top, nms_thresh)
        sel = order[keep[:nms_post_maxsize]]
        k = sel.shape[0]
        rois[b, :k], scores[b, :k], labels[b, :k] = batch_box_preds[b][sel], s[sel], lab[sel]
    return rois, scores, labels + 1


BATCHED_PROPOSALS = True


FUSED_TOPK = True


def topk_desc(scores, k):
    """torch.topk(scores, k, dim=1) of (B, A) scores, sorted: one launch per call on the device (csrc/glx_iou_nms.hip:
    k_topk_desc -- radix select + stable radix sort in LDS, one block per frame; equal scores by ascending index)
    where torch runs ~45 launches; torch itself for host tensors or k beyond the kernel's LDS budget."""
    from . import _lib
    if not (FUSED_TOPK and scores.is_cuda and scores.dim() == 2 and scores.dtype == torch.float32
            and 0 < k <= min(scores.shape[1], _lib.query("glx_topk_max_k"))):
        return torch.topk(scores, k=k, dim=1)
    s = scores.contiguous()
    top = torch.empty((s.shape[0], k), dtype=torch.float32, device=s.device)
    order = torch.empty((s.shape[0], k), dtype=torch.int64, device=s.device)
    if MULTI_BLOCK_TOPK and s.shape[0] <= 16 and 4096 <= s.shape[1] <= 131072:
        # 32 cooperating blocks per frame + a per-frame sort; the zero-initialised workspace belongs to (device, stream)
        key = (s.device.index, torch.cuda.current_stream(s.device).cuda_stream, s.shape[0], k)
        ws = _TOPK_WS.get(key)
        if ws is None and torch.cuda.is_current_stream_capturing():
            ws = False       # never allocate a cached buffer inside a capture (it would live in the graph's private pool)
        elif ws is None:
            ws = _TOPK_WS[key] = torch.zeros(_lib.query("glx_topk_workspace_bytes", s.shape[0], k), dtype=torch.uint8,
                                             device=s.device)
        if ws is False:
            _lib.call("glx_topk_desc", s, s.shape[0], s.shape[1], k, top, order)
            return top, order
        _lib.call("glx_topk_desc_ws", s, s.shape[0], s.shape[1], k, top, order, ws, _lib.size_arg(ws.numel()))
    else:
        _lib.call("glx_topk_desc", s, s.shape[0], s.shape[1], k, top, order)
    return top, order


MULTI_BLOCK_TOPK = True
_TOPK_WS = {}


_ZEROS_I64 = {}


def _zeros_i64(b, a, device):
    """A cached, never written (b, a) int64 zero tensor (the class index of a one-class head)."""
    key = (b, a, device.type, device.index)
    if key not in _ZEROS_I64:
        _ZEROS_I64[key] = torch.zeros((b, a), dtype=torch.int64, device=device)
    return _ZEROS_I64[key]


def _proposal_layer_batched(batch_box_preds, scores_all, nms_pre_maxsize, nms_post_maxsize, nms_thresh):
    """Same result as the per-frame loop above without its host round trips: batched class-max and
    top-k (already in descending order, so the NMS wrapper's own sort is the identity), ONE batched
    NMS launch sequence whose sweeps stop at nms_post_maxsize survivors, and a masked gather in
    place of the variable-length slice."""
    from .pcdet_ops.iou3d_nms import iou3d_nms_cuda
    B, A, C = batch_box_preds.shape
    if scores_all.shape[2] == 1:             # one class: the class maximum is the score itself, its arg-max 0 (no launches)
        s, lab = scores_all.reshape(B, A), _zeros_i64(B, A, scores_all.device)
    else:
        s, lab = scores_all.max(dim=2)                                          # (B, A)
    k = min(nms_pre_maxsize, A)
    top, order = topk_desc(s, k)
    cand = torch.gather(batch_box_preds, 1, order.unsqueeze(-1).expand(B, k, C))
    keep, num = iou3d_nms_cuda.nms_device_batch(cand[..., 0:7].contiguous(), nms_thresh,
                                                max_keep=nms_post_maxsize)
    p = nms_post_maxsize
    if (cand.dtype == torch.float32 and top.dtype == torch.float32 and lab.dtype == torch.int64
            and order.dtype == torch.int64 and num.dtype == torch.int32):
        from . import _lib
        cand, top, lab, order = cand.contiguous(), top.contiguous(), lab.contiguous(), order.contiguous()
        rois = torch.empty((B, p, C), dtype=torch.float32, device=cand.device)
        scores = torch.empty((B, p), dtype=torch.float32, device=cand.device)
        labels = torch.empty((B, p), dtype=torch.int64, device=cand.device)
        _lib.call("glx_gather_proposals", cand, top, lab, order, keep, num, B, A, k, keep.shape[1], p, C,
                  rois, scores, labels)
        return rois, scores, labels
    if keep.shape[1] < p:
        keep = torch.nn.functional.pad(keep, (0, p - keep.shape[1]))
    slot = torch.arange(p, device=keep.device)
    valid = slot[None, :] < num[:, None]                                        # (B, P)
    sel = torch.where(valid, keep[:, :p], torch.zeros_like(keep[:, :p]))        # into the top-k list
    rois = torch.gather(cand, 1, sel.unsqueeze(-1).expand(B, p, C)) * valid.unsqueeze(-1)
    scores = torch.gather(top, 1, sel) * valid
    labels = torch.gather(torch.gather(lab, 1, order), 1, sel) * valid
    return rois, scores, labels + 1


def refine_boxes(rois, box_preds):
    """RoI-frame residuals -> boxes in the LiDAR frame: decode against the RoI moved to the origin,
    rotate by the RoI heading, translate by its centre."""
    B, R = rois.shape[0], rois.shape[1]
    local = rois.clone().detach()
    local[..., 0:3] = 0
    boxes = decode_boxes(box_preds.view(B, R, -1), local).view(B * R, -1)
    boxes = rg.rotate_points_along_z(boxes.unsqueeze(1), rois[..., 6].reshape(-1)).squeeze(1)
    boxes[:, 0:3] += rois[..., 0:3].reshape(-1, 3)
    return boxes.view(B, R, -1)


# GLENet_VR.yaml:168-181 (GLENet_S / GLENet_C use the same keys with their own values)
POST_PROCESSING_CFG = dict(SCORE_THRESH=0.3, POST_SCORE_THRESH=0.81, NMS_THRESH=0.1, NMS_PRE_MAXSIZE=4096,
                           NMS_POST_MAXSIZE=500)


def post_processing(batch_cls_preds, batch_box_preds, batch_box_std_preds=None, roi_labels=None, cfg=None,
                    normalized=False):
    """Detector3DTemplate.post_processing with NMS_TYPE new_nms_gpu (pcdet/models/detectors/detector3d_template.py:179-317
    -> model_nms_utils.class_agnostic_nms :6-62 -> iou3d_nms_utils.new_nms_gpu / nms_func :200-273) for the whole batch on
    the device, shape-static, without a host loop or read-back: sigmoid + class max, SCORE_THRESH mask, top-k
    (NMS_PRE_MAXSIZE), heading wrap, variance = exp(log-variance), the greedy variance-voting NMS (every frame its own
    block), NMS_POST_MAXSIZE, POST_SCORE_THRESH.  The reference runs this per frame on the host in numpy.

    batch_cls_preds (B,R,C) logits (or scores when `normalized`), batch_box_preds (B,R,7+), batch_box_std_preds (B,R,7)
    log-variances or None (plain greedy NMS with the >= strictness of nms_func), roi_labels (B,R) 1-based or None
    (-> class arg-max + 1).  Returns pred_boxes (B,P,7), pred_scores (B,P), pred_labels (B,P), pred_index (B,P) source
    box (-1 = padding) and num (B,) int32 on the device, P = NMS_POST_MAXSIZE; rows past num[b] are zero.
    Equal scores: lower index first (torch.topk leaves that order unspecified in the reference)."""
    from . import _lib
    cfg = dict(POST_PROCESSING_CFG, **(cfg or {}))
    B, R, C = batch_cls_preds.shape
    _lib.check_cuda(batch_cls_preds, batch_box_preds, batch_box_std_preds, roi_labels)
    scores_all = batch_cls_preds.float() if normalized else torch.sigmoid(batch_cls_preds.float())
    scores, arg = scores_all.max(dim=-1)                                         # (B, R)
    labels = roi_labels.long().contiguous() if roi_labels is not None else (arg + 1).contiguous()
    thr = cfg.get("SCORE_THRESH")
    if thr is not None:
        passed = scores >= thr
        counts = passed.sum(dim=1, dtype=torch.int32)
        masked = torch.where(passed, scores, scores.new_full((), -1.0))
    else:
        counts = torch.full((B,), R, dtype=torch.int32, device=scores.device)
        masked = scores
    K = min(int(cfg["NMS_PRE_MAXSIZE"]), R)
    top, order = topk_desc(masked.contiguous(), K)
    counts = counts.clamp(max=K).contiguous()
    boxes = batch_box_preds.float().contiguous()
    std = batch_box_std_preds.float().contiguous() if batch_box_std_preds is not None else None
    dev = boxes.device
    cand = torch.empty((B, K, 7), dtype=torch.float32, device=dev)
    var = torch.empty((B, K, 7), dtype=torch.float32, device=dev) if std is not None else None
    _lib.call("glx_det_candidates", boxes, std, boxes.shape[-1], std.shape[-1] if std is not None else 0,
              order.contiguous(), counts, B, R, K, cand, var)
    ious_t = torch.empty((B, K, K), dtype=torch.float32, device=dev)
    _lib.call("glx_boxes_iou_bev_self_batch", cand, B, K, counts, 1, ious_t)
    new_scores = top.clone()
    scratch = torch.empty((B, K, 8), dtype=torch.float32, device=dev) if var is not None else None
    _lib.call("glx_nms_vote_batch", cand, new_scores, var, 7, ious_t, B, K, counts, float(cfg["NMS_THRESH"]), 0.0, scratch)
    P = int(cfg["NMS_POST_MAXSIZE"])
    out_boxes = torch.empty((B, P, 7), dtype=torch.float32, device=dev)
    out_scores = torch.empty((B, P), dtype=torch.float32, device=dev)
    out_labels = torch.empty((B, P), dtype=torch.int64, device=dev)
    out_index = torch.empty((B, P), dtype=torch.int64, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    post = cfg.get("POST_SCORE_THRESH")
    _lib.call("glx_det_gather", new_scores, top, cand, order, labels, counts, B, R, K, P,
              float(post) if post is not None else 0.0, 1 if post is not None else 0, out_boxes, out_scores, out_labels,
              out_index, num)
    return dict(pred_boxes=out_boxes, pred_scores=out_scores, pred_labels=out_labels, pred_index=out_index, num=num)


def pred_dicts(post):
    """The reference's list of per-frame dicts (detector3d_template.py:311-316) from post_processing's static output:
    ONE read-back of `num`, then slices."""
    n = post["num"].tolist()
    return [{"pred_boxes": post["pred_boxes"][b, :k], "pred_scores": post["pred_scores"][b, :k],
             "pred_labels": post["pred_labels"][b, :k]} for b, k in enumerate(n)]


class VoxelRCNNFlow(nn.Module):
    """GLENet-VR inference data flow on one batch of stacked device points."""

    POOL = {n: dict(mlps=[[32, 32]], query_ranges=[[4, 4, 4]], radii=[r], nsamples=[16])
            for n, r in (("x_conv2", 0.4), ("x_conv3", 0.8), ("x_conv4", 1.6))}      # GLENet_VR.yaml:117-139

    def __init__(self, cfg, num_point_features=4, nms_pre=2048, nms_post=100, nms_thresh=0.7):
        super().__init__()
        self.cfg = cfg
        self.nms = (nms_pre, nms_post, nms_thresh)                                   # GLENet_VR.yaml:108-115
        grid = gb.gv.grid_size_of(cfg["point_cloud_range"], cfg["voxel_size"])
        self.vfe = gb.MeanVFE()
        self.backbone_3d = gb.VoxelBackBone8x(num_point_features, grid)
        self.map_to_bev = gb.HeightCompression()
        # channels-last BEV map, left to the BEV backbone (first layer on the sparse tensor, dense_path.BEVBackbone)
        self.map_to_bev.channels_last = True
        self.map_to_bev.defer = dp.SPARSE_FIRST_BEV_LAYER
        self.backbone_2d = dp.BEVBackbone(256).to(memory_format=torch.channels_last)
        if self.map_to_bev.defer:
            d = 256 // self.backbone_3d.num_point_features
            self.backbone_3d.extra_plan = (gb.spconv.core.PlannedConv(dp.BEV_FIRST_KEY, (d, 3, 3), (d, 1, 1), (0, 1, 1),
                                                                       self.backbone_3d.num_point_features, 64),)
        self.dense_head = dp.AnchorHead(self.backbone_2d.num_bev_features, num_class=1,
                                        num_anchors_per_location=2)
        self.roi_pool = rg.RoIGridPool(self.backbone_3d.backbone_channels, self.POOL, 6, cfg["voxel_size"],
                                       cfg["point_cloud_range"])
        self.roi_fc = dp.RoIFCStack(self.roi_pool.num_features, 6)
        self.feature_map = (grid[0] // 8, grid[1] // 8)
        self._anchors = None

    def anchors(self, device):
        if self._anchors is None or self._anchors.device != device:
            self._anchors = generate_anchors(self.cfg["point_cloud_range"], self.feature_map, [[3.9, 1.6, 1.56]],
                                             [0, 1.57], [-1.78], device=device)           # GLENet_VR.yaml:66-77
        return self._anchors

    def forward(self, points, batch_idx, batch_size):
        bd = gb.voxelize_batch(points, batch_idx, batch_size, self.cfg, train=False)
        bd = self.map_to_bev(self.backbone_3d(self.vfe(bd)))
        return self.second_stage(bd, batch_size)

    def second_stage(self, bd, batch_size):
        """Everything after map_to_bev: BEV backbone + anchor head, proposals, RoI-grid pooling,
        FC refinement.  Free of host synchronisation in eval mode on the device."""
        mark = getattr(self, "mark", None) or (lambda name: None)      # optional callable(stage_name), as in GLENetVR
        bd = self.backbone_2d(bd)
        mark("BEV backbone")
        bd = self.dense_head(bd)
        mark("anchor head")
        cls, boxes = predicted_boxes(bd["cls_preds"], bd["box_preds"], bd.get("dir_cls_preds"),
                                     self.anchors(bd["spatial_features_2d"].device))
        rois, roi_scores, roi_labels = proposal_layer(boxes, cls, *self.nms)
        mark("decode + top-k + NMS")
        pooled = self.roi_pool(rois, bd["multi_scale_3d_features"], bd["multi_scale_3d_strides"], batch_size)
        mark("RoI-grid pooling")
        rcnn_cls, rcnn_reg = self.roi_fc(pooled)
        mark("FC towers")
        bd.update(rois=rois, roi_scores=roi_scores, roi_labels=roi_labels,
                  batch_cls_preds=rcnn_cls.view(batch_size, -1, rcnn_cls.shape[-1]),
                  batch_box_preds=refine_boxes(rois, rcnn_reg))
        return bd


class StaticDetectorPipeline(gb.StaticFramePipeline):
    """The whole two-stage inference flow as one shape-static launch sequence / HIP graph: the
    sparse-backbone pipeline of glenet_amd.backbone.StaticFramePipeline followed by
    VoxelRCNNFlow.second_stage.  Proposals are padded to NMS_POST_MAXSIZE by construction and the
    RoI-grid kernels take the sparse tensors at their capacity (only live rows are reachable through
    the cell index), so no stage needs a row count on the host."""

    def __init__(self, flow, batch_size, num_points, num_features=4, capacities=None, device=None):
        super().__init__(flow.backbone_3d, flow.cfg, batch_size, num_points, num_features,
                         train_voxel_cap=False, capacities=capacities, device=device)
        self.flow = flow
        self.post_cfg = None          # Detector3DTemplate.post_processing's settings (None: POST_PROCESSING_CFG, GLENet_VR.yaml:168-181)
        # the flow's own HeightCompression (channels-last, deferred to the BEV backbone's sparse first layer): with the
        # base class's default one the BEV backbone received an NCHW map and ran on the vendor's kernels
        self.hc = flow.map_to_bev

    def _tagged_modules(self):
        return (self.flow,)

    def enqueue(self):
        from ._lib import workspace
        bd = super().enqueue()
        with torch.no_grad(), workspace.scoped(id(self)):
            bd = self.flow.second_stage(bd, self.B, self.post_cfg) if self.post_cfg is not None else self.flow.second_stage(bd, self.B)
        self.out = bd
        return bd

    def run_checked(self, points, batch_idx):
        """load + replay + capacity verdict; recomputed on the exact-shape flow when a capacity was
        exceeded (one host synchronisation per batch)."""
        self.load(points, batch_idx)
        out = self.replay() if self.graph is not None else self.enqueue()
        torch.cuda.current_stream(self.points.device).synchronize()
        try:
            self.check()
            return out
        except RuntimeError:
            with torch.no_grad():
                return self.flow(points, batch_idx, self.B)
