"""RoI targets of the second stage on the device: our counterpart of ProposalTargetLayer
(pcdet/models/roi_heads/target_assigner/proposal_target_layer.py:8-239).  Matching (3-D IoU against the
frame's ground truths, per class when SAMPLE_ROI_BY_EACH_CLASS) and the foreground / hard / easy
background sampling run for the whole batch in two launches (csrc/glx_iou_nms.hip, glx_roi_targets)
without a host round trip; the reference loops over frames and classes in Python, draws on the host
(np.random.permutation / torch.randint on data-dependent sizes) and calls nonzero() per category.

The random draws enter as uniform numbers (`key`: order of the foreground RoIs, `pick`: draws with
replacement), by default from torch.rand on the device -- so the sampling distribution is the
reference's, and a test can replay the reference's own draws."""
import numpy as np
import torch

from . import _lib
from ._lib import call


def _get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


FUSED_GATHER = True      # False: the tensor-op statements of the reference's forward (tests compare the two)


class ProposalTargetLayer(torch.nn.Module):
    def __init__(self, roi_sampler_cfg):
        super().__init__()
        self.roi_sampler_cfg = roi_sampler_cfg

    def match_and_sample(self, rois, roi_labels, gt_boxes, key=None, pick=None):
        """rois (B,R,7+C), roi_labels (B,R) int64, gt_boxes (B,G,7+C+1) -> (max_overlaps (B,R),
        gt_assignment (B,R) int32, sampled (B,P) int32 RoI indices, sampled_gt (B,P) int32 ground-truth
        index or -1 for a frame without ground truth)."""
        cfg = self.roi_sampler_cfg
        rois = rois.contiguous().float()
        gt_boxes = gt_boxes.contiguous().float()
        roi_labels = roi_labels.contiguous().long()
        _lib.check_cuda(rois, roi_labels, gt_boxes)
        B, R, ld = rois.shape
        G, gld = gt_boxes.shape[1:]
        P = int(_get(cfg, "ROI_PER_IMAGE"))
        dev = rois.device
        if key is None:
            key = torch.rand((B, R), device=dev)
        if pick is None:
            pick = torch.rand((B, P), device=dev)
        fg_per_image = int(np.round(_get(cfg, "FG_RATIO") * P))                          # :127
        fg_thresh = min(_get(cfg, "REG_FG_THRESH"), _get(cfg, "CLS_FG_THRESH"))          # :128
        max_overlaps = torch.empty((B, R), dtype=torch.float32, device=dev)
        assignment = torch.empty((B, R), dtype=torch.int32, device=dev)
        n_gt = torch.empty((B,), dtype=torch.int32, device=dev)
        sampled = torch.zeros((B, P), dtype=torch.int32, device=dev)
        sampled_gt = torch.empty((B, P), dtype=torch.int32, device=dev)
        call("glx_roi_targets", rois, roi_labels, B, R, ld, gt_boxes, G, gld,
             1 if _get(cfg, "SAMPLE_ROI_BY_EACH_CLASS", False) else 0, key.contiguous().float(),
             pick.contiguous().float(), P, fg_per_image, float(fg_thresh), float(_get(cfg, "CLS_BG_THRESH_LO")),
             float(_get(cfg, "REG_FG_THRESH")), _lib.double_arg(float(_get(cfg, "HARD_BG_RATIO"))),
             max_overlaps, assignment, n_gt, sampled, sampled_gt)
        return max_overlaps, assignment, sampled, sampled_gt

    def _gather_fused(self, batch_dict, max_overlaps, sampled, sampled_gt):
        """Everything behind the sampling in one launch (csrc/glx_iou_nms.hip: k_roi_target_gather)."""
        cfg = self.roi_sampler_cfg
        rois, gt = batch_dict["rois"].contiguous().float(), batch_dict["gt_boxes"].contiguous().float()
        labels = batch_dict["roi_labels"].contiguous().long()
        scores = batch_dict["roi_scores"].contiguous().float() if "roi_scores" in batch_dict else None
        unc = batch_dict["gt_uncertaintys"].contiguous().float() if "gt_uncertaintys" in batch_dict else None
        B, R, ld = rois.shape
        G, gld = gt.shape[1:]
        P = sampled.shape[1]
        dev = rois.device
        kind = 0 if _get(cfg, "CLS_SCORE_TYPE") == "cls" else 1
        bg_t, fg_t = float(_get(cfg, "CLS_BG_THRESH")), float(_get(cfg, "CLS_FG_THRESH"))
        o_rois = torch.empty((B, P, ld), dtype=torch.float32, device=dev)
        o_gt = torch.empty((B, P, gld), dtype=torch.float32, device=dev)
        o_iou = torch.empty((B, P), dtype=torch.float32, device=dev)
        o_scores = torch.empty((B, P), dtype=torch.float32, device=dev) if scores is not None else None
        o_labels = torch.empty((B, P), dtype=torch.int64, device=dev)
        o_unc = torch.empty((B, P, unc.shape[-1]), dtype=torch.float32, device=dev) if unc is not None else None
        o_valid = torch.empty((B, P), dtype=torch.int64, device=dev)
        o_cls = torch.empty((B, P), dtype=torch.int64 if kind == 0 else torch.float32, device=dev)
        inv_span = float(np.float32(1.0) / np.float32(fg_t - bg_t)) if fg_t != bg_t else 0.0
        call("glx_roi_target_gather", rois, labels, scores, B, R, ld, gt, G, gld, unc,
             unc.shape[-1] if unc is not None else 0, max_overlaps, sampled, sampled_gt, P,
             float(_get(cfg, "REG_FG_THRESH")), fg_t, bg_t, inv_span, kind, o_rois, o_gt, o_iou, o_scores, o_labels,
             o_unc, o_valid, o_cls)
        out = {"rois": o_rois, "gt_of_rois": o_gt, "gt_iou_of_rois": o_iou, "roi_labels": o_labels,
               "reg_valid_mask": o_valid, "rcnn_cls_labels": o_cls, "gt_uncertaintys_of_rois": o_unc}
        out["roi_scores"] = o_scores if o_scores is not None else None
        return out

    def forward(self, batch_dict, key=None, pick=None):
        """Same keys in and out as the reference's forward (:13-63)."""
        cfg = self.roi_sampler_cfg
        rois, gt_boxes = batch_dict["rois"], batch_dict["gt_boxes"]
        max_overlaps, _, sampled, sampled_gt = self.match_and_sample(rois, batch_dict["roi_labels"], gt_boxes,
                                                                     key, pick)
        if FUSED_GATHER and _get(cfg, "CLS_SCORE_TYPE") in ("cls", "roi_iou"):
            return self._gather_fused(batch_dict, max_overlaps, sampled, sampled_gt)
        s = sampled.long()
        g = sampled_gt.long()
        has_gt = (g >= 0).unsqueeze(-1)
        g = g.clamp(min=0)
        batch_rois = rois.gather(1, s.unsqueeze(-1).expand(-1, -1, rois.shape[-1]))
        batch_gt_of_rois = torch.where(has_gt, gt_boxes.gather(1, g.unsqueeze(-1).expand(-1, -1, gt_boxes.shape[-1])),
                                       gt_boxes.new_zeros(()))
        batch_roi_ious = max_overlaps.gather(1, s)
        batch_roi_scores = batch_dict["roi_scores"].gather(1, s)
        batch_roi_labels = batch_dict["roi_labels"].gather(1, s)
        unc_of_rois = None
        if "gt_uncertaintys" in batch_dict:
            unc = batch_dict["gt_uncertaintys"]
            unc_of_rois = torch.where(has_gt, unc.gather(1, g.unsqueeze(-1).expand(-1, -1, unc.shape[-1])),
                                      unc.new_zeros(()))
        reg_valid_mask = (batch_roi_ious > _get(cfg, "REG_FG_THRESH")).long()            # :37
        bg_t, fg_t = _get(cfg, "CLS_BG_THRESH"), _get(cfg, "CLS_FG_THRESH")
        kind = _get(cfg, "CLS_SCORE_TYPE")
        if kind == "cls":                                                                # :40-44
            labels = (batch_roi_ious > fg_t).long()
            labels = torch.where((batch_roi_ious > bg_t) & (batch_roi_ious < fg_t), labels.new_full((), -1), labels)
        elif kind == "roi_iou":                                                          # :45-54
            fg_mask = batch_roi_ious > fg_t
            bg_mask = batch_roi_ious < bg_t
            labels = torch.where(fg_mask | bg_mask, fg_mask.float(), (batch_roi_ious - bg_t) / (fg_t - bg_t))
        else:
            raise NotImplementedError
        return {"rois": batch_rois, "gt_of_rois": batch_gt_of_rois, "gt_iou_of_rois": batch_roi_ious,
                "roi_scores": batch_roi_scores, "roi_labels": batch_roi_labels, "reg_valid_mask": reg_valid_mask,
                "rcnn_cls_labels": labels, "gt_uncertaintys_of_rois": unc_of_rois}
