"""Anchor target assignment of the dense head on the device: our counterpart of
AxisAlignedTargetAssigner.assign_targets (pcdet/models/dense_heads/target_assigner/
axis_aligned_target_assigner.py:36-213) for the configuration the GLENet models use -- nearest-BEV
IoU matching (MATCH_HEIGHT False), no sampling (POS_FRACTION -1), one shared head (no multihead).
Three launches per anchor class for the whole batch (csrc/glx_loss.hip, glx_assign_targets) and no
host synchronisation; the reference loops over frames and classes in Python with ~70 tensor
kernels and several `.nonzero()` round trips each."""
import torch

from . import _lib
from ._lib import call, query, size_arg, workspace


def assign_targets(all_anchors, gt_boxes_with_classes, anchor_class_ids, matched_thresholds,
                   unmatched_thresholds, norm_by_num_examples=False, gt_uncertaintys=None):
    """all_anchors: list (one per anchor class) of (nz, ny, nx, n_size, n_rot, 7) tensors;
    gt_boxes_with_classes (B, M, 8) zero-padded, last column = 1-based class id;
    anchor_class_ids[i] = class id of all_anchors[i].
    -> {'box_cls_labels' (B, A) int32, 'box_reg_targets' (B, A, 7), 'reg_weights' (B, A)} with the
    reference's anchor order (classes interleaved along the anchor-type axis).
    gt_uncertaintys (B, M, 7): the WeightedAxisAlignedTargetAssigner of GLENet-S / -C
    (weighted_axis_aligned_target_assigner.py:36-213) -- 'reg_weights' then holds, flattened to (B, A * 7) as
    upstream, the label uncertainty of the ground truth every positive anchor was matched to (zeros elsewhere), also
    returned as 'label_uncertainty' (B, A, 7)."""
    gt = gt_boxes_with_classes.contiguous().float()
    _lib.check_cuda(gt)
    B, M, C = gt.shape
    if gt_uncertaintys is not None:
        return _assign_weighted(all_anchors, gt, gt_uncertaintys.contiguous().float(), anchor_class_ids,
                                matched_thresholds, unmatched_thresholds)
    labels, targets, weights = [], [], []
    for anchors, cls, mt, ut in zip(all_anchors, anchor_class_ids, matched_thresholds, unmatched_thresholds):
        fmap = anchors.shape[:3]
        a = anchors.reshape(-1, anchors.shape[-1])[:, 0:7].contiguous().float()
        n = a.shape[0]
        lab = torch.empty((B, n), dtype=torch.int32, device=gt.device)
        tgt = torch.empty((B, n, 7), dtype=torch.float32, device=gt.device)
        w = torch.empty((B, n), dtype=torch.float32, device=gt.device)
        ws = workspace.get(query("glx_assign_targets_workspace_bytes", B, n), gt.device)
        call("glx_assign_targets", a, n, gt, B, M, C, int(cls), float(mt), float(ut),
             1 if norm_by_num_examples else 0, lab, tgt, w, ws, size_arg(ws.numel()))
        labels.append(lab.view(B, *fmap, -1))
        targets.append(tgt.view(B, *fmap, -1, 7))
        weights.append(w.view(B, *fmap, -1))
    if len(labels) == 1:          # one anchor class (GLENet-VR's Car model): nothing to interleave, no copies
        return {"box_cls_labels": labels[0].view(B, -1), "box_reg_targets": targets[0].view(B, -1, 7),
                "reg_weights": weights[0].view(B, -1)}
    return {"box_cls_labels": torch.cat(labels, dim=-1).view(B, -1),
            "box_reg_targets": torch.cat(targets, dim=-2).view(B, -1, 7),
            "reg_weights": torch.cat(weights, dim=-1).view(B, -1)}


def _assign_weighted(all_anchors, gt, unc, anchor_class_ids, matched_thresholds, unmatched_thresholds):
    B, M, C = gt.shape
    labels, targets, uncs = [], [], []
    for anchors, cls, mt, ut in zip(all_anchors, anchor_class_ids, matched_thresholds, unmatched_thresholds):
        fmap = anchors.shape[:3]
        a = anchors.reshape(-1, anchors.shape[-1])[:, 0:7].contiguous().float()
        n = a.shape[0]
        lab = torch.empty((B, n), dtype=torch.int32, device=gt.device)
        tgt = torch.empty((B, n, 7), dtype=torch.float32, device=gt.device)
        w = torch.empty((B, n), dtype=torch.float32, device=gt.device)
        ug = torch.empty((B, n), dtype=torch.int32, device=gt.device)
        ws = workspace.get(query("glx_assign_targets_workspace_bytes", B, n), gt.device)
        call("glx_assign_targets_ex", a, n, gt, B, M, C, int(cls), float(mt), float(ut), 0, lab, tgt, w, ug, ws,
             size_arg(ws.numel()))
        idx = ug.clamp(min=0).long().unsqueeze(-1).expand(-1, -1, 7)
        u = torch.gather(unc, 1, idx) * (ug >= 0).unsqueeze(-1).float()
        labels.append(lab.view(B, *fmap, -1))
        targets.append(tgt.view(B, *fmap, -1, 7))
        uncs.append(u.view(B, *fmap, -1, 7))
    lu = torch.cat(uncs, dim=-2).view(B, -1, 7) if len(uncs) > 1 else uncs[0].view(B, -1, 7)
    return {"box_cls_labels": (torch.cat(labels, dim=-1) if len(labels) > 1 else labels[0]).view(B, -1),
            "box_reg_targets": (torch.cat(targets, dim=-2) if len(targets) > 1 else targets[0]).view(B, -1, 7),
            "reg_weights": lu.reshape(B, -1), "label_uncertainty": lu}
