"""TEST INFRASTRUCTURE (oracle) -- numpy restatement of ProposalTargetLayer
(pcdet/models/roi_heads/target_assigner/proposal_target_layer.py:13-239), frame by frame like the
reference.  Pinned to tests/golden/roi_targets_ref.npz (the reference's own class run on CPU).  The
reference's host-side random draws enter as uniform numbers: `key` (B,R) orders the foreground RoIs
(ascending key = the permutation of :147-148), `pick` (B,P) draws with replacement, slot q taking
list[floor(pick[q] * len(list))] (torch.randint of :186-207, np.random.rand of :158-160)."""
import numpy as np

from . import boxes_iou3d


def max_iou(rois, roi_labels, gt, same_class):
    """:98-114 + get_max_iou_with_same_class :209-238 -> (max_overlaps (R,), gt_assignment (R,), n_gt)."""
    k = len(gt) - 1
    while k >= 0 and gt[k].sum() == 0:                                     # :99-101
        k -= 1
    cur = gt[:k + 1]
    if len(cur) == 0:                                                      # :103: one zero box
        return np.zeros(len(rois), np.float32), np.zeros(len(rois), np.int64), 0
    if not same_class:                                                     # :113-114
        iou = boxes_iou3d(rois[:, :7], cur[:, :7])
        return iou.max(1), iou.argmax(1), len(cur)
    mo = np.zeros(len(rois), np.float32)
    ga = np.zeros(len(rois), np.int64)
    gl = cur[:, -1].astype(np.int64)
    for c in range(int(gl.min()), int(gl.max()) + 1):                      # :225-236
        rm, gm = roi_labels == c, gl == c
        if rm.sum() > 0 and gm.sum() > 0:
            iou = boxes_iou3d(rois[rm][:, :7], cur[gm][:, :7])
            mo[rm] = iou.max(1)
            ga[rm] = np.nonzero(gm)[0][iou.argmax(1)]
    return mo, ga, len(cur)


def subsample(mo, cfg, key, pick):
    """subsample_rois + sample_bg_inds, :126-207."""
    P = cfg["ROI_PER_IMAGE"]
    fg_per_image = int(np.round(cfg["FG_RATIO"] * P))
    fg_thresh = min(cfg["REG_FG_THRESH"], cfg["CLS_FG_THRESH"])
    fg = np.nonzero(mo >= np.float32(fg_thresh))[0]
    easy = np.nonzero(mo < np.float32(cfg["CLS_BG_THRESH_LO"]))[0]
    hard = np.nonzero((mo < np.float32(cfg["REG_FG_THRESH"])) & (mo >= np.float32(cfg["CLS_BG_THRESH_LO"])))[0]

    def draw(lst, u):
        return lst[np.minimum((u.astype(np.float32) * np.float32(len(lst))).astype(np.int64), len(lst) - 1)]

    def sample_bg(n, u):
        if len(hard) and len(easy):
            nh = min(int(n * cfg["HARD_BG_RATIO"]), len(hard))
            return np.concatenate([draw(hard, u[:nh]), draw(easy, u[nh:n])])
        return draw(hard if len(hard) else easy, u[:n])

    nbg = len(hard) + len(easy)
    if len(fg) and nbg:
        take = min(fg_per_image, len(fg))
        order = fg[np.lexsort((fg, key[fg]))]
        return np.concatenate([order[:take], sample_bg(P - take, pick[take:])])
    if len(fg):
        return draw(fg, pick[:P])
    return sample_bg(P, pick)


def roi_targets(rois, roi_labels, roi_scores, gt_boxes, cfg, key, pick, gt_unc=None):
    B = len(rois)
    P = cfg["ROI_PER_IMAGE"]
    out = {"rois": np.zeros((B, P, rois.shape[-1]), np.float32), "gt_of_rois": np.zeros((B, P, gt_boxes.shape[-1]), np.float32),
           "gt_iou_of_rois": np.zeros((B, P), np.float32), "roi_scores": np.zeros((B, P), np.float32),
           "roi_labels": np.zeros((B, P), np.int64), "max_overlaps": np.zeros(rois.shape[:2], np.float32),
           "sampled": np.zeros((B, P), np.int64)}
    if gt_unc is not None:
        out["gt_uncertaintys_of_rois"] = np.zeros((B, P, gt_unc.shape[-1]), np.float32)
    for b in range(B):
        mo, ga, n = max_iou(rois[b], roi_labels[b], gt_boxes[b], cfg.get("SAMPLE_ROI_BY_EACH_CLASS", False))
        s = subsample(mo, cfg, key[b], pick[b])
        out["max_overlaps"][b], out["sampled"][b] = mo, s
        out["rois"][b], out["roi_labels"][b] = rois[b][s], roi_labels[b][s]                 # :118-122
        out["gt_iou_of_rois"][b], out["roi_scores"][b] = mo[s], roi_scores[b][s]
        if n:
            out["gt_of_rois"][b] = gt_boxes[b][ga[s]]
            if gt_unc is not None:
                out["gt_uncertaintys_of_rois"][b] = gt_unc[b][ga[s]]
    iou = out["gt_iou_of_rois"]
    out["reg_valid_mask"] = (iou > np.float32(cfg["REG_FG_THRESH"])).astype(np.int64)      # :37
    bg_t, fg_t = cfg["CLS_BG_THRESH"], cfg["CLS_FG_THRESH"]
    if cfg["CLS_SCORE_TYPE"] == "cls":                                                       # :40-44
        lab = (iou > np.float32(fg_t)).astype(np.int64)
        lab[(iou > np.float32(bg_t)) & (iou < np.float32(fg_t))] = -1
    else:                                                                                    # :45-54
        fgm, bgm = iou > np.float32(fg_t), iou < np.float32(bg_t)
        lab = fgm.astype(np.float32)
        mid = ~fgm & ~bgm
        lab[mid] = (iou[mid] - np.float32(bg_t)) / np.float32(fg_t - bg_t)
    out["rcnn_cls_labels"] = lab
    return out


def uniforms_for(max_overlaps, sampled, cfg):
    """The (key, pick) uniform numbers under which `subsample` reproduces the logged draws of one frame
    (test helper: replays the reference's own random choices through our sampler interface)."""
    R, P = len(max_overlaps), cfg["ROI_PER_IMAGE"]
    mo = max_overlaps
    fg_thresh = np.float32(min(cfg["REG_FG_THRESH"], cfg["CLS_FG_THRESH"]))
    lo = np.float32(cfg["CLS_BG_THRESH_LO"])
    cat = np.where(mo >= fg_thresh, 0, np.where(mo < lo, 2, 1))
    lists = [np.nonzero(cat == c)[0] for c in range(3)]
    key = np.full(R, 0.99, np.float32)
    pick = np.zeros(P, np.float32)
    nfg, nbg = len(lists[0]), len(lists[1]) + len(lists[2])
    take = min(int(np.round(cfg["FG_RATIO"] * P)), nfg) if (nfg and nbg) else 0
    for j in range(take):
        key[sampled[j]] = j / (2.0 * P)
    for q in range(take, P):
        lst = lists[cat[sampled[q]]]
        pos = int(np.nonzero(lst == sampled[q])[0][0])
        pick[q] = (pos + 0.5) / len(lst)
    return key, pick
