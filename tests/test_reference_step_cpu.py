"""tests/golden/ref_step.npz -- one training step and one inference pass of the REFERENCE'S OWN GLENet-VR classes
(make_golden.py refstep) -- checked on CPU: the fixture is self-consistent (parameters regenerate to the stored digest)
and the oracle's restatements, composed the way the product composes its kernels, reproduce what the reference's Python
produced: voxelization, the proposal layer (sigmoid, top-k, NMS 0.8, zero padding), RoI target sampling with the
replayed draws, the canonical transformation, and the variance-voting post-processing.  No GPU, no /root/reference."""
import json
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from oracle import roi_targets as ort

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import refstep_params as rp  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "ref_step.npz"))
VOXEL = [0.05, 0.05, 0.1]
TARGET = dict(ROI_PER_IMAGE=128, FG_RATIO=0.5, SAMPLE_ROI_BY_EACH_CLASS=True, CLS_SCORE_TYPE="roi_iou", CLS_FG_THRESH=0.75,
              CLS_BG_THRESH=0.25, CLS_BG_THRESH_LO=0.1, HARD_BG_RATIO=0.8, REG_FG_THRESH=0.55)      # GLENet_VR.yaml:140-153


def spec():
    return list(zip(G["param_names"].tolist(), [json.loads(s) for s in G["param_shapes"]], G["param_dtypes"].tolist()))


def test_parameters_regenerate_to_the_stored_digest():
    params = rp.make_params(spec(), int(G["seed"]))
    d = rp.digest(params)
    got = np.array([d[k] for k in G["param_names"].tolist()])
    np.testing.assert_allclose(got, G["param_digest"], rtol=1e-12, atol=0)
    assert sum(int(np.prod(v.shape)) for v in params.values()) > 7_000_000


def test_oracle_voxelizer_reproduces_the_input_the_reference_saw():
    pts, bidx = G["points"], G["batch_idx"]
    R = G["point_cloud_range"].tolist()
    coords, vox, num = [], [], []
    for b in range(2):
        v, c, n = oracle.voxelize_hard(pts[bidx == b], VOXEL, R, 5, 16000)
        coords.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
        vox.append(v)
        num.append(n)
    assert np.array_equal(np.concatenate(coords), G["voxel_coords"])
    assert np.array_equal(np.concatenate(vox), G["voxels"]) and np.array_equal(np.concatenate(num), G["voxel_num_points"])
    # MeanVFE (mean_vfe.py:14-31)
    np.testing.assert_array_equal(oracle.mean_vfe(G["voxels"], G["voxel_num_points"]), G["train_voxel_features"])


def _proposals(cls, boxes, pre, post, thr):
    """RoIHeadTemplate.proposal_layer + class_agnostic_nms over the oracle (roi_head_template.py:52-128)."""
    B = cls.shape[0]
    rois, scores, labels = np.zeros((B, post, 7), np.float32), np.zeros((B, post), np.float32), np.zeros((B, post), np.int64)
    for b in range(B):
        s = torch.sigmoid(torch.from_numpy(cls[b])).numpy().max(1)
        keep = oracle.nms_gpu(boxes[b], s, thr, pre_maxsize=min(pre, len(s)))[:post]
        rois[b, :len(keep)], scores[b, :len(keep)] = boxes[b][keep], s[keep]
        labels[b, :len(keep)] = 0
    return rois, scores, labels + 1


@pytest.mark.parametrize("tag,cfg", [("train", (9000, 512, 0.8)), ("eval", (2048, 100, 0.7))])
def test_oracle_proposal_layer_equals_the_references(tag, cfg):
    cls = G["train_batch_cls_preds"] if tag == "train" else G["eval_rpn_batch_cls_preds"]
    box = G["train_batch_box_preds"] if tag == "train" else G["eval_rpn_batch_box_preds"]
    rois, scores, labels = _proposals(cls, box, *cfg)
    want = (G["train_proposal_rois"], G["train_proposal_roi_scores"], G["train_proposal_roi_labels"]) if tag == "train" \
        else (G["eval_rois"], G["eval_roi_scores"], G["eval_roi_labels"])
    for b in range(rois.shape[0]):       # rows of exactly equal score: an unspecified order in the reference, compared as sets
        pg, pw = rp.canon_ties(scores[b], rois[b]), rp.canon_ties(want[1][b], want[0][b])
        assert np.array_equal(rois[b][pg], want[0][b][pw]) and np.array_equal(scores[b][pg], want[1][b][pw])
    assert np.array_equal(labels, want[2])


def test_oracle_roi_targets_equal_the_references_with_its_draws_replayed():
    kp = [ort.uniforms_for(G["train_max_overlaps"][b], G["train_sampled"][b], TARGET) for b in range(2)]
    key, pick = np.stack([k for k, _ in kp]), np.stack([p for _, p in kp])
    o = ort.roi_targets(G["train_proposal_rois"], G["train_proposal_roi_labels"], G["train_proposal_roi_scores"],
                        G["gt_boxes"], TARGET, key, pick, G["gt_uncertaintys"])
    assert np.array_equal(o["max_overlaps"], G["train_max_overlaps"]) and np.array_equal(o["sampled"], G["train_sampled"])
    for k, ref in (("rois", "train_rois"), ("gt_of_rois", "train_gt_of_rois_src"), ("gt_iou_of_rois", "train_gt_iou_of_rois"),
                   ("roi_labels", "train_roi_labels"), ("reg_valid_mask", "train_reg_valid_mask"),
                   ("rcnn_cls_labels", "train_rcnn_cls_labels"), ("gt_uncertaintys_of_rois", "train_gt_uncertaintys_of_rois")):
        assert np.array_equal(o[k], G[ref]), k
    assert int(G["train_reg_valid_mask"].sum()) >= 4                 # the step has foreground to regress


def test_oracle_post_processing_equals_the_references_pred_dicts():
    """oracle.post_processing (detector3d_template.py:179-317 over oracle.new_nms_gpu: score threshold 0.3, top-k, voting
    NMS 0.1 with variance = exp(std), post max 500, POST_SCORE_THRESH 0.81) against the reference's pred_dicts."""
    cls, box, std, lab = (G["eval_batch_cls_preds"], G["eval_batch_box_preds"], G["eval_batch_box_std_preds"],
                          G["eval_roi_labels"])
    total = 0
    for b in range(cls.shape[0]):
        sig = torch.sigmoid(torch.from_numpy(cls[b])).numpy()                      # the reference's own sigmoid
        boxes, scores, labels, _ = oracle.post_processing(sig, box[b], std[b], lab[b], normalized=True)
        np.testing.assert_array_equal(scores, G["eval_pred_scores_%d" % b])
        np.testing.assert_allclose(boxes, G["eval_pred_boxes_%d" % b], rtol=1e-5, atol=1e-5)
        assert np.array_equal(labels, G["eval_pred_labels_%d" % b])
        # ... and with the restated float32 sigmoid: the same detections, scores to one ulp
        b2, s2, l2, _ = oracle.post_processing(cls[b], box[b], std[b], lab[b])
        assert len(s2) == len(scores) and np.abs(s2 - scores).max() <= 1.2e-7 and np.array_equal(l2, labels)
        total += len(scores)
    assert total >= 4
    # some RoIs fell to each of the two score thresholds
    sc = torch.sigmoid(torch.from_numpy(cls[..., 0])).numpy()
    assert (sc < 0.3).any() and ((sc >= 0.3) & (sc <= 0.81)).any() and (sc > 0.81).any()
