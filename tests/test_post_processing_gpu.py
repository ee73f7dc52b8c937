"""Inference post-processing on the device (glenet_amd.detector.post_processing: score threshold, top-k, heading wrap,
variance = exp(log-variance), the greedy variance-voting NMS per frame, post max size, POST_SCORE_THRESH) against the
oracle's restatement of Detector3DTemplate.post_processing + class_agnostic_nms + new_nms_gpu (oracle.post_processing,
itself pinned to the reference's own pred_dicts by tests/test_reference_step_cpu.py), at the sizes the configs reach:
100 RoIs per frame (GLENet-VR) and NMS_PRE_MAXSIZE = 4096 candidates per frame (GLENet-S / -C)."""
import numpy as np
import pytest
import torch

import oracle
from glenet_amd import synth

pytestmark = pytest.mark.gpu


def _frames(seed, B, R, lo=0.02, hi=0.99):
    rng = np.random.default_rng(seed)
    boxes = np.stack([synth.random_boxes(rng, R, xy_range=60.0 if R > 1000 else 30.0, near_dup=0.6) for _ in range(B)])
    boxes[..., 6] += rng.choice([0.0, 2 * np.pi, -2 * np.pi], (B, R))            # headings outside [-pi, pi): the wrap
    scores = rng.uniform(lo, hi, (B, R)).astype(np.float32)
    logits = np.log(scores / (1 - scores)).astype(np.float32)[..., None]
    std = rng.normal(-2.0, 0.7, (B, R, 7)).astype(np.float32)
    labels = rng.integers(1, 4, (B, R)).astype(np.int64)
    return boxes.astype(np.float32), logits, std, labels


def _compare(dev, boxes, logits, std, labels, cfg, use_std=True):
    from glenet_amd import detector as det
    t = lambda a: torch.from_numpy(a).to(dev)                                    # noqa: E731
    post = det.post_processing(t(logits), t(boxes), t(std) if use_std else None, t(labels), cfg)
    num = post["num"].cpu().numpy()
    sig = torch.sigmoid(t(logits)).cpu().numpy()                                 # the device's own sigmoid: exact compare
    kept = 0
    for b in range(boxes.shape[0]):
        wb, ws, wl, wsel = oracle.post_processing(sig[b], boxes[b], std[b] if use_std else None, labels[b], normalized=True,
                                                  score_thresh=cfg.get("SCORE_THRESH"), post_score_thresh=cfg.get("POST_SCORE_THRESH"),
                                                  nms_thresh=cfg["NMS_THRESH"], nms_pre_maxsize=cfg["NMS_PRE_MAXSIZE"],
                                                  nms_post_maxsize=cfg["NMS_POST_MAXSIZE"])
        n = int(num[b])
        assert n == len(ws), "frame %d: %d detections, oracle %d" % (b, n, len(ws))
        assert np.array_equal(post["pred_index"][b, :n].cpu().numpy(), wsel)         # keep list and order: bit-exact
        assert np.array_equal(post["pred_scores"][b, :n].cpu().numpy(), ws)
        assert np.array_equal(post["pred_labels"][b, :n].cpu().numpy(), wl)
        np.testing.assert_allclose(post["pred_boxes"][b, :n].cpu().numpy(), wb, rtol=1e-4, atol=1e-4)
        assert not post["pred_boxes"][b, n:].any() and (post["pred_index"][b, n:] == -1).all()
        kept += n
    return kept


def test_post_processing_glenet_vr_sizes_vs_oracle(dev):
    cfg = dict(SCORE_THRESH=0.3, POST_SCORE_THRESH=0.81, NMS_THRESH=0.1, NMS_PRE_MAXSIZE=4096, NMS_POST_MAXSIZE=500)
    kept = _compare(dev, *_frames(1, 4, 100), cfg)
    assert kept > 8
    # no POST_SCORE_THRESH, truncation by NMS_POST_MAXSIZE
    cfg2 = dict(SCORE_THRESH=0.1, POST_SCORE_THRESH=None, NMS_THRESH=0.1, NMS_PRE_MAXSIZE=64, NMS_POST_MAXSIZE=10)
    assert _compare(dev, *_frames(2, 3, 100), cfg2) == 30
    # a frame in which nothing passes the score threshold, and plain greedy NMS without variances
    boxes, logits, std, labels = _frames(3, 2, 100)
    logits[1] = -5.0
    assert _compare(dev, boxes, logits, std, labels, cfg) > 0
    assert _compare(dev, boxes, logits, std, labels, cfg, use_std=False) > 0


def test_post_processing_at_nms_pre_maxsize_4096_vs_oracle(dev):
    """GLENet-S / -C hand every anchor above SCORE_THRESH 0.1, up to 4096, to new_nms_gpu (GLENet_S.yaml:93-106): 16.8 M
    rotated IoUs and up to 4096 greedy rounds per frame."""
    cfg = dict(SCORE_THRESH=0.1, POST_SCORE_THRESH=None, NMS_THRESH=0.01, NMS_PRE_MAXSIZE=4096, NMS_POST_MAXSIZE=500)
    boxes, logits, std, labels = _frames(4, 2, 6000, lo=0.05)
    kept = _compare(dev, boxes, logits, std, labels, cfg)
    assert kept > 100
