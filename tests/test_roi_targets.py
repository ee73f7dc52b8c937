"""RoI targets of the second stage (ProposalTargetLayer): the numpy oracle against the golden fixture
made by the reference's own class (tests/golden/make_golden.py roitgt), and the device path
(glenet_amd.roi_targets, glx_roi_targets) against both.  The reference draws its samples on the host;
the fixture logs every draw and `uniforms_for` turns the log into the uniform numbers under which our
sampler has to reproduce exactly those choices."""
import os

import numpy as np
import pytest
import torch

from oracle import roi_targets as ort

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "roi_targets_ref.npz"))
BASE = dict(ROI_PER_IMAGE=32, FG_RATIO=0.5, CLS_FG_THRESH=0.75, CLS_BG_THRESH=0.25, CLS_BG_THRESH_LO=0.1,
            HARD_BG_RATIO=0.8, REG_FG_THRESH=0.55)
CASES = [("each_iou", True, "roi_iou", [0, 1, 2, 3]), ("all_cls", False, "cls", [0, 1, 2, 3]),
         ("unc", True, "roi_iou", [0, 2, 3])]
KEYS = ("rois", "gt_of_rois", "gt_iou_of_rois", "roi_scores", "roi_labels", "reg_valid_mask", "rcnn_cls_labels")


def _replayed(tag, cfg, n):
    kp = [ort.uniforms_for(G[tag + "_max_overlaps"][i], G[tag + "_sampled"][i], cfg) for i in range(n)]
    return np.stack([k for k, _ in kp]), np.stack([p for _, p in kp])


@pytest.mark.parametrize("tag,each,kind,frames", CASES)
def test_oracle_matches_reference_golden(tag, each, kind, frames):
    """Bit-exact: overlaps, sampled indices, every gathered tensor and both label formulas; frames with
    foreground + both backgrounds, without ground truth, all-foreground, interior padding row."""
    cfg = dict(BASE, SAMPLE_ROI_BY_EACH_CLASS=each, CLS_SCORE_TYPE=kind)
    key, pick = _replayed(tag, cfg, len(frames))
    o = ort.roi_targets(G["rois"][frames], G["roi_labels"][frames], G["roi_scores"][frames], G["gt_boxes"][frames],
                        cfg, key, pick, G["gt_uncertaintys"][frames] if tag == "unc" else None)
    assert np.array_equal(o["max_overlaps"], G[tag + "_max_overlaps"])
    assert np.array_equal(o["sampled"], G[tag + "_sampled"])
    for k in KEYS + (("gt_uncertaintys_of_rois",) if tag == "unc" else ()):
        assert np.array_equal(o[k], G[tag + "_" + k]), k


def _device_forward(dev, cfg, rois, labels, scores, gt, key, pick, unc=None):
    from glenet_amd import roi_targets
    layer = roi_targets.ProposalTargetLayer(cfg)
    bd = {"rois": torch.from_numpy(rois).to(dev), "roi_labels": torch.from_numpy(labels).to(dev),
          "roi_scores": torch.from_numpy(scores).to(dev), "gt_boxes": torch.from_numpy(gt).to(dev)}
    if unc is not None:
        bd["gt_uncertaintys"] = torch.from_numpy(unc).to(dev)
    td = layer(bd, key=None if key is None else torch.from_numpy(key).to(dev),
               pick=None if pick is None else torch.from_numpy(pick).to(dev))
    mo, ga, s, sg = layer.match_and_sample(bd["rois"], bd["roi_labels"], bd["gt_boxes"],
                                           key=None if key is None else torch.from_numpy(key).to(dev),
                                           pick=None if pick is None else torch.from_numpy(pick).to(dev))
    return td, mo.cpu().numpy(), s.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,each,kind,frames", CASES)
def test_device_roi_targets_match_reference_golden(dev, tag, each, kind, frames):
    """Overlaps within the 3-D IoU tolerance of the device routine (rtol 1e-5, atol 2e-6: trig is
    rounded from double on the device, glibc float in the oracle), sampled indices and all gathered
    tensors identical, soft labels 2e-5 (IoU tolerance / threshold span 0.5)."""
    cfg = dict(BASE, SAMPLE_ROI_BY_EACH_CLASS=each, CLS_SCORE_TYPE=kind)
    key, pick = _replayed(tag, cfg, len(frames))
    td, mo, s = _device_forward(dev, cfg, G["rois"][frames], G["roi_labels"][frames], G["roi_scores"][frames],
                                G["gt_boxes"][frames], key, pick, G["gt_uncertaintys"][frames] if tag == "unc" else None)
    np.testing.assert_allclose(mo, G[tag + "_max_overlaps"], rtol=1e-5, atol=2e-6)
    assert np.array_equal(s, G[tag + "_sampled"])
    for k in KEYS + (("gt_uncertaintys_of_rois",) if tag == "unc" else ()):
        got, want = td[k].cpu().numpy(), G[tag + "_" + k]
        if k == "gt_iou_of_rois" or (k == "rcnn_cls_labels" and kind == "roi_iou"):
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=2e-5)
        else:
            assert np.array_equal(got, want), k


def _scene(rng, B, R, G_, n_gt, near_frac):
    def boxes(n):
        return np.concatenate([rng.uniform([0, -20, -2], [50, 20, 0], (n, 3)), rng.uniform([3, 1.4, 1.3], [4.5, 1.9, 1.8], (n, 3)),
                               rng.uniform(-3.1, 3.1, (n, 1))], 1).astype(np.float32)
    gt = np.zeros((B, G_, 8), np.float32)
    rois = np.zeros((B, R, 7), np.float32)
    labels = rng.integers(1, 4, (B, R)).astype(np.int64)
    for b in range(B):
        n = n_gt[b]
        gt[b, :n, :7] = boxes(n)
        gt[b, :n, 7] = rng.integers(1, 4, n)
        rois[b] = boxes(R)
        if n:
            src = rng.integers(0, n, R)
            near = rng.random(R) < near_frac[b]
            jit = rng.normal(0, 1, (R, 7)).astype(np.float32) * np.array([0.6, 0.35, 0.1, 0.2, 0.1, 0.1, 0.2], np.float32)
            rois[b][near] = (gt[b, src, :7] + jit)[near]
            labels[b][near] = gt[b, src, 7][near]
    return rois, labels, rng.random((B, R)).astype(np.float32), gt


@pytest.mark.gpu
@pytest.mark.parametrize("each", [True, False])
def test_device_roi_targets_match_oracle_on_random_scenes(dev, each):
    """512 RoIs x up to 40 ground truths, 128 samples (the GLENet_VR configuration), random draws:
    same overlaps / samples / targets as the oracle; includes a frame whose background is hard only."""
    rng = np.random.default_rng(3 + each)
    cfg = dict(BASE, ROI_PER_IMAGE=128, SAMPLE_ROI_BY_EACH_CLASS=each, CLS_SCORE_TYPE="roi_iou")
    rois, labels, scores, gt = _scene(rng, 4, 512, 40, [23, 40, 1, 0], [0.5, 0.9, 0.3, 0.0])
    rois[2] = gt[2, 0, :7] + np.array([1.2, 0.3, 0, 0, 0, 0, 0.05], np.float32) * rng.uniform(0.5, 1.0, (512, 1)).astype(np.float32)
    labels[2] = int(gt[2, 0, 7])                                     # frame 2: every RoI overlaps the one box partly
    key, pick = rng.random((4, 512)).astype(np.float32), rng.random((4, 128)).astype(np.float32)
    o = ort.roi_targets(rois, labels, scores, gt, cfg, key, pick)
    td, mo, s = _device_forward(dev, cfg, rois, labels, scores, gt, key, pick)
    np.testing.assert_allclose(mo, o["max_overlaps"], rtol=1e-5, atol=2e-6)
    assert np.array_equal(s, o["sampled"])
    for k in KEYS:
        if k in ("rcnn_cls_labels", "gt_iou_of_rois"):
            np.testing.assert_allclose(td[k].cpu().numpy(), o[k], rtol=1e-5, atol=2e-5)
        else:
            assert np.array_equal(td[k].cpu().numpy(), o[k]), k
    m2 = o["max_overlaps"][2]
    assert ((m2 >= 0.1) & (m2 < 0.55)).sum() > 0 and (m2 < 0.1).sum() == 0


@pytest.mark.gpu
def test_device_roi_sampler_with_overlapping_foreground_and_hard_background(dev):
    """CLS_FG_THRESH < REG_FG_THRESH: RoIs with an overlap in [CLS_FG, REG_FG) are in the foreground list AND in the
    hard-background list (three independent nonzero() calls, proposal_target_layer.py:128-137); a ROI_PER_IMAGE /
    proposal count that needs more than 64 KB of LDS for the three lists runs too."""
    rng = np.random.default_rng(9)
    cfg = dict(BASE, ROI_PER_IMAGE=64, SAMPLE_ROI_BY_EACH_CLASS=False, CLS_SCORE_TYPE="cls", CLS_FG_THRESH=0.4,
               REG_FG_THRESH=0.7)
    rois, labels, scores, gt = _scene(rng, 3, 6000, 30, [20, 30, 5], [0.5, 0.8, 0.3])
    key, pick = rng.random((3, 6000)).astype(np.float32), rng.random((3, 64)).astype(np.float32)
    o = ort.roi_targets(rois, labels, scores, gt, cfg, key, pick)
    td, mo, s = _device_forward(dev, cfg, rois, labels, scores, gt, key, pick)
    assert np.array_equal(s, o["sampled"])
    both = (o["max_overlaps"] >= 0.4) & (o["max_overlaps"] < 0.7)
    assert both.sum() > 0
    bg_slots = o["sampled"][:, 32:]
    assert np.take_along_axis(both, bg_slots, 1).sum() > 0          # such RoIs were drawn as hard background


@pytest.mark.gpu
def test_device_roi_sampler_default_draws_have_the_reference_composition(dev):
    """With the built-in generator: foreground slots hold min(64, #fg) DISTINCT foreground RoIs, then
    min(int(bg * 0.8), #hard) hard-background draws, then easy ones (proposal_target_layer.py:139-193)."""
    rng = np.random.default_rng(11)
    cfg = dict(BASE, ROI_PER_IMAGE=128, SAMPLE_ROI_BY_EACH_CLASS=True, CLS_SCORE_TYPE="roi_iou")
    rois, labels, scores, gt = _scene(rng, 3, 512, 30, [20, 30, 5], [0.6, 0.05, 0.4])
    torch.manual_seed(0)
    td, mo, s = _device_forward(dev, cfg, rois, labels, scores, gt, None, None)
    for b in range(3):
        cat = np.where(mo[b] >= 0.55, 0, np.where(mo[b] < 0.1, 2, 1))
        nfg, nhard = int((cat == 0).sum()), int((cat == 1).sum())
        take = min(64, nfg)
        hard_num = min(int((128 - take) * 0.8), nhard)
        got = cat[s[b]]
        assert (got[:take] == 0).all() and len(set(s[b][:take].tolist())) == take
        assert (got[take:take + hard_num] == 1).all() and (got[take + hard_num:] == 2).all()


@pytest.mark.gpu
def test_device_roi_targets_reject_host_tensors_and_oversized_inputs(dev):
    """No CPU path: host tensors raise; more ground-truth rows than the kernel's LDS table raise."""
    from glenet_amd import roi_targets, _lib
    layer = roi_targets.ProposalTargetLayer(dict(BASE, SAMPLE_ROI_BY_EACH_CLASS=True, CLS_SCORE_TYPE="roi_iou"))
    with pytest.raises(_lib.GlxError):
        layer.match_and_sample(torch.zeros(1, 8, 7), torch.ones(1, 8, dtype=torch.long), torch.zeros(1, 4, 8))
    with pytest.raises(_lib.GlxError):
        layer.match_and_sample(torch.zeros(1, 8, 7, device=dev), torch.ones(1, 8, dtype=torch.long, device=dev),
                               torch.zeros(1, 300, 8, device=dev))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["roi_iou", "cls"])
def test_fused_target_gather_equals_the_tensor_statements(dev, kind):
    """glx_roi_target_gather (one launch) == the gather / where / compare statements of the reference's forward
    (proposal_target_layer.py:13-63) as tensor ops, bit for bit, including a frame without ground truth."""
    from glenet_amd import roi_targets
    rng = np.random.default_rng(21)
    rois, labels, scores, gt = _scene(rng, 3, 300, 12, [7, 0, 12], [0.5, 0.3, 0.7])
    unc = rng.random((3, 12, 7)).astype(np.float32)
    cfg = dict(BASE, CLS_SCORE_TYPE=kind)
    key, pick = rng.random((3, 300)).astype(np.float32), rng.random((3, BASE["ROI_PER_IMAGE"])).astype(np.float32)
    got, _, _ = _device_forward(dev, cfg, rois, labels, scores, gt, key, pick, unc)
    roi_targets.FUSED_GATHER = False
    try:
        want, _, _ = _device_forward(dev, cfg, rois, labels, scores, gt, key, pick, unc)
    finally:
        roi_targets.FUSED_GATHER = True
    assert got.keys() == want.keys()
    for k in want:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        assert torch.equal(got[k], want[k]), k
    assert float(got["gt_of_rois"][1].abs().max()) == 0.0 and float(got["gt_of_rois"][0].abs().max()) > 0
