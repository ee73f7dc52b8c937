"""Anchor target assignment on the device against the golden fixture generated from the reference's
own AxisAlignedTargetAssigner (tests/golden/make_golden.py assign)."""
import os

import numpy as np
import pytest
import torch

from glenet_amd import target_assign

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "target_assign_ref.npz"))


@pytest.mark.parametrize("norm", [False, True])
def test_anchor_target_assignment_matches_reference_golden(dev, norm):
    """Labels (class id / background / don't care) bit-identical, regression targets 1e-6, weights
    exact; frames without ground truth of a class and zero-padded rows included; two anchor classes
    interleaved along the anchor-type axis like the reference's concatenation."""
    anchors = [torch.from_numpy(G["anchors_car"]).to(dev), torch.from_numpy(G["anchors_cyc"]).to(dev)]
    gt = torch.from_numpy(G["gt"]).to(dev)
    out = target_assign.assign_targets(anchors, gt, anchor_class_ids=[1, 3], matched_thresholds=[0.6, 0.5],
                                       unmatched_thresholds=[0.45, 0.35], norm_by_num_examples=norm)
    tag = "norm" if norm else "plain"
    lab = out["box_cls_labels"].cpu().numpy()
    assert lab.dtype == np.int32 and np.array_equal(lab, G["labels_" + tag])
    np.testing.assert_allclose(out["box_reg_targets"].cpu().numpy(), G["targets_" + tag], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out["reg_weights"].cpu().numpy(), G["weights_" + tag], rtol=1e-7, atol=0)
    assert (lab > 0).sum() > 10 and (lab < 0).sum() > 0 and (lab[2] == 0).all()


def test_dense_head_loss_kernel_matches_reference_golden(dev):
    """glx_rpn_loss (focal classification + sin-difference smooth-L1 + direction cross-entropy, all
    gradients) vs the reference's AnchorHeadTemplate.get_loss called unmodified on the same tensors
    (fixture: make_golden.py assign); targets come from our own assignment of the same ground truth."""
    from glenet_amd import losses
    anchors = torch.from_numpy(G["anchors_car"]).to(dev)
    tgt = target_assign.assign_targets([anchors], torch.from_numpy(G["gt"]).to(dev), [1], [0.6], [0.45])
    assert np.array_equal(tgt["box_cls_labels"].cpu().numpy(), G["rpn_labels"])
    cls = torch.from_numpy(G["rpn_cls_preds"]).to(dev).requires_grad_(True)
    box = torch.from_numpy(G["rpn_box_preds"]).to(dev).requires_grad_(True)
    dr = torch.from_numpy(G["rpn_dir_preds"]).to(dev).requires_grad_(True)
    loss, parts = losses.rpn_loss(cls, box, dr, tgt["box_cls_labels"], tgt["box_reg_targets"], anchors)
    np.testing.assert_allclose(float(loss.detach()), float(G["rpn_loss"]), rtol=2e-5)
    for k in ("rpn_loss_cls", "rpn_loss_loc", "rpn_loss_dir"):
        np.testing.assert_allclose(float(parts[k]), float(G[k]), rtol=2e-5, atol=1e-7)
    loss.backward()
    for t, k in ((cls, "rpn_grad_cls"), (box, "rpn_grad_box"), (dr, "rpn_grad_dir")):
        w = G[k]
        np.testing.assert_allclose(t.grad.cpu().numpy(), w, rtol=1e-4, atol=1e-6 * max(1e-3, np.abs(w).max()))
        assert np.abs(w).max() > 0


def test_anchor_target_assignment_vs_oracle_larger_case(dev):
    """Kernel vs the numpy oracle (itself pinned to the reference's golden) at a size the golden does
    not reach: 5 frames, up to 60 ground truths of two classes on a 88 x 100 map, exact duplicates of a
    ground truth (ties for a column's best anchor) and a frame padded with zeros only."""
    from glenet_amd import detector as det
    from oracle import assign
    rng = np.random.default_rng(5)
    pcr = [0, -40.0, -3, 70.4, 40.0, 1]
    a_car = det.generate_anchors(pcr, (88, 100), [[3.9, 1.6, 1.56]], [0, 1.57], [-1.78]).numpy()
    a_cyc = det.generate_anchors(pcr, (88, 100), [[1.76, 0.6, 1.73]], [0, 1.57], [-0.6]).numpy()
    B, M = 5, 64
    gt = np.zeros((B, M, 8), np.float32)
    for b, n in enumerate((60, 33, 1, 0, 17)):
        cls = rng.integers(0, 2, n) * 2 + 1
        size = np.where(cls[:, None] == 1, [[3.9, 1.6, 1.56]], [[1.76, 0.6, 1.73]]) * (1 + rng.normal(0, 0.07, (n, 3)))
        gt[b, :n] = np.concatenate([rng.uniform([1, -38, -1.5], [69, 38, -0.5], (n, 3)), size,
                                    rng.uniform(-3.14, 3.14, (n, 1)), cls[:, None]], 1)
    gt[0, 1] = gt[0, 0]                                              # duplicate ground truth: tied columns
    gt[4, 3, :7] = a_car.reshape(-1, 7)[4321]                        # exactly on an anchor
    gt[4, 3, 7] = 1
    want = assign.assign_targets([a_car, a_cyc], gt, [1, 3], [0.6, 0.5], [0.45, 0.35], norm=True)
    got = target_assign.assign_targets([torch.from_numpy(a_car).to(dev), torch.from_numpy(a_cyc).to(dev)],
                                       torch.from_numpy(gt).to(dev), [1, 3], [0.6, 0.5], [0.45, 0.35],
                                       norm_by_num_examples=True)
    assert np.array_equal(got["box_cls_labels"].cpu().numpy(), want[0])
    np.testing.assert_allclose(got["box_reg_targets"].cpu().numpy(), want[1], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(got["reg_weights"].cpu().numpy(), want[2], rtol=1e-6, atol=0)
    assert (want[0] > 0).sum() > 100 and (want[0][3] == 0).all()
