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


def test_dense_head_loss_kernel_vs_mirror_at_full_size(dev):
    """glx_rpn_loss vs the tensor-op mirror (pinned to the reference on CPU) at the GLENet-VR size:
    4 frames x 70 400 anchors, targets from our own assignment; NaN-free, parts 2e-5, gradients 1e-4."""
    from glenet_amd import detector as det, losses
    torch.manual_seed(11)
    anchors = det.generate_anchors([0, -40.0, -3, 70.4, 40.0, 1], (176, 200), [[3.9, 1.6, 1.56]], [0, 1.57], [-1.78], device=dev)
    gt = torch.zeros(4, 30, 8, device=dev)
    for b, n in enumerate((25, 12, 0, 30)):
        gt[b, :n, 0] = torch.rand(n, device=dev) * 68 + 1
        gt[b, :n, 1] = torch.rand(n, device=dev) * 76 - 38
        gt[b, :n, 2] = -1.0
        gt[b, :n, 3:6] = torch.tensor([3.9, 1.6, 1.56], device=dev) * (1 + torch.randn(n, 3, device=dev) * 0.05)
        gt[b, :n, 6] = torch.rand(n, device=dev) * 6.28 - 3.14
        gt[b, :n, 7] = 1
    tgt = target_assign.assign_targets([anchors], gt, [1], [0.6], [0.45])
    preds = [torch.randn(4, 200, 176, c, device=dev) * s for c, s in ((2, 2.0), (14, 0.3), (4, 1.0))]
    a = [p.clone().requires_grad_(True) for p in preds]
    b = [p.clone().requires_grad_(True) for p in preds]
    la, pa = losses.rpn_loss(*a, tgt["box_cls_labels"], tgt["box_reg_targets"], anchors)
    lb, pb = losses.rpn_loss_torch(*b, tgt["box_cls_labels"], tgt["box_reg_targets"], anchors)
    np.testing.assert_allclose(float(la.detach()), float(lb.detach()), rtol=2e-5)
    for k in pa:
        np.testing.assert_allclose(float(pa[k]), float(pb[k]), rtol=2e-5, atol=1e-7)
    la.backward()
    lb.backward()
    for x, y in zip(a, b):
        w = y.grad.cpu().numpy()
        np.testing.assert_allclose(x.grad.cpu().numpy(), w, rtol=1e-4, atol=1e-6 * max(1e-3, np.abs(w).max()))
    assert int((tgt["box_cls_labels"][2] > 0).sum()) == 0          # the frame without ground truth


def test_forced_anchor_carries_the_uncertainty_of_its_argmax_ground_truth(dev):
    """weighted_axis_aligned_target_assigner.py:166-169: `gt_inds_force = anchor_to_gt_argmax[anchors_with_max_overlap]`,
    `reg_weights[anchors_with_max_overlap] = gt_uncertaintys[gt_inds_force]` -- an anchor that ground truth B forces
    positive (it is B's best anchor) but whose own arg-max is ground truth A takes A's label uncertainty, the box its
    regression targets are encoded from (it used to take B's)."""
    def box(x, y):
        return [x, y, -1.0, 3.9, 1.6, 1.56, 0.0]
    # anchor 1 overlaps A (y 0) by 0.56 and B (y 1) by 0.49: arg-max A, below the 0.6 threshold, and nobody overlaps B more
    anchors = torch.tensor([box(10, 0.1), box(10, 0.45), box(30, 5.0)], device=dev).view(1, 1, 3, 1, 1, 7)
    gt = torch.zeros((1, 4, 8), device=dev)
    gt[0, 0, :7], gt[0, 1, :7] = torch.tensor(box(10, 0.0)), torch.tensor(box(10, 1.0))
    gt[0, :2, 7] = 1
    unc = torch.zeros((1, 4, 7), device=dev)
    unc[0, 0], unc[0, 1] = 0.11, 0.77
    out = target_assign.assign_targets([anchors], gt, [1], [0.6], [0.45], gt_uncertaintys=unc)
    lab = out["box_cls_labels"].cpu().numpy()[0]
    lu = out["label_uncertainty"].cpu().numpy()[0]
    assert lab.tolist() == [1, 1, 0]
    np.testing.assert_array_equal(lu[0], np.full(7, 0.11, np.float32))      # over the threshold: arg-max A
    np.testing.assert_array_equal(lu[1], np.full(7, 0.11, np.float32))      # forced by B, arg-max A -> A's
    np.testing.assert_array_equal(lu[2], np.zeros(7, np.float32))
    # its regression targets are A's too: dy = (0 - 0.45) / diag
    want = (0.0 - 0.45) / np.sqrt(3.9 ** 2 + 1.6 ** 2)
    np.testing.assert_allclose(out["box_reg_targets"].cpu().numpy()[0, 1, 1], want, rtol=1e-6)
