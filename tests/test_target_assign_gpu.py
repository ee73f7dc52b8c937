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
