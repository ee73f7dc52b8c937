"""GLENet's KL regression loss of the RoI head against the golden fixture generated from the
reference's own code (tests/golden/make_golden.py kl): tensor-op mirror on CPU, fused kernel on GPU."""
import os

import numpy as np
import pytest
import torch

from glenet_amd import losses

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kl_loss_ref.npz"))


def _inputs(dev):
    t = lambda k: torch.from_numpy(G[k]).to(dev)   # noqa: E731
    reg = t("rcnn_reg").requires_grad_(True)
    std = t("rcnn_reg_std").requires_grad_(True)
    return reg, std, t("rois"), t("gt_of_rois"), t("gt_uncertainty"), t("reg_valid_mask")


def _check(loss, parts, reg, std, rtol):
    np.testing.assert_allclose(float(loss.detach()), float(G["loss"]), rtol=rtol)
    for k, g in (("src", "loss_src"), ("square", "loss_square"), ("log", "loss_log")):
        np.testing.assert_allclose(float(parts[k]), float(G[g]), rtol=rtol, atol=1e-6)
    loss.backward()
    np.testing.assert_allclose(reg.grad.cpu().numpy(), G["grad_reg"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(std.grad.cpu().numpy(), G["grad_std"], rtol=1e-5, atol=1e-7)
    assert float(std.grad[5, 2]) == 0.0                      # clamped at -50: no gradient


def test_kl_reg_loss_tensor_ops_match_reference_golden():
    reg, std, rois, gt, unc, valid = _inputs("cpu")
    loss, parts = losses.kl_reg_loss_torch(reg, std, rois, gt, unc, valid, code_weights=G["code_weights"].tolist(),
                                           beta=float(G["beta"]))
    _check(loss, parts, reg, std, 1e-6)
    with pytest.raises(Exception):          # the public entry points take device tensors only: no CPU fallback
        losses.kl_reg_loss(reg, std, rois, gt, unc, valid)


@pytest.mark.gpu
def test_kl_reg_loss_kernel_matches_reference_golden(dev):
    """One launch, no read-back: loss, its three parts, #foreground and both gradients."""
    reg, std, rois, gt, unc, valid = _inputs(dev)
    loss, parts = losses.kl_reg_loss(reg, std, rois, gt, unc, valid, code_weights=G["code_weights"].tolist(),
                                     beta=float(G["beta"]))
    assert int(parts["fg"]) == int(G["fg_sum"])
    _check(loss, parts, reg, std, 2e-6)
    # no foreground at all: loss 0, gradients 0 (the reference divides by max(fg_sum, 1))
    reg2, std2 = reg.detach().clone().requires_grad_(True), std.detach().clone().requires_grad_(True)
    l0, p0 = losses.kl_reg_loss(reg2, std2, rois, gt, unc, torch.zeros_like(valid))
    l0.backward()
    assert float(l0.detach()) == 0.0 and float(reg2.grad.abs().max()) == 0.0 and float(std2.grad.abs().max()) == 0.0


def test_corner_loss_tensor_ops_match_reference_golden():
    reg, _, rois, _, _, valid = _inputs("cpu")
    loss = losses.corner_loss_torch(reg, rois, torch.from_numpy(G["gt_of_rois_src"]), valid)
    np.testing.assert_allclose(float(loss.detach()), float(G["loss_corner"]), rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(reg.grad.numpy(), G["grad_reg_corner"], rtol=1e-4, atol=1e-7)


@pytest.mark.gpu
def test_corner_loss_kernel_matches_reference_golden(dev):
    """Analytic gradient of the fused kernel vs the reference's autograd (flipped-heading branch and
    both smooth-L1 regimes occur in the fixture)."""
    reg, _, rois, _, _, valid = _inputs(dev)
    loss = losses.corner_loss(reg, rois, torch.from_numpy(G["gt_of_rois_src"]).to(dev), valid)
    np.testing.assert_allclose(float(loss.detach()), float(G["loss_corner"]), rtol=1e-5)
    loss.backward()
    got, want = reg.grad.cpu().numpy(), G["grad_reg_corner"]
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-7 + 1e-5 * np.abs(want).max())
    assert np.abs(want).max() > 0 and (got[G["reg_valid_mask"] == 0] == 0).all()
    reg2 = reg.detach().clone().requires_grad_(True)
    l0 = losses.corner_loss(reg2, rois, torch.from_numpy(G["gt_of_rois_src"]).to(dev), torch.zeros_like(valid))
    l0.backward()
    assert float(l0.detach()) == 0.0 and float(reg2.grad.abs().max()) == 0.0


def _canon_close(got, want):
    # headings on the +-pi/2 fold may land on either side: compare them modulo pi
    np.testing.assert_allclose(got[..., [0, 1, 2, 3, 4, 5, 7]], want[..., [0, 1, 2, 3, 4, 5, 7]], rtol=1e-5, atol=1e-5)
    dh = np.abs(got[..., 6] - want[..., 6])
    assert (np.minimum(dh, np.abs(dh - np.pi)) < 1e-5).all()
    assert (np.abs(got[..., 6]) <= np.pi / 2 + 1e-6).all()


def test_canonical_gt_tensor_ops_match_reference_golden():
    got = losses.canonical_gt_of_rois_torch(torch.from_numpy(G["canon_rois"]), torch.from_numpy(G["canon_gt"]))
    _canon_close(got.numpy(), G["canon_out"])


@pytest.mark.gpu
def test_canonical_gt_kernel_matches_reference_golden(dev):
    got = losses.canonical_gt_of_rois(torch.from_numpy(G["canon_rois"]).to(dev), torch.from_numpy(G["canon_gt"]).to(dev))
    assert got.shape == G["canon_out"].shape
    _canon_close(got.cpu().numpy(), G["canon_out"])


def test_rcnn_cls_loss_tensor_ops_match_reference_golden():
    x = torch.from_numpy(G["cls_logits"]).requires_grad_(True)
    loss = losses.rcnn_cls_loss_torch(x, torch.from_numpy(G["cls_labels"]))
    np.testing.assert_allclose(float(loss.detach()), float(G["cls_loss"]), rtol=1e-6)
    loss.backward()
    np.testing.assert_allclose(x.grad.numpy(), G["cls_grad"], rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
def test_rcnn_cls_loss_kernel_matches_reference_golden(dev):
    x = torch.from_numpy(G["cls_logits"]).to(dev).requires_grad_(True)
    loss = losses.rcnn_cls_loss(x, torch.from_numpy(G["cls_labels"]).to(dev))
    np.testing.assert_allclose(float(loss.detach()), float(G["cls_loss"]), rtol=2e-6)
    loss.backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), G["cls_grad"], rtol=2e-5, atol=1e-9)
    # ignored RoIs (label -1, CLS_SCORE_TYPE cls): out of the sum, the count and the gradient
    lab = torch.from_numpy(G["cls_labels"]).to(dev).clone()
    lab[0, 5:25] = -1
    x2 = x.detach().clone().requires_grad_(True)
    l2 = losses.rcnn_cls_loss(x2, lab)
    keep = (lab.reshape(-1) >= 0)
    want = torch.nn.functional.binary_cross_entropy(torch.sigmoid(x2.detach().reshape(-1)[keep]), lab.reshape(-1)[keep])
    np.testing.assert_allclose(float(l2.detach()), float(want), rtol=1e-5)
    l2.backward()
    assert float(x2.grad.reshape(2, -1)[0, 5:25].abs().max()) == 0.0


@pytest.mark.gpu
def test_fused_score_rescaling_and_cls_loss_match_the_tensor_formulation(dev):
    """glx_cls_rescale_loss (rescaling of voxelrcnn_kl_label_iou_head.py:70-76 + get_box_cls_layer_loss + chain rule,
    one launch) vs the tensor ops: cls_rescale_torch feeding rcnn_cls_loss_torch (pinned by the reference golden
    above) under autograd -- value, the rescaled logit, and the gradients of both logits."""
    g = torch.Generator().manual_seed(5)
    R = 1000
    a = (torch.randn(R, 1, generator=g) * 3).to(dev)
    b = (torch.randn(R, 1, generator=g) * 2 + 1).to(dev)
    a[:4, 0] = torch.tensor([40.0, -40.0, 12.0, -12.0])           # saturated logits: the 1e-6 guards at work
    b[:4, 0] = torch.tensor([40.0, 40.0, -30.0, 30.0])
    lab = torch.rand(R, generator=g).to(dev)
    lab[10:60] = -1.0                                             # ignored RoIs
    lab[60:90] = 0.0
    lab[90:120] = 1.0
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    zr = losses.cls_rescale_torch(ar, br)
    want = losses.rcnn_cls_loss_torch(zr, lab, weight=1.5)
    (want * 0.7).backward()
    af, bf = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    got, z = losses.cls_rescale_loss(af, bf, lab, weight=1.5)
    assert z.shape == a.shape and not z.requires_grad
    (got * 0.7).backward()
    np.testing.assert_allclose(float(got.detach()), float(want.detach()), rtol=1e-5)
    np.testing.assert_allclose(z.cpu().numpy(), zr.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    scale = float(ar.grad.abs().max())
    np.testing.assert_allclose(af.grad.cpu().numpy(), ar.grad.cpu().numpy(), rtol=2e-4, atol=1e-6 * scale)
    np.testing.assert_allclose(bf.grad.cpu().numpy(), br.grad.cpu().numpy(), rtol=2e-4, atol=1e-6 * scale)
    assert float(af.grad[10:60].abs().max()) == 0.0
    # the rescaling alone (inference) and the unit-gradient shortcut of the training step
    np.testing.assert_allclose(losses.cls_rescale(a, b).cpu().numpy(), z.cpu().numpy(), rtol=0, atol=0)
    a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    losses.UNIT_ROOT_GRAD = True
    try:
        l2, _ = losses.cls_rescale_loss(a2, b2, lab, weight=1.5)
        l2.backward()
    finally:
        losses.UNIT_ROOT_GRAD = False
    np.testing.assert_allclose(a2.grad.cpu().numpy() * 0.7, af.grad.cpu().numpy(), rtol=1e-6, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("no_fg", [False, True])
def test_roi_head_losses_one_launch_equals_the_three_entry_points(dev, no_fg):
    """glx_roi_head_losses (the three RoI-head terms, their sum and d loss / d rcnn_reg = KL + corner in one launch,
    8-column ground-truth rows through their stride, int64 mask) vs cls_rescale_loss + kl_reg_loss + corner_loss --
    each pinned by the reference goldens above -- on the golden fixture's rows: values 2e-6, gradients 1e-5."""
    reg0, std0, rois, gt, unc, valid = _inputs(dev)
    R = reg0.shape[0]
    g = torch.Generator().manual_seed(11)
    a0 = (torch.randn(R, 1, generator=g) * 3).to(dev)
    b0 = (torch.randn(R, 1, generator=g) * 2 + 1).to(dev)
    lab = torch.rand(R, generator=g).to(dev)
    lab[::5] = -1.0
    gt_src = torch.from_numpy(G["gt_of_rois_src"]).to(dev)
    pad = lambda t: torch.cat([t.reshape(R, 7), torch.full((R, 1), 3.0, device=dev)], 1).contiguous()   # noqa: E731  + class column
    gt8, src8 = pad(gt), pad(gt_src)
    valid = (torch.zeros_like(valid) if no_fg else valid).long().contiguous()
    cw, beta = G["code_weights"].tolist(), float(G["beta"])
    leaves = lambda: [t.detach().clone().requires_grad_(True) for t in (a0, b0, reg0, std0)]   # noqa: E731
    a, b, reg, std = leaves()
    l_cls, z = losses.cls_rescale_loss(a, b, lab, weight=1.5)
    l_kl, kp = losses.kl_reg_loss(reg, std, rois, gt, unc, valid, code_weights=cw, beta=beta, weight=0.8)
    l_cor = losses.corner_loss(reg, rois, gt_src, valid, weight=1.2)
    want = l_cls + l_kl + l_cor
    (want * 0.7).backward()
    af, bf, regf, stdf = leaves()
    assert losses.roi_head_losses_supported(af, regf, lab, rois, gt8, src8, unc, valid)
    got, parts, zf = losses.roi_head_losses(af, bf, regf, stdf, lab, rois, gt8, src8, unc, valid, code_weights=cw, beta=beta,
                                            w_cls=1.5, w_reg=0.8, w_corner=1.2)
    assert zf.shape == a0.shape and not zf.requires_grad and torch.equal(zf, z)
    (got * 0.7).backward()
    np.testing.assert_allclose(float(got.detach()), float(want.detach()), rtol=2e-6)
    for k, w in (("cls", l_cls), ("kl", l_kl), ("corner", l_cor), ("src", kp["src"]), ("square", kp["square"]),
                 ("log", kp["log"]), ("fg", kp["fg"])):
        np.testing.assert_allclose(float(parts[k]), float(w.detach()), rtol=2e-6, atol=1e-7, err_msg=k)
    for gf, gw, name in ((af, a, "ori"), (bf, b, "std logit"), (regf, reg, "reg"), (stdf, std, "reg std")):
        scale = float(gw.grad.abs().max())
        np.testing.assert_allclose(gf.grad.cpu().numpy(), gw.grad.cpu().numpy(), rtol=1e-5, atol=1e-7 + 1e-6 * scale,
                                   err_msg=name)
    if no_fg:
        assert float(regf.grad.abs().max()) == 0.0 and float(parts["corner"]) == 0.0 and float(parts["kl"]) == 0.0
    else:
        assert float(parts["corner"]) > 0 and float(regf.grad.abs().max()) > 0


def _kl_head_golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "kl_label_head_ref.npz"))


def test_single_stage_kl_label_head_losses_match_reference_golden():
    """GLENet-S / -C dense head (anchor_head_kl_label.py): KL box regression + IoU-prediction losses and their gradients
    against the reference's AnchorHeadKLLabelIoU run on CPU (tests/golden/make_golden.py:make_kl_label_head_ref), given the
    reference's own target assignment.  The aligned IoU3D targets come from the oracle through the statements of
    iou3d_utils.boxes_aligned_iou3d_gpu (the device test uses the kernel)."""
    import oracle
    from glenet_amd import losses
    from glenet_amd.pcdet_ops.iou3d.iou3d_utils import boxes3d_to_bev_torch
    g = _kl_head_golden()
    T = lambda k: torch.from_numpy(g[k])                                                  # noqa: E731
    B = g["in/box_preds"].shape[0]
    box, std, iou = (T("in/" + k).clone().requires_grad_(True) for k in ("box_preds", "box_std_preds", "iou_preds"))
    labels, targets = T("box_cls_labels"), T("box_reg_targets")
    unc = T("reg_weights").view(B, -1, 7)
    loc, parts = losses.rpn_kl_box_loss(box, std, targets, labels, unc, loc_weight=float(g["loc_weight"]),
                                        code_weights=tuple(g["code_weights"].tolist()))
    for k in ("rpn_loss_loc", "rpn_loss_loc_src", "rpn_loss_loc_square", "rpn_loss_loc_log"):
        np.testing.assert_allclose(float(parts[k]), float(g["tb/" + k]), rtol=2e-5)
    np.testing.assert_allclose(float(loc) + float(g["tb/rpn_loss_dir"]), float(g["box_loss"]), rtol=2e-5)

    def aligned_cpu(a, b):
        ov = torch.from_numpy(oracle.iou3d_boxes_aligned_overlap_bev(boxes3d_to_bev_torch(a).numpy(),
                                                                   boxes3d_to_bev_torch(b).numpy())).view(-1, 1)
        hmin = torch.max(a[:, 2] - a[:, 5] / 2, b[:, 2] - b[:, 5] / 2).view(-1, 1)
        hmax = torch.min(a[:, 2] + a[:, 5] / 2, b[:, 2] + b[:, 5] / 2).view(-1, 1)
        o3 = ov * torch.clamp(hmax - hmin, min=0)
        va, vb = (a[:, 3] * a[:, 4] * a[:, 5]).view(-1, 1), (b[:, 3] * b[:, 4] * b[:, 5]).view(-1, 1)
        return o3 / torch.clamp(va + vb - o3, min=1e-7)
    il, iparts = losses.rpn_iou_loss(iou, box, targets, labels, T("anchors"), aligned_iou3d=aligned_cpu)
    np.testing.assert_allclose(float(il), float(g["iou_loss"]), rtol=2e-5)
    (loc + il).backward()
    for t, k in ((box, "box_preds"), (std, "box_std_preds"), (iou, "iou_preds")):
        want = g["grad/" + k]
        np.testing.assert_allclose(t.grad.numpy(), want, rtol=1e-4, atol=1e-6 * (np.abs(want).max() + 1e-12), err_msg=k)
    # the direction term of the same head is AnchorHeadTemplate's (pinned by target_assign_ref.npz): same value here
    from glenet_amd.losses import rpn_loss_torch
    _, p = rpn_loss_torch(T("in/cls_preds"), T("in/box_preds"), T("in/dir_cls_preds"), labels, targets, T("anchors"),
                          code_weights=tuple(g["code_weights"].tolist()), loc_weight=float(g["loc_weight"]),
                          dir_weight=float(g["dir_weight"]), dir_offset=float(g["dir_offset"]))
    np.testing.assert_allclose(float(p["rpn_loss_dir"]), float(g["tb/rpn_loss_dir"]), rtol=2e-5)


@pytest.mark.gpu
def test_single_stage_kl_label_head_on_device(dev):
    """The same head on the device: glx_assign_targets_ex reproduces the reference's WeightedAxisAlignedTargetAssigner
    (labels bit-identical, targets 1e-6, per-anchor label uncertainties exact) and the two losses (IoU targets from the
    glx aligned-IoU kernel) match the reference-generated values."""
    from glenet_amd import losses, target_assign
    g = _kl_head_golden()
    T = lambda k: torch.from_numpy(g[k]).to(dev)                                          # noqa: E731
    anchors6 = T("anchors").view(*[int(v) for v in g["anchors_grid"]])
    got = target_assign.assign_targets([anchors6], T("gt_boxes"), [1], [float(g["matched_threshold"])],
                                       [float(g["unmatched_threshold"])], gt_uncertaintys=T("gt_uncertaintys"))
    assert np.array_equal(got["box_cls_labels"].cpu().numpy(), g["box_cls_labels"])
    np.testing.assert_allclose(got["box_reg_targets"].cpu().numpy(), g["box_reg_targets"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(got["reg_weights"].cpu().numpy(), g["reg_weights"])
    assert (g["box_cls_labels"] > 0).sum() == 29
    box, std, iou = (T("in/" + k).clone().requires_grad_(True) for k in ("box_preds", "box_std_preds", "iou_preds"))
    loc, parts = losses.rpn_kl_box_loss(box, std, got["box_reg_targets"], got["box_cls_labels"], got["label_uncertainty"],
                                        loc_weight=float(g["loc_weight"]), code_weights=tuple(g["code_weights"].tolist()))
    il, _ = losses.rpn_iou_loss(iou, box, got["box_reg_targets"], got["box_cls_labels"], T("anchors"))
    np.testing.assert_allclose(float(loc.detach()), float(g["tb/rpn_loss_loc"]), rtol=2e-5)
    np.testing.assert_allclose(float(il.detach()), float(g["iou_loss"]), rtol=1e-4)
    (loc + il).backward()
    for t, k in ((box, "box_preds"), (std, "box_std_preds"), (iou, "iou_preds")):
        want = g["grad/" + k]
        np.testing.assert_allclose(t.grad.cpu().numpy(), want, rtol=1e-3, atol=1e-5 * (np.abs(want).max() + 1e-12), err_msg=k)
