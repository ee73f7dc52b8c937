"""Dense-path harness modules and RoI-grid geometry against goldens
produced by the reference's own Python on CPU (tests/golden/make_golden.py::make_dense_path_ref)."""
import os

import numpy as np
import torch

from glenet_amd import dense_path as dp

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dense_path_ref.npz"))


def _load(module, prefix):
    sd = {k[len(prefix) + 1:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(prefix + "/")}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected           # every reference parameter has a home
    assert all("num_batches_tracked" in k for k in missing), missing
    return module.eval()


def test_bev_backbone_matches_reference_module():
    m = _load(dp.BEVBackbone(6, (1, 2), (1, 2), (8, 16), (1, 2), (8, 8)), "bev")
    with torch.no_grad():
        y = m({"spatial_features": torch.from_numpy(G["bev_in"])})["spatial_features_2d"]
    np.testing.assert_allclose(y.numpy(), G["bev_out"], rtol=1e-5, atol=1e-5)
    # the GLENet-VR configuration has the parameter names of released checkpoints
    keys = dp.BEVBackbone(256).state_dict().keys()
    assert "blocks.0.1.weight" in keys and "blocks.1.16.weight" in keys and "deblocks.1.0.weight" in keys
    assert abs(dp.BEVBackbone.flops_per_frame(200, 176) / 1e9 - 39.4) < 1.0     # ~39.7 GFLOP


def test_cvae_matches_reference_generator():
    m = _load(dp.CVAE(4, 8), "cvae")
    pts, eps = torch.from_numpy(G["cvae_points"]), torch.from_numpy(G["cvae_eps"])
    cond = torch.from_numpy(G["cvae_cond"])
    with torch.no_grad():
        box = m.sample(pts, eps)
        _, mu, logvar = m.x_encoder(pts)
        _, _, kl, (mu_xy, logvar_xy, _, _) = m.posterior_prior(pts, cond)
        dec = m.obj_encoder(pts, eps)
    np.testing.assert_allclose(mu.numpy(), G["cvae_mu_x"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(logvar.numpy(), G["cvae_logvar_x"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(mu_xy.numpy(), G["cvae_mu_xy"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(logvar_xy.numpy(), G["cvae_logvar_xy"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dec.numpy(), G["cvae_dec"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(kl.numpy(), G["cvae_kl"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(box.numpy(), G["cvae_box"], rtol=1e-5, atol=1e-5)


def test_roi_fc_stack_and_anchor_head_shapes():
    fc = dp.RoIFCStack(96, grid_size=6).eval()
    keys = fc.state_dict().keys()
    for k in ("shared_fc_layer.0.weight", "shared_fc_layer.4.weight", "cls_fc_layers.0.weight",
              "cls_pred_layer.bias", "reg_fc_layers.5.running_mean", "reg_pred_layer.weight"):
        assert k in keys, k
    assert tuple(fc.shared_fc_layer[0].weight.shape) == (256, 20736)
    with torch.no_grad():
        cls, reg = fc(torch.randn(4, 216, 96))
    assert cls.shape == (4, 1) and reg.shape == (4, 7)
    head = dp.AnchorHead(256).eval()
    with torch.no_grad():
        d = head({"spatial_features_2d": torch.randn(1, 256, 10, 8)})
    assert d["cls_preds"].shape == (1, 10, 8, 18) and d["box_preds"].shape == (1, 10, 8, 42)
    assert d["dir_cls_preds"].shape == (1, 10, 8, 12)


def test_anchor_head_fused_convolution_equals_the_three_heads():
    """Training: conv_cls / conv_box / conv_dir_cls run as one convolution over the concatenated filters
    (AnchorHead._forward_fused) -- same outputs and the same gradients of every parameter and of the input as the
    three separate heads (anchor_head_single.py:41-58)."""
    torch.manual_seed(3)
    head = dp.AnchorHead(64, num_class=1, num_anchors_per_location=2).train()
    x = torch.randn(2, 64, 12, 9)
    res = []
    for fused in (True, False):
        head.FUSE_HEADS = fused
        head.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        d = head({"spatial_features_2d": xi})
        assert d["cls_preds"].shape == (2, 12, 9, 2) and d["box_preds"].shape == (2, 12, 9, 14)
        assert d["dir_cls_preds"].shape == (2, 12, 9, 4) and d["box_preds"].is_contiguous()
        loss = (d["cls_preds"] * 1.3).sum() + (d["box_preds"] ** 2).sum() + d["dir_cls_preds"].sin().sum()
        loss.backward()
        res.append(([d[k].detach() for k in ("cls_preds", "box_preds", "dir_cls_preds")], xi.grad,
                    {n: p.grad.clone() for n, p in head.named_parameters()}))
    head.FUSE_HEADS = True
    for a, b in zip(res[0][0], res[1][0]):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(res[0][1].numpy(), res[1][1].numpy(), rtol=1e-4, atol=1e-6)
    assert res[0][2].keys() == res[1][2].keys() and len(res[0][2]) == 6
    for n in res[0][2]:
        np.testing.assert_allclose(res[0][2][n].numpy(), res[1][2][n].numpy(), rtol=1e-4, atol=1e-5, err_msg=n)


def test_roi_grid_geometry_matches_reference_helpers():
    # the module under test needs no GPU for its geometry; import lazily (the package pulls torch ops)
    from glenet_amd import roi_grid as rg
    coords = torch.from_numpy(G["vc_coords"])
    for stride in (1, 2, 4, 8):
        c = rg.get_voxel_centers(coords, stride, [0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1])
        np.testing.assert_array_equal(c.numpy(), G["vc_centers_%d" % stride])
    out = rg.rotate_points_along_z(torch.from_numpy(G["rot_points"]), torch.from_numpy(G["rot_angle"]))
    np.testing.assert_allclose(out.numpy(), G["rot_out"], rtol=1e-6, atol=1e-6)
    # grid points: enumeration order and scaling of voxelrcnn_head.py:206-215
    rois = torch.tensor([[1.0, 2.0, 0.5, 4.0, 2.0, 1.5, 0.0]])
    glob, local = rg.global_grid_points_of_roi(rois, 2)
    assert local.shape == (1, 8, 3)
    np.testing.assert_allclose(local[0, 0].numpy(), [-1.0, -0.5, -0.375])
    np.testing.assert_allclose(local[0, 1].numpy(), [-1.0, -0.5, 0.375])     # z index runs fastest
    np.testing.assert_allclose(local[0, 4].numpy(), [1.0, -0.5, -0.375])
    np.testing.assert_allclose(glob[0, 0].numpy(), [0.0, 1.5, 0.125])


def test_cvae_training_losses_and_gradients_match_reference_golden():
    """cvae_train_ref.npz = the reference Generator's training branch + get_training_loss + autograd on CPU
    (tests/golden/make_golden.py:make_cvae_train_ref).  Our CVAE.training_losses loads its state dict by name and
    reproduces the three loss terms, the tb_dict parts, the decoder output, every parameter gradient and the
    BatchNorm running statistics after the step."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cvae_train_ref.npz"))
    m = dp.CVAE(4, 8).train()
    sd = {k[len("cvae/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("cvae/")}
    m.load_state_dict(sd, strict=True)
    T = lambda k: torch.from_numpy(g[k])                                                   # noqa: E731
    (reg, lat, regular), parts = m.training_losses(T("points"), T("cond"), T("labels"), eps_post=T("eps_post"))
    np.testing.assert_allclose(parts["box_pred_post"].detach().numpy(), g["box_pred_post"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(reg), float(g["reg_loss_post"]), rtol=1e-5)
    np.testing.assert_allclose(float(lat), float(g["lattent_loss"]), rtol=1e-5)
    np.testing.assert_allclose(float(regular), float(g["regular_loss"]), rtol=1e-5)
    for k in ("loss_loc", "loss_dir", "loss_reg"):
        np.testing.assert_allclose(float(parts[k]), float(g["tb/%s_post" % k]), rtol=1e-5)
    assert np.array_equal(dp.cvae_direction_target(T("labels"), 0.78539, 2).numpy(), g["dir_targets"].argmax(-1))
    (reg + lat + regular).backward()
    for name, p in m.named_parameters():
        want = g["grad/" + name]
        scale = np.abs(want).max() + 1e-12
        np.testing.assert_allclose(p.grad.numpy(), want, rtol=1e-4, atol=2e-5 * scale, err_msg=name)
    for k, v in m.state_dict().items():
        if "running_" in k or "num_batches" in k:
            np.testing.assert_allclose(v.numpy(), g["after/" + k], rtol=1e-5, atol=1e-6, err_msg=k)
