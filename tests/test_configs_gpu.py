"""BASELINE configs that round 1 left without a `-m gpu` test:
  * rows a21-a23 against the reference-generated golden (tests/golden/dense_path_ref.npz) ON THE DEVICE
    (MIOpen / hipBLASLt and the fused PointNet kernel instead of torch-CPU);
  * configs[3]: the CVAE at its full size -- 4096 objects x 512 points, 30 latent samples per object --
    against the golden's state dict run through plain fp32 torch modules (the unfused path) on the device;
  * configs[4]: a Waymo-shaped shard at full size (2 x 180 000 points, VoxelResBackBone8x) through the
    size-independent properties the domain offers: voxel coordinates unique / in range / first-seen order equal
    to the oracle's voxelizer (cheap at this size), every output cell of a strided conv has an active input
    in its window, the shape-static graph equals the exact-shape path, duplicated frames give duplicated rows.
  * configs[1]: the sparse convolutions' VALUES on the full-size active sets (4 x 20 000 points) against torch's dense
    conv3d on a window, plus linearity and the two adjoint identities.
"""
import os

import numpy as np
import pytest
import torch

import oracle
from glenet_amd import backbone as gb
from glenet_amd import dense_path as dp
from glenet_amd import synth

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dense_path_ref.npz"))


def _load(module, prefix, dev):
    sd = {k[len(prefix) + 1:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(prefix + "/")}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing)
    return module.to(dev).eval()


def test_bev_backbone_golden_on_device(dev):
    m = _load(dp.BEVBackbone(6, (1, 2), (1, 2), (8, 16), (1, 2), (8, 8)), "bev", dev)
    with torch.no_grad():
        y = m({"spatial_features": torch.from_numpy(G["bev_in"]).to(dev)})["spatial_features_2d"]
    np.testing.assert_allclose(y.cpu().numpy(), G["bev_out"], rtol=1e-4, atol=1e-4)


def test_cvae_golden_on_device_fused_extractor(dev):
    """Generator.forward's eval path with the PointNet extractors on the fused MFMA kernel."""
    m = _load(dp.CVAE(4, 8), "cvae", dev)
    pts, eps = torch.from_numpy(G["cvae_points"]).to(dev), torch.from_numpy(G["cvae_eps"]).to(dev)
    cond = torch.from_numpy(G["cvae_cond"]).to(dev)
    with torch.no_grad():
        assert m.x_encoder.fe._fusable(pts) and m._sample_fusable(pts)        # the hand-written kernels are the ones that run
        box = m.sample(pts, eps)
        dp.CVAE.FUSED_SAMPLER = False                         # module by module behind the extractors' kernels: the same boxes
        try:
            box_modules = m.sample(pts, eps)
        finally:
            dp.CVAE.FUSED_SAMPLER = True
        d = (box - box_modules).abs()
        d[:, 6] = torch.minimum(d[:, 6], (d[:, 6] - np.pi).abs())
        assert float(d.max()) < 1e-4, float(d.max())
        _, mu, logvar = m.x_encoder(pts)
        _, _, kl, (mu_xy, logvar_xy, _, _) = m.posterior_prior(pts, cond)
    for got, key in ((mu, "cvae_mu_x"), (logvar, "cvae_logvar_x"), (mu_xy, "cvae_mu_xy"), (logvar_xy, "cvae_logvar_xy"),
                     (box, "cvae_box")):
        np.testing.assert_allclose(got.cpu().numpy(), G[key], rtol=1e-4, atol=1e-4, err_msg=key)
    np.testing.assert_allclose(kl.cpu().numpy(), G["cvae_kl"], rtol=1e-3, atol=1e-3)


def _cvae_objects(n, seed=2000):
    return synth.cvae_objects(n, seed)                         # (n, 4, 512), SURVEY 8d


def test_cvae_config4_full_size_30_samples(dev):
    """4096 objects x 512 points x 30 latent samples (predict.sh:8-11): the fused path against plain fp32
    torch modules with the same (golden) weights; per-object results do not depend on the batch."""
    m = _load(dp.CVAE(4, 8), "cvae", dev)
    pts = torch.from_numpy(_cvae_objects(4096)).to(dev)
    gen = torch.Generator(device=dev).manual_seed(7)
    eps = torch.randn((30, 4096, 8), device=dev, generator=gen)
    with torch.no_grad():
        got = torch.stack([m.sample(pts, eps[s]) for s in range(30)])            # (30, 4096, 9)
        fused = dp.PointFeat._fusable
        dp.PointFeat._fusable = lambda self, x: False                             # unfused reference path
        dp.CVAE.FUSED_SAMPLER = False
        try:
            want = torch.stack([m.sample(pts[:512], eps[s, :512]) for s in (0, 17, 29)])
        finally:
            dp.PointFeat._fusable = fused
            dp.CVAE.FUSED_SAMPLER = True
        alone = m.sample(pts[100:101].contiguous(), eps[3, 100:101])
    assert got.shape == (30, 4096, 9) and torch.isfinite(got).all()
    # the decoded heading jumps by the bin period when the direction logits tie: compare modulo the period
    d = (got[[0, 17, 29], :512] - want).abs()
    d[..., 6] = torch.minimum(d[..., 6], (d[..., 6] - np.pi).abs())
    assert float(d.max()) < 2e-3, float(d.max())
    assert float((got[[0, 17, 29], :512, :6] - want[..., :6]).abs().max()) < 2e-4
    np.testing.assert_allclose(alone.cpu().numpy(), got[3, 100:101].cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert float(got.std(0)[:, :6].mean()) > 0                                    # the 30 samples differ


@pytest.mark.parametrize("path", ["default", "library_products", "bias_in_the_product", "fp32_wide_layer", "library_moments",
                                  "batchnorm_as_tensor_statements", "losses_as_tensor_statements", "h2_written",
                                  "backward_sums_as_a_pass", "narrow_extractor_layer_by_layer", "first_layer_on_the_row_kernels"])
def test_cvae_training_step_matches_reference_golden_on_device(dev, path, monkeypatch):
    """The training branch on the device (row kernels + fused training BatchNorm) against the reference-generated
    golden of tests/test_dense_path_cpu.py: loss terms, decoder output, every gradient, running statistics -- on the default
    path and with each of its class-level choices turned off (library products instead of csrc/glx_rows.hip; the conv biases added
    in the product instead of folded into the running means; the 128 -> 512 layer with fp32 MFMA products)."""
    if path == "library_products":
        monkeypatch.setattr(dp.PointFeat, "OWN_ROW_LAYERS", False)
    elif path == "bias_in_the_product":
        monkeypatch.setattr(dp.PointFeat, "BIAS_INTO_RUNNING_MEAN", False)
    elif path == "fp32_wide_layer":
        monkeypatch.setattr(dp.PointMaxBN, "F16X2", False)
    elif path == "library_moments":
        monkeypatch.setattr(dp.PointMaxBN, "OWN_MOMENTS", False)
    elif path == "batchnorm_as_tensor_statements":
        monkeypatch.setattr(dp.PointMaxBN, "FUSED_BN", False)
    elif path == "losses_as_tensor_statements":
        monkeypatch.setattr(dp.CVAE, "FUSED_LOSSES", False)
    elif path == "h2_written":
        monkeypatch.setattr(dp.PointFeat, "LAZY_H2", False)
    elif path == "narrow_extractor_layer_by_layer":      # the decoder's 8-wide extractor on the row kernels, not csrc/glx_narrowfeat.hip
        monkeypatch.setattr(dp.PointFeat, "NARROW_FUSED_TRAIN", False)
    elif path == "first_layer_on_the_row_kernels":       # conv1 + bn1 + relu of the wide extractors as a row layer, not from the points
        monkeypatch.setattr(dp.PointFeat, "LAYER1_FROM_POINTS", False)
    elif path == "backward_sums_as_a_pass":      # the second layer's BatchNorm-backward sums by glx_bn_backward_sums, not by dh2's producers
        monkeypatch.setattr(dp.PointMaxBN, "BWD_SUMS_IN_PRODUCERS", False)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cvae_train_ref.npz"))
    m = dp.CVAE(4, 8)
    m.load_state_dict({k[len("cvae/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("cvae/")}, strict=True)
    m = m.to(dev).train()
    T = lambda k: torch.from_numpy(g[k]).to(dev)                                          # noqa: E731
    assert m.x_encoder.fe._rows_trainable(T("points"))
    (reg, lat, regular), parts = m.training_losses(T("points"), T("cond"), T("labels"), eps_post=T("eps_post"))
    np.testing.assert_allclose(parts["box_pred_post"].detach().cpu().numpy(), g["box_pred_post"], rtol=1e-4, atol=1e-5)
    for got, key in ((reg, "reg_loss_post"), (lat, "lattent_loss"), (regular, "regular_loss")):
        np.testing.assert_allclose(float(got.detach()), float(g[key]), rtol=2e-5)
    (reg + lat + regular).backward()
    # biases in front of a BatchNorm have an analytically zero gradient (the mean subtraction removes them): what the
    # reference stores there is summation noise of ~1e-5, so the absolute floor follows the model's gradient scale
    top = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("grad/"))
    for name, p in m.named_parameters():
        want = g["grad/" + name]
        atol = 2e-4 * max(float(np.abs(want).max()), 1e-3 * top)
        np.testing.assert_allclose(p.grad.cpu().numpy(), want, rtol=2e-3, atol=atol, err_msg=name)
    for k, v in m.state_dict().items():
        if "running_" in k or "num_batches" in k:
            np.testing.assert_allclose(v.cpu().numpy(), g["after/" + k], rtol=1e-4, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("f16x2", [True, False, "statements"])
@pytest.mark.parametrize("B,P,neg", [(6, 100, False), (16, 512, True), (3, 130, True)])
def test_point_max_batchnorm_without_the_wide_tensor_equals_the_modules(dev, B, P, neg, f16x2, monkeypatch):
    """dense_path.PointMaxBN (csrc/glx_pointnet.hip: max / min / arg / moments in one pass over h2, backward through
    128 x 128 algebra) against Conv1d(128, 512, 1) + BatchNorm1d(512) in training mode + max over the points run by torch
    on the (B, 512, P) tensor (point_net.py:22-28): output, running statistics, gradients of input and parameters; negative
    BatchNorm weights take the min branch; P not a multiple of the 128-point pass.  Both arithmetics of the 128 -> 512 product
    (f16 x 2 with the statistics from the moments of h2; fp32 MFMA with the sums from the pass)."""
    monkeypatch.setattr(dp.PointMaxBN, "F16X2", bool(f16x2))
    monkeypatch.setattr(dp.PointMaxBN, "FUSED_BN", f16x2 is True)       # "statements": the f16 x 2 pass with the BatchNorm as tensor statements
    torch.manual_seed(B * P)
    conv, bn = torch.nn.Conv1d(128, 512, 1).to(dev), torch.nn.BatchNorm1d(512).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(512, device=dev) if neg else torch.rand(512, device=dev) + 0.5)
        bn.bias.copy_(torch.randn(512, device=dev))
    bn2 = torch.nn.BatchNorm1d(512).to(dev).train()
    bn2.load_state_dict(bn.state_dict())
    h = torch.relu(torch.randn(B * P, 128, device=dev) + 0.3).requires_grad_(True)
    gout = torch.randn(B, 512, device=dev)
    want = bn(conv(h.view(B, P, 128).transpose(1, 2))).amax(dim=2)
    want.backward(gout)
    ref = [t.grad.clone() for t in (h, conv.weight, conv.bias, bn.weight, bn.bias)]
    for t in (h, conv.weight, conv.bias, bn.weight, bn.bias):
        t.grad = None
    got = dp.PointMaxBN.apply(h, conv.weight[:, :, 0], conv.bias, bn2.weight, bn2.bias, bn2, B, P)
    got.backward(gout)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=2e-4, atol=2e-4)
    for k in ("running_mean", "running_var", "num_batches_tracked"):
        np.testing.assert_allclose(getattr(bn2, k).cpu().numpy(), getattr(bn, k).cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    for name, a, b in zip(("h2", "conv.weight", "conv.bias", "bn.weight", "bn.bias"),
                          (h.grad, conv.weight.grad[:, :, 0], conv.bias.grad, bn2.weight.grad, bn2.bias.grad),
                          (ref[0], ref[1][:, :, 0], ref[2], ref[3], ref[4])):
        scale = float(b.abs().max()) if name != "conv.bias" else float(ref[1].abs().max())   # the bias gradient is 0 + noise
        assert float((a - b).abs().max()) <= 2e-4 * scale + 1e-6, (name, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("bins", [2, 3])
def test_cvae_fused_losses_equal_the_tensor_statements(dev, bins):
    """glx_cvae_losses (regression + direction + KL terms and their gradients, one launch) against cvae_reg_loss + torch.distributions'
    KL with autograd: values to 1e-5, every gradient to 1e-5 of its scale; NaN targets, headings on both sides of the bins, large and
    small residuals (both smooth-L1 branches), non-unit incoming gradients."""
    g = torch.Generator(device=dev).manual_seed(bins)
    B, L = 1000, 8
    pred = (torch.randn(B, 7 + bins, device=dev, generator=g) * 0.5).requires_grad_(True)
    labels = torch.randn(B, 7, device=dev, generator=g) * 0.5
    labels[:, 6] = (torch.rand(B, device=dev, generator=g) - 0.5) * 12.0
    labels[::17, 2] = float("nan")
    pred.data[::5, :3] += labels[::5, :3].nan_to_num() + 0.01                 # small residuals: the quadratic branch
    mu1, lv1, mu2, lv2 = ((torch.randn(B, L, device=dev, generator=g) * 0.7).requires_grad_(True) for _ in range(4))
    w = dict(dp.CVAE.LOSS_WEIGHTS, code_weights=(1.0, 0.8, 1.2, 1.0, 0.5, 1.0, 2.0))
    cw = dp._code_weights(tuple(w["code_weights"]), dev)
    loc, dr, lat = dp.CvaeLosses.apply(pred, labels, mu1, lv1, mu2, lv2, cw, float(w["loc_weight"]), float(w["dir_weight"]),
                                       float(w["latent_weight"]), 0.78539, bins)
    (1.5 * loc + 0.5 * dr + 0.3 * lat).backward()
    got = [t.grad.clone() for t in (pred, mu1, lv1, mu2, lv2)]
    for t in (pred, mu1, lv1, mu2, lv2):
        t.grad = None
    mk = lambda m, lv: torch.distributions.Independent(torch.distributions.Normal(m, torch.exp(lv) + 3e-22), 1)       # noqa: E731
    lat_ref = torch.distributions.kl.kl_divergence(mk(mu1, lv1), mk(mu2, lv2)).mean() * w["latent_weight"]
    reg_ref, parts = dp.cvae_reg_loss(pred, labels, w, 0.78539, bins)
    (1.5 * parts["loss_loc"] + 0.5 * parts["loss_dir"] + 0.3 * lat_ref).backward()
    want = [t.grad for t in (pred, mu1, lv1, mu2, lv2)]
    for a, b in ((loc, parts["loss_loc"]), (dr, parts["loss_dir"]), (lat, lat_ref)):
        assert abs(float(a) - float(b)) <= 1e-5 * max(1.0, abs(float(b))), (float(a), float(b))
    for a, b in zip(got, want):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9


@pytest.mark.parametrize("cin,B,P", [(4, 37, 77), (5, 3, 300), (8, 600, 16), (3, 1, 1000)])
def test_narrow_extractor_training_pass_without_intermediates(dev, cin, B, P):
    """PointFeat(C, (8, 8, 8)) in training mode through csrc/glx_narrowfeat.hip (batch statistics from moments, every pass recomputed
    from the points) against the same modules in fp64 autograd: the output, every parameter gradient (the convolutions' biases: exact
    zeros), the running statistics and the batch counters; 3 .. 8 point features, fewer and more objects than the passes' 512 blocks,
    objects of 16 and 1000 points, duplicated points (ties go to the lower point), a channel the first ReLU shuts."""
    import copy
    torch.manual_seed(cin * 31 + B)
    m = dp.PointFeat(cin, (8, 8, 8)).to(dev).train()
    with torch.no_grad():
        for bn in (m.bn1, m.bn2, m.bn3):
            bn.weight.copy_(torch.rand(8, device=dev) + 0.5)
            bn.bias.copy_(torch.randn(8, device=dev) * 0.3)
            bn.running_mean.copy_(torch.randn(8, device=dev) * 0.1)
            bn.running_var.copy_(torch.rand(8, device=dev) + 0.5)
        m.bn1.bias[2] = -30.0                          # channel 2 of layer 1: the ReLU passes nothing
    ref = copy.deepcopy(m).double()
    x = torch.randn(B, cin, P, device=dev)
    x[0, :, 1] = x[0, :, 0]                              # a duplicated point
    gout = torch.randn(B, 8, device=dev)
    assert m._narrow_trainable(x)
    out = m(x)
    (out * gout).sum().backward()
    yd = ref(x.double())
    (yd * gout.double()).sum().backward()
    torch.cuda.synchronize()
    scale = float(yd.detach().abs().max())
    assert float((out.double() - yd).abs().max()) < 2e-5 * scale
    for (name, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if name.startswith("conv") and name.endswith("bias"):
            assert float(p.grad.abs().max()) == 0.0, name
            assert float(q.grad.abs().max()) < 1e-9 * max(1.0, float(gout.abs().sum())), name      # (zero in exact arithmetic)
            continue
        tol = 2e-4 * float(q.grad.abs().max()) + 1e-6
        assert float((p.grad.double() - q.grad).abs().max()) < tol, (name, float((p.grad.double() - q.grad).abs().max()), tol)
    for bn, bd in ((m.bn1, ref.bn1), (m.bn2, ref.bn2), (m.bn3, ref.bn3)):
        np.testing.assert_allclose(bn.running_mean.cpu().numpy(), bd.running_mean.float().cpu().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(bn.running_var.cpu().numpy(), bd.running_var.float().cpu().numpy(), rtol=1e-4, atol=1e-6)
        assert int(bn.num_batches_tracked) == int(bd.num_batches_tracked) == 1


@pytest.mark.parametrize("cin,B,P", [(4, 37, 77), (5, 3, 300), (8, 300, 16), (3, 1, 1000)])
def test_first_point_layer_from_the_points(dev, cin, B, P):
    """dense_path.PointLayer1Train (conv1 + bn1 + relu of the wide extractor in training mode: statistics from the moments of x, h1
    written in one pass, the backward's sums in one pass over its gradient) against the modules in fp64 autograd: h1, the weight /
    gamma / beta gradients (the bias: exact zeros), running statistics; a channel the ReLU shuts and one it never does."""
    torch.manual_seed(cin * 17 + P)
    conv, bn = torch.nn.Conv1d(cin, 64, 1).to(dev), torch.nn.BatchNorm1d(64).to(dev).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(64, device=dev) + 0.5)
        bn.bias.copy_(torch.randn(64, device=dev) * 0.3)
        bn.bias[7], bn.bias[9] = -40.0, 40.0
    import copy
    cd, bd = copy.deepcopy(conv).double(), copy.deepcopy(bn).double()
    x = torch.randn(B, cin, P, device=dev) * 2.0 + 0.5
    g = torch.randn(B * P, 64, device=dev)
    h1 = dp.PointLayer1Train.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn)
    (h1 * g).sum().backward()
    hd = torch.relu(bd(cd(x.double()))).transpose(1, 2).reshape(B * P, 64)
    (hd * g.double()).sum().backward()
    torch.cuda.synchronize()
    assert float((h1.detach().double() - hd.detach()).abs().max()) < 2e-5 * float(hd.detach().abs().max())
    assert float(conv.bias.grad.abs().max()) == 0.0
    for name, a, b in (("weight", conv.weight.grad, cd.weight.grad), ("gamma", bn.weight.grad, bd.weight.grad),
                       ("beta", bn.bias.grad, bd.bias.grad)):
        tol = 2e-4 * float(b.abs().max()) + 1e-6
        assert float((a.double() - b).abs().max()) < tol, (name, float((a.double() - b).abs().max()), tol)
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), bd.running_mean.float().cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), bd.running_var.float().cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,P", [(5, 77), (3, 8200), (2, 1)])
def test_pointmax_scatter_forms_against_index_add(dev, B, P):
    """glx_pointmax_scatter (every row written: init + the rows' sums) and glx_pointmax_scatter_add(_scaled) (only the rows some channel
    points at move: found through the LDS bitmap for P <= 8192, by scanning all rows beyond) against torch.index_add in fp64:
    coefficients that are exactly zero (their rows do not move), channels sharing a row, a one-point object."""
    import ctypes  # noqa: F401
    from glenet_amd import _lib
    torch.manual_seed(B * 13 + P)
    arg = torch.randint(0, P, (B, 512), device=dev, dtype=torch.int32)
    arg[0, :40] = arg[0, 0]                              # forty channels at one row
    g = torch.randn(B, 512, device=dev)
    g[:, 5::7] = 0.0                                     # exact zeros: those channels add nothing
    cs = torch.rand(512, device=dev) + 0.5
    W3 = torch.randn(512, 128, device=dev) * 0.2
    init = torch.randn(128, device=dev)
    base = torch.randn(B * P, 128, device=dev)
    rows = (torch.arange(B, device=dev)[:, None] * P + arg.long()).reshape(-1)
    add = torch.zeros(B * P, 128, dtype=torch.float64, device=dev)
    add.index_add_(0, rows, ((g * cs).reshape(-1, 1).double() * W3.double().repeat(B, 1)))
    got = torch.empty(B * P, 128, device=dev)
    _lib.call("glx_pointmax_scatter", arg, (g * cs).contiguous(), W3, init, B, P, got)
    want = init.double()[None, :] + add
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) < 2e-5 * scale
    for scaled in (False, True):
        acc = base.clone()
        if scaled:
            _lib.call("glx_pointmax_scatter_add_scaled", arg, g, cs, W3, B, P, acc)
        else:
            _lib.call("glx_pointmax_scatter_add", arg, (g * cs).contiguous(), W3, B, P, acc)
        torch.cuda.synchronize()
        want = base.double() + add
        assert float((acc.double() - want).abs().max()) < 2e-5 * float(want.abs().max())
        moved = torch.zeros(B * P, dtype=torch.bool, device=dev)
        moved[rows[(g * cs).reshape(-1) != 0]] = True
        assert torch.equal(acc[~moved], base[~moved])            # rows nobody points at: untouched, bit for bit


def test_batchnorm_backward_sums_taken_by_the_gradients_producers(dev):
    """glx_rows128_affine_f16x2_sums + glx_pointmax_scatter_add_scaled_sums + glx_bn_backward_from_partials against
    glx_bn_backward_sums on the finished gradient: the same coef3 / dgamma / dbeta to rounding, the gradient itself bit for bit (the sums
    ride along, they change nothing); rows that no channel points at, objects whose extremes share a row, a channel the ReLU shuts
    everywhere and a row count that is not a multiple of the kernels' 32."""
    import ctypes
    from glenet_amd import _lib
    from glenet_amd.spconv import core
    torch.manual_seed(11)
    B, P = 37, 77
    R = B * P
    z = torch.randn(R, 128, device=dev)
    gamma, beta = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.3
    beta[5] = -50.0                                     # channel 5: the ReLU passes nothing
    mean, var = z.mean(0), z.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    pre = torch.cat([scale, shift]).contiguous()
    M = torch.randn(128, 128, device=dev) * 0.05
    mh, em = dp.PointFeat._f16x2_image(M.t(), scale=-1.0)
    nv = torch.randn(128, device=dev) * 0.1
    arg = torch.randint(0, P, (B, 512), device=dev, dtype=torch.int32)
    arg[3] = 7                                          # object 3: every extreme at the same row
    gext = torch.randn(B, 512, device=dev)
    cscale = torch.rand(512, device=dev) + 0.5
    W3 = torch.randn(512, 128, device=dev) * 0.1
    d_plain = torch.empty_like(z)
    _lib.call("glx_rows128_affine_f16x2", z, ctypes.c_longlong(R), mh, em, nv, d_plain, pre)
    _lib.call("glx_pointmax_scatter_add_scaled", arg, gext, cscale, W3, B, P, d_plain)
    na = int(_lib.load().glx_rows128_affine_blocks(ctypes.c_longlong(R)))
    pa, pb = torch.full((na, 2, 128), 9.0, device=dev), torch.full((B, 2, 128), 9.0, device=dev)
    d_sums = torch.empty_like(z)
    _lib.call("glx_rows128_affine_f16x2_sums", z, ctypes.c_longlong(R), mh, em, nv, d_sums, pre, pa)
    _lib.call("glx_pointmax_scatter_add_scaled_sums", arg, gext, cscale, W3, B, P, d_sums, z, pre, pb)
    assert torch.equal(d_plain, d_sums)
    got = [torch.empty(n, device=dev) for n in (128, 128, 384)]
    _lib.call("glx_bn_backward_from_partials", pa, na, pb, B, 128, ctypes.c_longlong(R), gamma, mean, invstd, *got)
    want = [torch.empty(n, device=dev) for n in (128, 128, 384)]
    _lib.call("glx_bn_backward_sums", z, d_plain, R, 128, gamma, beta, mean, invstd, 1, want[0], want[1], None, want[2],
              core._bn_state(dev))
    torch.cuda.synchronize()
    dz = torch.where(z * scale + shift > 0, d_plain, torch.zeros_like(d_plain)).double()
    ref_beta, ref_gamma = dz.sum(0), (dz * ((z.double() - mean.double()) * invstd.double())).sum(0)
    mag = dz.abs().sum(0).clamp_min(1e-30)
    assert float(got[1][5]) == 0.0 and float(got[0][5]) == 0.0
    for name, a, b, r in (("dgamma", got[0], want[0], ref_gamma), ("dbeta", got[1], want[1], ref_beta)):
        assert float(((a.double() - r).abs() / mag).max()) < 2e-6, name           # against fp64
        assert float(((b.double() - r).abs() / mag).max()) < 2e-6, name
    np.testing.assert_allclose(got[2].cpu().numpy(), want[2].cpu().numpy(), rtol=1e-4, atol=1e-7)


def test_cvae_train_step_graph_follows_an_eager_loop(dev):
    """CVAETrainStep (one HIP graph: forward, backward, clip 10, AdamW on flat buffers) against zero_grad / backward /
    clip_grad_norm_ / torch.optim.AdamW on a copy of the model run through the unfused (B, C, P) modules."""
    import copy
    from glenet_amd import cvae_train as ct
    torch.manual_seed(0)
    B, P = 256, 128
    pts, box8, box7 = (torch.from_numpy(a).to(dev) for a in synth.cvae_objects(B, 5, P, with_labels=True))
    model = dp.CVAE(4, 8).to(dev).train()
    ref = copy.deepcopy(model)
    gen = torch.Generator(device=dev).manual_seed(1)
    eps = [torch.randn((B, 8), device=dev, generator=gen) for _ in range(4)]
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, betas=ct.OPTIM_CFG["BETAS"], weight_decay=ct.OPTIM_CFG["WEIGHT_DECAY"])
    rows = dp.PointFeat._rows_trainable
    dp.PointFeat._rows_trainable = lambda self, x: False
    want = []
    try:
        for e in eps:
            opt.zero_grad(set_to_none=True)
            (reg, lat, regular), _ = ref.training_losses(pts, box8, box7, eps_post=e)
            loss = reg + lat + regular
            loss.backward()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), ct.OPTIM_CFG["GRAD_NORM_CLIP"])
            opt.step()
            want.append(float(loss))
    finally:
        dp.PointFeat._rows_trainable = rows
    step = ct.CVAETrainStep(model, B, P, lr=1e-3)
    step.load(pts, box8, box7, eps[0])
    step.capture()
    got = []
    for e in eps:
        step.load(pts, box8, box7, e)
        step.step()
        got.append(float(step.loss))
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4)                  # same weights: capture() restored them
    np.testing.assert_allclose(got, want, rtol=2e-2)
    assert int(step.optimizer.step_count) == 4


def test_cvae_step_regulariser_from_the_flat_buffers_equals_autograds(dev, monkeypatch):
    """cvae_train.CVAETrainStep with the weight regulariser taken from the optimizer's flat buffers (glx_flat_l2_norms /
    glx_flat_l2_norm_grad_add, the default) against the same step with it behind autograd: the same loss and, after two eager
    steps, the same parameters to rounding -- and switching the flag off on a live step does not add a stale gradient."""
    import copy
    from glenet_amd import cvae_train as ct
    torch.manual_seed(3)
    B, P = 128, 64
    pts, box8, box7 = (torch.from_numpy(a).to(dev) for a in synth.cvae_objects(B, 7, P, with_labels=True))
    base = dp.CVAE(4, 8).to(dev).train()
    eps = torch.randn((B, 8), device=dev)
    out = {}
    for flat in (True, False):
        monkeypatch.setattr(ct.CVAETrainStep, "FLAT_REGULARISER", flat)
        step = ct.CVAETrainStep(copy.deepcopy(base), B, P, lr=1e-3)
        step.load(pts, box8, box7, eps)
        losses = [float(step.enqueue()) for _ in range(2)]
        if flat:                                   # flag off on the live object: the autograd path, nothing pending behind it
            monkeypatch.setattr(ct.CVAETrainStep, "FLAT_REGULARISER", False)
            step.enqueue()
            assert step._regulariser_pending is False
            out["third_flat_then_autograd"] = step.optimizer.flat_param.detach().clone()
        else:
            step.enqueue()
            out["third_autograd"] = step.optimizer.flat_param.detach().clone()
        out[flat] = losses
        torch.cuda.synchronize()
    np.testing.assert_allclose(out[True], out[False], rtol=1e-5)
    a, b = out["third_flat_then_autograd"], out["third_autograd"]
    assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


def test_waymo_shaped_full_size_shard(dev):
    """configs[4] per GPU: 2 frames x 180 000 points, 5 features, VoxelResBackBone8x."""
    W = synth.WAYMO
    torch.manual_seed(6)
    grid = oracle.grid_size_of(W["point_cloud_range"], W["voxel_size"])
    model = gb.SparseBackbone8x(5, grid, residual=True).to(dev).eval()
    f0 = synth.waymo_frame(11)[0]
    frames = [f0, f0.copy()]                                                     # the same cloud twice
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    with torch.no_grad():
        bd = gb.voxelize_batch(pts, bidx, 2, W, train=False)
        v, c, n = oracle.voxelize_hard_batch(frames, W["voxel_size"], W["point_cloud_range"], 5, 150000)
        coords = bd["voxel_coords"].cpu().numpy()
        assert np.array_equal(coords, c) and np.array_equal(bd["voxels"].cpu().numpy(), v)   # bit-exact at full size
        shape = np.array([2, 41, grid[1], grid[0]])
        assert (coords >= 0).all() and (coords < shape).all()
        assert len(np.unique(coords, axis=0)) == len(coords)
        bd = gb.HeightCompression()(model(gb.MeanVFE()(bd)))
    # every output cell of the strided convs is reached by an active input; rows ascend in (b, z, y, x)
    feats = bd["multi_scale_3d_features"]
    prev = coords
    for name, pad in (("x_conv2", (1, 1, 1)), ("x_conv3", (1, 1, 1)), ("x_conv4", (0, 1, 1))):
        st = feats[name]
        out = st.indices.cpu().numpy().astype(np.int64)
        key = ((out[:, 0] * 64 + out[:, 1]) * 4096 + out[:, 2]) * 4096 + out[:, 3]
        assert (np.diff(key) > 0).all()
        # candidate outputs of every input cell under k=3, s=2: o = (i + pad - k) / 2 where divisible
        cand = []
        pin = prev.astype(np.int64)
        for kz in range(3):
            for ky in range(3):
                for kx in range(3):
                    nz, ny, nx = pin[:, 1] + pad[0] - kz, pin[:, 2] + pad[1] - ky, pin[:, 3] + pad[2] - kx
                    ok = (nz >= 0) & (ny >= 0) & (nx >= 0) & (nz % 2 == 0) & (ny % 2 == 0) & (nx % 2 == 0)
                    oz, oy, ox = nz // 2, ny // 2, nx // 2
                    ok &= (oz < st.spatial_shape[0]) & (oy < st.spatial_shape[1]) & (ox < st.spatial_shape[2])
                    cand.append(((pin[ok, 0] * 64 + oz[ok]) * 4096 + oy[ok]) * 4096 + ox[ok])
        assert np.array_equal(key, np.unique(np.concatenate(cand))), name
        prev = out
    # the two identical frames give identical rows
    o = bd["encoded_spconv_tensor"]
    idx, f = o.indices.cpu().numpy(), o.features.cpu().numpy()
    a, b = idx[:, 0] == 0, idx[:, 0] == 1
    assert a.sum() == b.sum() and np.array_equal(idx[a][:, 1:], idx[b][:, 1:])
    np.testing.assert_allclose(f[a], f[b], rtol=1e-5, atol=1e-6)
    dense = bd["spatial_features"]
    assert dense.shape[0] == 2 and torch.equal(dense[0] != 0, dense[1] != 0)
    # shape-static graph == exact shapes at this size
    pipe = gb.StaticFramePipeline(model, W, 2, pts.shape[0], 5, train_voxel_cap=False)
    pipe.calibrate(pts, bidx)
    pipe.load(pts, bidx)
    pipe.capture()
    out = pipe.replay()
    torch.cuda.synchronize()
    pipe.check()
    np.testing.assert_allclose(out["spatial_features"].cpu().numpy(), dense.cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_bev_backbone_training_nhwc_fused_bn_equals_torch_path(dev):
    """BEVBackbone in training mode on a channels-last map: ZeroPad2d + conv folded into one padded conv and
    BatchNorm2d + ReLU on the fused row kernels == the plain nn.Sequential modules on the same NCHW data: output,
    input gradient, every parameter gradient, running statistics."""
    import copy
    from glenet_amd.spconv import core
    torch.manual_seed(3)
    a = dp.BEVBackbone(16, (2, 2), (1, 2), (32, 64), (1, 2), (32, 32)).to(dev).train()
    for m in a.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    b = copy.deepcopy(a)
    x0 = torch.randn(3, 16, 24, 20, device=dev) * (torch.rand(3, 1, 24, 20, device=dev) < 0.3)
    res = []
    for mod, fused in ((a, True), (b, False)):
        x = (x0.contiguous(memory_format=torch.channels_last) if fused else x0.clone()).requires_grad_(True)
        if fused:
            assert dp.BEVBackbone._can_fuse_bn(mod.blocks[0][2], mod.blocks[0][1](torch.nn.functional.pad(x, (1, 1, 1, 1))))
            y = mod({"spatial_features": x})["spatial_features_2d"]
        else:
            h, ups = x, []
            for i, blk in enumerate(mod.blocks):          # the reference's statement sequence (base_bev_backbone.py:88-104)
                h = blk(h)
                ups.append(mod.deblocks[i](h))
            y = torch.cat(ups, dim=1)
        (y * torch.linspace(0.5, 1.5, y.shape[1], device=dev).view(1, -1, 1, 1)).square().mean().backward()
        res.append((y.detach(), x.grad.clone(), {n: p.grad.clone() for n, p in mod.named_parameters()},
                    {n: t.clone() for n, t in mod.named_buffers()}))
    (y1, g1, p1, b1), (y2, g2, p2, b2) = res
    np.testing.assert_allclose(y1.cpu().numpy(), y2.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g1.cpu().numpy(), g2.cpu().numpy(), rtol=1e-3, atol=1e-7 + 1e-4 * float(g2.abs().max()))
    for k in p1:
        np.testing.assert_allclose(p1[k].cpu().numpy(), p2[k].cpu().numpy(), rtol=2e-3,
                                   atol=1e-7 + 2e-4 * float(p2[k].abs().max()), err_msg=k)
    for k in b1:
        np.testing.assert_allclose(b1[k].cpu().numpy(), b2[k].cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)


def test_config1_full_size_sparse_conv_values(dev):
    """configs[1] at its full size (4 frames x 20 000 points): the values of the sparse convolutions on the real active sets,
    checked without the oracle (it would take minutes here) by what does not depend on the size --
      * an independent formulation: torch's dense conv3d on a 41 x 128 x 128 window of frame 0, compared at every
        active output cell whose receptive field lies inside the window (submanifold and strided conv);
      * linearity in the features;  * <conv(x), g> = <x, dgrad(g)> and <dW, V> = <conv_V(x), g> (fp64 inner products)."""
    from glenet_amd.spconv import core as sp
    K = synth.KITTI
    frames = [synth.kitti_frame(1000 + i)[0] for i in range(4)]
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
    grid = oracle.grid_size_of(K["point_cloud_range"], K["voxel_size"])
    with torch.no_grad():
        bd = gb.MeanVFE()(gb.voxelize_batch(pts, bidx, 4, K))
    coords = bd["voxel_coords"]
    shape = [grid[2] + 1, grid[1], grid[0]]
    assert coords.shape[0] > 50000
    torch.manual_seed(3)
    gen = torch.Generator(device="cpu").manual_seed(5)
    y0, x0, win = 736, 0, 128                                        # window of frame 0 next to the sensor
    for cin, cout, subm in ((16, 16, True), (16, 32, False), (64, 64, True)):
        feats = torch.randn((coords.shape[0], cin), generator=gen).to(dev)
        if subm:
            conv = sp.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="p%d" % cin).to(dev)
        else:
            conv = sp.SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=False, indice_key="q%d" % cin).to(dev)
        x = sp.SparseConvTensor(feats.clone().requires_grad_(True), coords, shape, 4)
        y = conv(x)
        out_idx = y.indices.long()
        # ---- dense conv3d on the window
        c = coords.long()
        inw = (c[:, 0] == 0) & (c[:, 2] >= y0) & (c[:, 2] < y0 + win) & (c[:, 3] >= x0) & (c[:, 3] < x0 + win)
        assert int(inw.sum()) > 3000
        dense = torch.zeros((1, cin, shape[0], win, win), device=dev)
        ci = c[inw]
        dense[0, :, ci[:, 1], ci[:, 2] - y0, ci[:, 3] - x0] = feats[inw].t()
        wt = conv.weight.detach().permute(4, 3, 0, 1, 2).contiguous()
        s = 1 if subm else 2
        ref = torch.nn.functional.conv3d(dense, wt, stride=s, padding=1)
        o = out_idx
        # output cell (z, y, x) reads inputs s*o - 1 .. s*o + 1: inside the window when 1 <= s*(o - origin/s) <= win - 2
        oy, ox = o[:, 2] * s - y0, o[:, 3] * s - x0
        inside = (o[:, 0] == 0) & (oy >= 1) & (oy <= win - 2) & (ox >= 1) & (ox <= win - 2)
        assert int(inside.sum()) > 1000
        oi = o[inside]
        want = ref[0, :, oi[:, 1], (oi[:, 2] * s - y0) // s, (oi[:, 3] * s - x0) // s].t()
        got = y.features.detach()[inside]
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=2e-4)
        # ---- linearity
        with torch.no_grad():
            f2 = torch.randn((coords.shape[0], cin), generator=gen).to(dev)
            y2 = conv(sp.SparseConvTensor(f2, coords, shape, 4)).features
            y3 = conv(sp.SparseConvTensor(2.0 * feats - 3.0 * f2, coords, shape, 4)).features
        np.testing.assert_allclose(y3.cpu().numpy(), (2.0 * y.features.detach() - 3.0 * y2).cpu().numpy(), rtol=1e-4, atol=2e-4)
        # ---- adjoints (fp64 inner products)
        g = torch.randn(y.features.shape, generator=gen).to(dev)
        y.features.backward(g)
        lhs = float((y.features.detach().double() * g.double()).sum())
        rhs = float((feats.double() * x.features.grad.double()).sum())
        assert abs(lhs - rhs) <= 1e-5 * (abs(lhs) + float(y.features.detach().double().abs().mul(g.double().abs()).sum()) * 1e-2), (lhs, rhs)
        v = torch.randn(conv.weight.shape, generator=gen).to(dev)
        dwv = float((conv.weight.grad.double() * v.double()).sum())
        with torch.no_grad():
            w0 = conv.weight.detach().clone()
            conv.weight.copy_(v)
            yv = conv(sp.SparseConvTensor(feats, coords, shape, 4)).features
            conv.weight.copy_(w0)
        ref_dwv = float((yv.double() * g.double()).sum())
        scale = float((yv.double().abs() * g.double().abs()).sum())
        assert abs(dwv - ref_dwv) <= 1e-5 * scale, (dwv, ref_dwv, scale)
