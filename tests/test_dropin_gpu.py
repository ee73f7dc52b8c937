"""glenet_amd.dropin on the device: (i) a training step under dropin.reference_layout() -- the reference's module
layout driven through the drop-in's operators only: vendor 2-D convolutions on an NCHW map from dense(), per-frame proposal
loop, RoI-grid pooling through VoxelQueryAndGrouping / grouping_operation and Conv modules -- computes what the fused /
batched paths compute; (ii) dropin.accelerate() re-classes modules that carry the reference's class names and attribute
layout (stand-ins defined here: /root/reference does not exist on the GPU box), keeps every state-dict key and leaves the
results where they were."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

from glenet_amd import dropin, synth

pytestmark = pytest.mark.gpu
K = dict(synth.KITTI, point_cloud_range=[0.0, -16.0, -3.0, 35.2, 16.0, 1.0])
JIT = [0.2, -0.15, 0.05, 0.1, -0.05, 0.03, 0.08]


def _batch(dev, ids, n):
    frames = [synth.kitti_frame(i, num_points=n) for i in ids]
    r = K["point_cloud_range"]
    pts, bidx, gts = [], [], []
    for b, (p, bx) in enumerate(frames):
        keep = (p[:, 0] < r[3]) & (np.abs(p[:, 1]) < r[4])
        pts.append(p[keep])
        bidx.append(np.full(int(keep.sum()), b, np.int32))
        gts.append(bx[(bx[:, 0] < r[3] - 3) & (np.abs(bx[:, 1]) < r[4] - 3)])
    G = max(1, max(len(g) for g in gts))
    gt, unc = np.zeros((len(ids), G, 8), np.float32), np.zeros((len(ids), G, 7), np.float32)
    for b, g in enumerate(gts):
        gt[b, :len(g), :7], gt[b, :len(g), 7] = g, 1
        unc[b, :len(g)] = 0.05
    t = lambda a: torch.from_numpy(a).to(dev)                                       # noqa: E731
    return t(np.concatenate(pts)), t(np.concatenate(bidx)), t(gt), t(unc)


def _step(model, batch, dev, draws):
    pts, bidx, gt, unc = batch
    model.zero_grad(set_to_none=True)
    model.fixed_draws = draws
    loss, parts = model.training_step(pts, bidx, gt.shape[0], gt, unc, seed_rois_with_gt=torch.tensor(JIT, device=dev))
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    out = {k: float(v) for k, v in parts.items()}
    own = tuple(t.clone() for t in model.last["own_proposals"])
    model.last = None
    return out, grads, own


def test_reference_layout_step_equals_the_fused_step(dev):
    from glenet_amd import glenet_vr as gvr
    torch.backends.cudnn.benchmark = False
    torch.manual_seed(0)
    fast = gvr.GLENetVR(K).to(dev).train()
    state = copy.deepcopy(fast.state_dict())
    batch = _batch(dev, [60, 61], 8000)
    R, P = fast.roi_cfg["NMS_TRAIN"][1], fast.roi_cfg["TARGET"]["ROI_PER_IMAGE"]
    g = torch.Generator(device=dev).manual_seed(1)
    draws = (torch.rand((2, R), device=dev, generator=g), torch.rand((2, P), device=dev, generator=g))
    for m in fast.roi_head.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    want, wgrads, props = _step(fast, batch, dev, draws)
    with dropin.reference_layout() as layout:
        assert len(layout.saved) >= 12
        from glenet_amd import dense_path as dp, detector as det, roi_grid as rg
        assert not dp.OWN_CONV3X3 and not det.BATCHED_PROPOSALS and not rg.RoIGridPool.USE_ROWS
        slow = gvr.GLENetVR(K, bev_channels_last=False).to(dev).train()
        slow.load_state_dict(state)
        for m in slow.roi_head.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        assert not slow.map_to_bev_module.defer
        # the two flows' head maps differ by rounding (vendor vs own convolutions): a 1e-7 score difference may flip the order of
        # two proposals, which changes the sampled RoIs -- the fused flow's proposals are handed to this one
        slow.fixed_proposals = props
        got, ggrads, own = _step(slow, batch, dev, draws)
        ng = batch[2].shape[1]                                       # the first slots were overwritten by the seeding
        same = float(((own[0][:, ng:] - props[0][:, ng:]).abs().amax(-1) < 1e-3).float().mean())
        assert same > 0.9, "the per-frame proposal loop reproduces %.3f of the batched proposals" % same
    from glenet_amd import dense_path as dp
    assert dp.OWN_CONV3X3                                                          # switches restored
    for k, v in want.items():
        np.testing.assert_allclose(got[k], v, rtol=5e-4, atol=1e-6, err_msg=k)
    assert ggrads.keys() == wgrads.keys()
    rel = sorted(float((ggrads[k] - v).abs().max()) / (float(v.abs().max()) + 1e-12) for k, v in wgrads.items())
    assert rel[len(rel) // 2] < 1e-3 and rel[int(len(rel) * 0.9)] < 1e-2, (rel[len(rel) // 2], rel[-1])


class HeightCompression(nn.Module):
    """Stand-in with the reference's attribute layout (height_compression.py:4-26)."""

    def __init__(self, num_bev_features):
        super().__init__()
        self.num_bev_features = num_bev_features

    def forward(self, batch_dict):
        d = batch_dict["encoded_spconv_tensor"].dense()
        n, c, dd, h, w = d.shape
        batch_dict["spatial_features"] = d.view(n, c * dd, h, w)
        batch_dict["spatial_features_stride"] = batch_dict["encoded_spconv_tensor_stride"]
        return batch_dict


def test_accelerate_reclasses_reference_layout_modules_and_keeps_results(dev):
    from glenet_amd import backbone as gb, dense_path as dp, glenet_vr as gvr, roi_targets as rt
    from glenet_amd.pcdet_ops.pointnet2.pointnet2_stack import voxel_pool_modules as vpm
    torch.manual_seed(0)
    model = gvr.GLENetVR(K, bev_channels_last=False).to(dev).eval()
    # dress three modules as the reference's: same names, same attributes, plain classes
    model.map_to_bev_module = HeightCompression(256)
    model.backbone_2d.__class__ = type("BaseBEVBackbone", (nn.Module,), {})
    msg = type("NeighborVoxelSAModuleMSG", (nn.Module,), {})
    ptl = type("ProposalTargetLayer", (nn.Module,), {})
    for layer in model.roi_head.roi_grid_pool_layers:
        layer.__class__ = msg
    model.target_layer.__class__ = ptl
    keys = list(model.state_dict().keys())
    changed = dropin.accelerate(model)
    assert "backbone_2d" in changed and "map_to_bev_module" in changed and "target_layer" in changed
    assert isinstance(model.backbone_2d, dp.BEVBackbone)
    assert sum(c.startswith("roi_head.roi_grid_pool_layers.") for c in changed) == 3
    assert isinstance(model.map_to_bev_module, gb.HeightCompression) and model.map_to_bev_module.channels_last
    assert all(isinstance(l, vpm.NeighborVoxelSAModuleMSG) for l in model.roi_head.roi_grid_pool_layers)
    assert isinstance(model.target_layer, rt.ProposalTargetLayer)
    assert list(model.state_dict().keys()) == keys
    # the re-classed flow computes what an untouched model computes
    ref = gvr.GLENetVR(K).to(dev).eval()
    ref.load_state_dict(model.state_dict())
    pts, bidx, _, _ = _batch(dev, [62, 63], 8000)
    a, b = model(pts, bidx, 2), ref(pts, bidx, 2)
    for k in ("batch_cls_preds", "batch_box_preds", "batch_box_std_preds"):
        scale = float(b[k].abs().max()) + 1e-9
        assert float((a[k] - b[k]).abs().max()) <= 2e-4 * scale, k
    assert torch.equal(a["rois"], b["rois"])


def test_pointwise_as_gemm_equals_the_vendor_layers_and_is_undone(dev):
    """dropin.pointwise_as_gemm(): 1 x 1 Conv1d / Conv2d modules as matrix products and training-mode BatchNorm1d / 2d of
    stacked tensors on the channel-major kernels (no per-voxel-count preparation in the vendor library:
    voxel_pool_modules.py:70-130 feeds them (1, C, M) and (1, C, M, nsample) tensors whose M changes every step) -- outputs,
    input gradients, parameter gradients and running statistics equal the native kernels'; 3-tap layers keep the original
    forward; the switch is undone."""
    import time
    torch.manual_seed(0)
    conv_f = nn.Conv1d.forward
    c1 = nn.Sequential(nn.Conv1d(16, 32, 1, bias=False), nn.BatchNorm1d(32), nn.ReLU()).to(dev).train()
    c2 = nn.Sequential(nn.Conv2d(3, 16, 1, bias=False), nn.BatchNorm2d(16)).to(dev).train()   # (a bias in front of a BatchNorm has a zero gradient: noise on both sides)
    c3 = nn.Conv1d(16, 8, 3, padding=1).to(dev)
    rows = torch.randn(5003, 16, device=dev)
    x1 = rows.t().unsqueeze(0)                                   # the reference's (1, C, M) view of row-major features: strided
    x2 = torch.randn(1, 3, 777, 16, device=dev)

    def run():
        out = []
        for mod, x in ((c1, x1), (c2, x2), (c3, x1)):
            for m in mod.modules():
                if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                    m.reset_running_stats()
            xi = x.detach().clone().requires_grad_(True)
            mod.zero_grad(set_to_none=True)
            y = mod(xi)
            # a fixed random cotangent ((y * y).sum() behind a BatchNorm is all cancellation: its input gradient is noise)
            gen = torch.Generator(device=dev).manual_seed(7)
            y.backward(torch.randn(y.shape, device=dev, generator=gen))
            out.append([y.detach().clone(), xi.grad.clone()] + [p.grad.clone() for p in mod.parameters()]
                       + [b.clone() for b in mod.buffers() if b.dtype.is_floating_point])
        return out
    # the yardstick: torch's native kernels for every layer.  (The vendor's training BatchNorm with running statistics is NOT a
    # yardstick on these shapes: on the (1, 32, 5003) tensor here it deviates from an fp64 evaluation by 3e-3, torch's native
    # kernel by 1e-6 -- tools/pointwise_bn_probe.py, round 5.)
    with torch.backends.cudnn.flags(enabled=False):
        want = run()
        dropin.pointwise_as_gemm()
        try:
            got = run()
        finally:
            dropin.pointwise_as_gemm(False)
    assert set(dropin.pointwise_as_gemm()) == {nn.Conv1d, nn.Conv2d, nn.BatchNorm1d, nn.BatchNorm2d}
    try:
        # a length the library has never seen costs nothing to prepare (the vendor path: ~0.3 s per new problem size)
        xs = [torch.randn(1, 16, 4001 + 13 * i, device=dev) for i in range(5)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for x in xs:
            c1(x)                    # convolution AND training BatchNorm: nothing is prepared per length
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 0.25
    finally:
        dropin.pointwise_as_gemm(False)
    assert nn.Conv1d.forward is conv_f
    for a, b in zip(want, got):
        for u, v in zip(a, b):
            assert float((u - v).abs().max()) <= 2e-5 * float(u.abs().max()) + 1e-6
