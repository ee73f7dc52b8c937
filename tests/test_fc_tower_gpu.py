"""The RoI head's FC towers on the fused kernels (csrc/glx_fctower.hip: glx_fc_tower_forward / _backward) against the
module-by-module arithmetic of voxelrcnn_kl_label_iou_head.py:38-92 restated with torch ops in float64 (Linear,
training-mode BatchNorm1d, ReLU, dropout with the SAME uniform draws): the four outputs, every parameter gradient, the
pooled features' gradient and the running statistics; one launch with grid barriers and one launch per phase."""
import copy

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


class _Head(nn.Module):
    """The attributes of glenet_vr.VoxelRCNNKLHead the towers use (same construction, a narrower first Linear)."""

    def __init__(self, pre, dp_ratio):
        super().__init__()
        from glenet_amd import dense_path as dp
        self.shared_fc_layer, c = dp._fc_tower(pre, (256, 256), dp_ratio)
        self.cls_fc_layers, c = dp._fc_tower(c, (256, 256), dp_ratio)
        self.cls_pred_layer = nn.Linear(c, 1)
        self.reg_fc_layers, c = dp._fc_tower(256, (256, 256), dp_ratio)
        self.reg_pred_layer = nn.Linear(c, 7)
        self.reg_std_layer = nn.Linear(c, 7)
        self.reg_std_bn = nn.BatchNorm1d(7)
        self.reg_std_fc1 = nn.Linear(7, 64)
        self.reg_std_bn1 = nn.BatchNorm1d(64)
        self.reg_std_fc2 = nn.Linear(64, 1)
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if p.dim() == 2:
                    p.copy_(torch.randn(p.shape, generator=g) * (1.5 / p.shape[1] ** 0.5))
                elif "bn" in name or name.split(".")[-2] in ("1", "4", "5"):       # BatchNorm weight / bias
                    p.copy_(torch.rand(p.shape, generator=g) + 0.5 if name.endswith("weight") else torch.randn(p.shape, generator=g) * 0.3)
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.1)


def _reference(head, x, u, p):
    """float64 restatement; returns the four outputs (head's buffers are updated like nn.BatchNorm1d does)."""
    def tower(seq, h, masks):
        mods = list(seq)
        i = 0
        for m in mods:
            if isinstance(m, nn.Dropout):
                h = h * (masks.pop(0) >= p).to(h.dtype) / (1.0 - p)
            else:
                h = m(h)
            i += 1
        return h
    masks = [u[0], u[1], u[2]] if u is not None else []
    shared = tower(head.shared_fc_layer, x, masks[:1])
    c = tower(head.cls_fc_layers, shared, masks[1:2])
    reg_feat = tower(head.reg_fc_layers, shared, masks[2:3])
    ori = head.cls_pred_layer(c)
    reg = head.reg_pred_layer(reg_feat)
    std = head.reg_std_layer(reg_feat)
    s = torch.relu(head.reg_std_bn(std.clone()))
    s = torch.relu(head.reg_std_bn1(head.reg_std_fc1(s)))
    return ori, head.reg_std_fc2(s), reg, std


@pytest.mark.parametrize("R,pre,p,coop", [(512, 1024, 0.3, True), (512, 1024, 0.3, False), (64, 256, 0.0, True),
                                          (48, 512, 0.0, False), (1024, 256, 0.3, True)])
def test_fc_towers_equal_the_modules(R, pre, p, coop, monkeypatch):
    from glenet_amd import dense_path as dp
    dev = torch.device("cuda", 0)
    torch.manual_seed(R + int(p * 10))
    head = _Head(pre, p).to(dev).train()
    ref = copy.deepcopy(head).double()
    x = (torch.randn(R, pre, device=dev) * 0.7).requires_grad_(True)
    xr = x.detach().double().requires_grad_(True)
    u = torch.rand(3, R, 256, device=dev) if p > 0 else None
    real_rand = torch.rand
    monkeypatch.setattr(torch, "rand", lambda *a, **k: u if (a and tuple(a[0]) == (3, R, 256)) else real_rand(*a, **k))
    monkeypatch.setattr(dp, "FC_TOWER_COOPERATIVE", coop)
    assert dp.fc_tower_usable(head, x)
    outs = dp.fc_towers(head, x)
    want = _reference(ref, xr, None if u is None else u.double(), p)
    gens = [torch.randn(o.shape, device=dev, generator=torch.Generator(dev).manual_seed(i)) for i, o in enumerate(outs)]
    torch.autograd.backward(outs, gens)
    torch.autograd.backward(want, [g.double() for g in gens])
    for name, o, w in zip(("ori_cls", "std_logit", "rcnn_reg", "rcnn_reg_std"), outs, want):
        scale = float(w.detach().abs().max()) + 1e-6
        assert float((o.double() - w).abs().max()) <= 2e-5 * scale + 1e-6, name
    worst = {}
    for (name, q), (_, qr) in zip(head.named_parameters(), ref.named_parameters()):
        assert q.grad is not None, name
        scale = float(qr.grad.abs().max()) + 1e-9
        if name == "reg_std_fc1.bias":     # in front of a training-mode BatchNorm: zero gradient, rounding noise on both sides
            scale = float(ref.reg_std_fc1.weight.grad.abs().max())
        worst[name] = float((q.grad.double() - qr.grad).abs().max()) / scale
    bad = {k: v for k, v in worst.items() if v > 2e-4}
    assert not bad, bad
    scale = float(xr.grad.abs().max())
    assert float((x.grad.double() - xr.grad).abs().max()) <= 2e-4 * scale
    for (name, b), (_, br) in zip(head.named_buffers(), ref.named_buffers()):
        if b.dtype.is_floating_point:
            assert float((b.double() - br).abs().max()) <= 1e-5 * (float(br.abs().max()) + 1e-6), name
        else:
            assert int(b) == int(br), name


def test_fc_towers_leave_the_barrier_clean_and_repeat():
    """Two calls in a row give the same bits (the barrier counters return to zero; fixed summation order)."""
    from glenet_amd import dense_path as dp
    dev = torch.device("cuda", 0)
    head = _Head(256, 0.0).to(dev).train()
    x = torch.randn(256, 256, device=dev)
    assert dp.fc_tower_usable(head, x.clone().requires_grad_(True))
    a = [o.clone() for o in dp.fc_towers(head, x.clone().requires_grad_(True))]
    b = [o.clone() for o in dp.fc_towers(head, x.clone().requires_grad_(True))]
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    bars = head.__dict__["_glx_fct_barriers"]           # owned by the module, one per (device, stream)
    assert len(bars) == 1
    for bar in bars.values():
        assert int(bar.abs().sum()) == 0
    assert not dp.fc_tower_barrier_gave_up(head)


def test_fc_tower_residency_is_checked_not_assumed(monkeypatch):
    """the one-launch towers spin at grid barriers across 32 blocks; the entry point asks the runtime whether
    those blocks are resident together (occupancy x CUs) and otherwise launches per phase.  On a whole MI355X both
    directions are supported and cooperative; a row count whose LDS request cannot fit a block is reported unsupported
    (fc_tower_usable then keeps the module path); counters are never allocated inside a stream capture."""
    from glenet_amd import dense_path as dp
    dev = torch.device("cuda", 0)
    assert dp.fc_tower_support(dev, 512) == (True, True)
    assert dp.fc_tower_support(dev, 1024)[0] is True
    assert dp.fc_tower_support(dev, 1 << 20)[0] is False          # 100 MB of LDS per block: no device has that
    head = _Head(256, 0.0).to(dev).train()
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    with pytest.raises(RuntimeError, match="warm-up"):
        dp._fct_barrier(head, dev)
