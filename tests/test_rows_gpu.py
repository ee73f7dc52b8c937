"""csrc/glx_rows.hip: Conv(k = 1, bias = False) + training-mode BatchNorm (+ ReLU) on (rows, C) matrices -- the input / output
MLPs of the RoI-grid pool (pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py:70-130) -- against an fp64 evaluation of the
reference's modules (nn.Conv1d + nn.BatchNorm1d + nn.ReLU on the (1, C, M) layout), forward, all four gradients and the running
statistics; the plain product through the C ABI; live-row counts of shape-static matrices."""
import ctypes

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _modules(cin, cout, relu, seed):
    torch.manual_seed(seed)
    conv = nn.Conv1d(cin, cout, 1, bias=False)
    bn = nn.BatchNorm1d(cout)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.2, 0.2)
        bn.running_var.uniform_(0.5, 1.5)
    return nn.Sequential(conv, bn, nn.ReLU()) if relu else nn.Sequential(conv, bn)


def _reference(seq, x, cot, live):
    """fp64, the reference's layout: (1, C, M) through Conv1d / BatchNorm1d / ReLU on the live rows."""
    import copy
    ref = copy.deepcopy(seq).double().train()
    xr = x[:live].double().detach().clone().requires_grad_(True)
    y = ref(xr.t().unsqueeze(0)).squeeze(0).t()
    (y * cot[:live].double()).sum().backward()
    return ref, y.detach(), xr.grad


@pytest.mark.parametrize("rows,cin,cout,relu,live", [
    (20011, 32, 32, True, None),
    (110592, 32, 32, True, None),
    (61858, 64, 32, False, 50021),
    (30000, 64, 32, False, None),
    (4099, 16, 64, True, 4001),
    (9000, 64, 64, True, None),
    (2048, 32, 16, False, 37),
    (50000, 64, 128, True, None),          # the CVAE's 64 -> 128 point layer (forward f16 x 2 or fp32 column halves)
    (33333, 64, 128, False, 30011),
    (7001, 16, 128, False, 6500),
])
@pytest.mark.parametrize("f16x2_wide", [True, False])
def test_rows_conv_bn_matches_the_reference_modules_in_fp64(dev, rows, cin, cout, relu, live, f16x2_wide, monkeypatch):
    from glenet_amd.pcdet_ops.pointnet2.pointnet2_stack import voxel_pool_modules as vpm
    if not f16x2_wide and (cin, cout) != (64, 128):
        pytest.skip("the arithmetic choice exists for 64 -> 128 only")
    monkeypatch.setattr(vpm, "ROWS_64_128_F16X2", f16x2_wide)
    seq = _modules(cin, cout, relu, rows).to(dev).train()
    g = torch.Generator(device=dev).manual_seed(rows + cin)
    x = torch.randn(rows, cin, device=dev, generator=g) * 1.5 + 0.3
    cot = torch.randn(rows, cout, device=dev, generator=g)
    count = None
    n = rows
    if live is not None:
        n = live
        x[n:] = 0                                             # the contract: rows past the count are zero
        count = torch.tensor([n], dtype=torch.int32, device=dev)
    ref, y_ref, gx_ref = _reference(seq, x, cot, n)
    x.requires_grad_(True)
    assert vpm.rows_conv_bn_supported(seq, x)
    y = vpm.rows_conv_bn(seq, x, count)
    (y * cot).sum().backward()
    torch.cuda.synchronize()
    ys = float(y_ref.abs().max())
    assert float((y[:n].double() - y_ref).abs().max()) <= 2e-6 * ys + 1e-6
    if live is not None:
        assert float(y[n:].abs().max()) == 0.0 and float(x.grad[n:].abs().max()) == 0.0
    # a ReLU that flips on a rounding-level pre-activation moves one element's gradient: mean error, and the maximum loosely
    for got, want in ((x.grad[:n], gx_ref), (seq[0].weight.grad, ref[0].weight.grad), (seq[1].weight.grad, ref[1].weight.grad),
                      (seq[1].bias.grad, ref[1].bias.grad)):
        sc = float(want.abs().max()) + 1e-30
        err = (got.double() - want).abs()
        assert float(err.mean()) <= 2e-6 * sc, (float(err.mean()), sc)
        assert float(err.max()) <= (2e-3 if relu else 2e-5) * sc, (float(err.max()), sc)
    assert float((seq[1].running_mean.double() - ref[1].running_mean).abs().max()) <= 1e-6
    assert float((seq[1].running_var.double() - ref[1].running_var).abs().max()) <= 1e-5 * float(ref[1].running_var.max())
    assert int(seq[1].num_batches_tracked) == 1


def test_rows_conv_bn_equals_the_library_formulation_it_replaces(dev):
    """The same Sequential through _linear_rows + the fused BatchNorm kernels (GLX_ROWS_CONV_BN=0's path): results agree to fp32
    rounding, including a live-row count."""
    from glenet_amd.pcdet_ops.pointnet2.pointnet2_stack import voxel_pool_modules as vpm
    from glenet_amd.spconv import core
    import copy
    rows, cin, cout, n = 40960, 64, 32, 33333
    seq = _modules(cin, cout, True, 5).to(dev).train()
    old = copy.deepcopy(seq)
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(rows, cin, device=dev, generator=g)
    x[n:] = 0
    cot = torch.randn(rows, cout, device=dev, generator=g)
    count = torch.tensor([n], dtype=torch.int32, device=dev)
    xa = x.clone().requires_grad_(True)
    ya = vpm.rows_conv_bn(seq, xa, count)
    (ya * cot).sum().backward()
    xb = x.clone().requires_grad_(True)
    w = old[0].weight.reshape(cout, cin)
    yb = core.fused_train_bn(old[1], vpm.NeighborVoxelSAModuleMSG._linear_rows(xb, w, None), True, count)
    (yb * cot).sum().backward()
    torch.cuda.synchronize()
    assert float((ya - yb).abs().max()) <= 1e-5 * float(yb.abs().max())
    for a, b in ((xa.grad, xb.grad), (seq[0].weight.grad, old[0].weight.grad), (seq[1].weight.grad, old[1].weight.grad),
                 (seq[1].bias.grad, old[1].bias.grad)):
        assert float((a - b).abs().mean()) <= 1e-5 * float(b.abs().max())
    assert torch.allclose(seq[1].running_var, old[1].running_var, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("rows,cin,cout", [(16, 32, 32), (33, 16, 16), (70001, 64, 64), (12345, 32, 64), (40003, 64, 128)])
def test_plain_product_through_the_c_abi(dev, rows, cin, cout):
    """bn_state == NULL / coef3 == NULL: z = x w^T, gx = dy w, gw = dy^T x (fp64 products as the yardstick); twice: the weight
    gradient's partial sums are added in a fixed order (bitwise equal)."""
    from glenet_amd import _lib
    g = torch.Generator(device=dev).manual_seed(rows)
    x = torch.randn(rows, cin, device=dev, generator=g)
    w = torch.randn(cout, cin, device=dev, generator=g) / cin ** 0.5
    dy = torch.randn(rows, cout, device=dev, generator=g)
    z = torch.empty(rows, cout, device=dev)
    zero = ctypes.c_float(0.0)
    _lib.call("glx_rows_linear_bn_forward", x, rows, cin, w, cout, None, z, None, None, zero, zero, None, None, None, None, None, None)
    ws = torch.empty(_lib.query("glx_rows_linear_workspace_bytes", cin, cout), dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(2):
        gx, gw = torch.empty_like(x), torch.empty_like(w)
        _lib.call("glx_rows_linear_bn_backward", x, None, dy, rows, cin, w, cout, None, None, 0, None, None, None, gx, gw, ws,
                  _lib.size_arg(ws.numel()))
        outs.append((gx, gw))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    xd, wd, dd = x.double(), w.double(), dy.double()
    for got, want in ((z, xd @ wd.t()), (outs[0][0], dd @ wd), (outs[0][1], dd.t() @ xd)):
        assert float((got.double() - want).abs().max()) <= 3e-6 * float(want.abs().max())


def test_unsupported_widths_are_refused_loudly(dev):
    from glenet_amd import _lib
    x = torch.zeros(64, 48, device=dev)
    w = torch.zeros(32, 48, device=dev)
    z = torch.zeros(64, 32, device=dev)
    zero = ctypes.c_float(0.0)
    assert _lib.query("glx_rows_linear_supported", 48, 32) == 0 and _lib.query("glx_rows_linear_supported", 64, 32) == 1
    with pytest.raises(_lib.GlxError, match="channels 48 -> 32"):
        _lib.call("glx_rows_linear_bn_forward", x, 64, 48, w, 32, None, z, None, None, zero, zero, None, None, None, None, None, None)


@pytest.mark.parametrize("njobs,rows,cin,cout", [(5, 512, 256, 256), (3, 510, 256, 256), (8, 7, 64, 64), (1, 1021, 192, 128)])
def test_grouped_weight_gradients_of_small_linear_layers(dev, njobs, rows, cin, cout):
    """glx_linear_wgrad_multi: out_j = gy_j^T x_j for several layers of one shape in one launch (the RoI towers' 256 x 256 filters,
    voxelrcnn_head.py:40-66) against fp64 products; a second call gives the same bits."""
    from glenet_amd import _lib
    g = torch.Generator(device=dev).manual_seed(rows * 7 + njobs)
    xs = [torch.randn(rows, cin, device=dev, generator=g) for _ in range(njobs)]
    gys = [torch.randn(rows, cout, device=dev, generator=g) for _ in range(njobs)]
    ptrs = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    results = []
    for _ in range(2):
        outs = [torch.full((cout, cin), float("nan"), device=dev) for _ in range(njobs)]
        _lib.call("glx_linear_wgrad_multi", njobs, ptrs(xs), ptrs(gys), ptrs(outs), rows, cin, cout)
        results.append(outs)
    torch.cuda.synchronize()
    for j in range(njobs):
        assert torch.equal(results[0][j], results[1][j])
        want = gys[j].double().t() @ xs[j].double()
        assert float((results[0][j].double() - want).abs().max()) <= 3e-6 * float(want.abs().max())
    with pytest.raises(_lib.GlxError, match="cout % 64"):
        _lib.call("glx_linear_wgrad_multi", 1, ptrs(xs[:1]), ptrs(gys[:1]), ptrs(results[0][:1]), rows, cin, 48)


def test_deferred_tower_weight_gradients_run_grouped_and_match_the_library_products(dev):
    """dense_path.run_deferred_fc_wgrads: the five (256, 256) jobs of the FC towers in one launch + the long first Linear as before;
    gradients equal the per-job library products (GLX_FC_WGRADS_GROUPED=0's path), including accumulation into an existing .grad."""
    from glenet_amd import dense_path as dp
    g = torch.Generator(device=dev).manual_seed(3)
    rows = 512
    shapes = [(256, 1024)] + [(256, 256)] * 5 + [(64, 96), (128, 64)]
    def make():
        ws = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in shapes]
        ws[2].grad = torch.ones_like(ws[2])                    # one filter already has a gradient: added, not overwritten
        return ws
    xs = [torch.randn(rows, s[1], device=dev, generator=g) for s in shapes]
    gys = [torch.randn(rows, s[0], device=dev, generator=g) for s in shapes]
    res = []
    for grouped in (True, False):
        old = dp.FC_WGRADS_GROUPED
        dp.FC_WGRADS_GROUPED = grouped
        try:
            ws = make()
            dp.run_deferred_fc_wgrads([(x, gy, w, None, dp._SplitKLinearFn.weight_grad) for x, gy, w in zip(xs, gys, ws)])
        finally:
            dp.FC_WGRADS_GROUPED = old
        res.append([w.grad.clone() for w in ws])
    torch.cuda.synchronize()
    for a, b, x, gy in zip(res[0], res[1], xs, gys):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    want = gys[2].double().t() @ xs[2].double() + 1.0
    assert float((res[0][2].double() - want).abs().max()) <= 3e-6 * float(want.abs().max())


@pytest.mark.parametrize("rows,k,n", [(512, 20736, 256), (37, 4160, 64), (130, 4096, 128)])
def test_wide_first_linear_on_own_kernels_matches_fp64(dev, rows, k, n):
    """voxelrcnn_head.py:40-52's Linear(20 736, 256, bias=False) on the RoI rows: forward (split along K, fixed-order sum), input
    gradient and weight gradient on the package's fp32-MFMA kernels (glx_linear_wide_forward / _input_grad, glx_linear_wgrad_multi)
    against fp64 products -- exact fp32 products, so the error is that of fp32 accumulation; ragged row counts; the autograd
    function of SplitKLinear takes the same path; two calls give the same bits."""
    from glenet_amd import dense_path as dp
    g = torch.Generator(device=dev).manual_seed(rows + k)
    x = torch.randn(rows, k, device=dev, generator=g)
    w = torch.randn(n, k, device=dev, generator=g) / k ** 0.5
    gy = torch.randn(rows, n, device=dev, generator=g)
    assert dp._wide_linear_ok(x, w)
    y = dp.wide_linear_forward(x, w)
    gx = dp.wide_linear_input_grad(gy, w)
    gw = dp._SplitKLinearFn.weight_grad(x, gy, w)
    y64, gx64, gw64 = x.double() @ w.double().t(), gy.double() @ w.double(), gy.double().t() @ x.double()
    for got, want, mag in ((y, y64, x.double().abs() @ w.double().abs().t()), (gx, gx64, gy.double().abs() @ w.double().abs()),
                           (gw, gw64, gy.double().abs().t() @ x.double().abs())):
        assert float(((got.double() - want).abs() / mag).max()) <= 2.0 ** -18, (rows, k, n)
    assert torch.equal(y, dp.wide_linear_forward(x, w)) and torch.equal(gx, dp.wide_linear_input_grad(gy, w))
    # through autograd
    xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    lin = dp.SplitKLinear(k, n, bias=False).to(dev)
    with torch.no_grad():
        lin.weight.copy_(w)
    out = lin(xa)
    assert torch.equal(out, y)
    out.backward(gy)
    assert torch.equal(xa.grad, gx) and torch.equal(lin.weight.grad, gw)
    # the library path stays one assignment away
    dp.OWN_WIDE_LINEAR = False
    try:
        ref = dp._SplitKLinearFn.apply(x, w)
    finally:
        dp.OWN_WIDE_LINEAR = True
    assert float((ref - y).abs().max()) <= 1e-4 * float(y.abs().max())
