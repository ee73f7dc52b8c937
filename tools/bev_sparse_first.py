"""First BEV layer (ZeroPad2d + Conv2d(C*D -> 64, 3)) on the SPARSE tensor instead of its dense() image: the 3x3 conv
over (y, x) with the depth folded into channels is a sparse conv with kernel (D, 3, 3), stride (D, 1, 1), padding
(0, 1, 1).  Prototype: values against the dense form, time of forward + backward of both."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from glenet_amd import conv2d as c2
from glenet_amd.spconv import core as sp

dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, D, H, W, C, CO = 4, 2, 200, 176, 128, 64
occ = torch.rand(B, H, W, device=dev) < 0.156
zz = torch.rand(B, D, H, W, device=dev) < 0.6
act = (occ[:, None] & zz)
idx = act.nonzero().int().contiguous()             # (N, 4) = b, z, y, x
N = idx.shape[0]
feats = torch.randn(N, C, device=dev, requires_grad=True)
wt = torch.nn.Parameter((torch.randn(CO, C * D, 3, 3, device=dev) / 48).contiguous(memory_format=torch.channels_last))
print("active voxels", N, "active cells", int(occ.sum()))


def dense_form():
    st = sp.SparseConvTensor(feats, idx, [D, H, W], B)
    st._ensure_index()
    x = st.dense_bev()
    return c2.conv3x3(x, wt)


def sparse_form():
    st = sp.SparseConvTensor(feats, idx, [D, H, W], B)
    w = wt.permute(2, 3, 1, 0).unflatten(2, (C, D)).permute(3, 0, 1, 2, 4).reshape(D * 9, C, CO)
    rs = sp.build_strided_rules(st, (D, 3, 3), (D, 1, 1), (0, 1, 1))
    f = sp.SparseConvFunction.apply(st.features, w, None, rs, False, None, False)
    out = sp.SparseConvTensor(f, rs.out_indices, rs.out_spatial_shape, B)
    out._index = rs.out_index
    return out.dense_bev()


yd, ys = dense_form(), sparse_form()
with torch.no_grad():
    st_ = sp.SparseConvTensor(feats.detach(), idx, [D, H, W], B)
    xr = st_.dense().reshape(B, C * D, H, W)
    yl = F.conv2d(xr, wt.detach().contiguous(), None, 1, 1)
    print("dense vs lib", float((yd - yl).abs().max()), "sparse vs lib", float((ys - yl).abs().max()))
    bad = ((ys - yl).abs() > 1e-3)
    print("bad fraction", float(bad.float().mean()), "bad pixels", int(bad.any(1).sum()), "of", B * H * W,
          "active out", int((yl.abs().sum(1) > 0).sum()))
    bp = bad.any(1).nonzero()[:5]
    print(bp.tolist())
print("shapes", tuple(yd.shape), tuple(ys.shape), "max diff", float((yd - ys).abs().max()), "scale", float(yd.abs().max()))
g = torch.randn_like(yd)
res = []
for fn in (dense_form, sparse_form):
    feats.grad = None; wt.grad = None
    fn().backward(g)
    res.append((feats.grad.clone(), wt.grad.clone()))
print("grad diffs", float((res[0][0] - res[1][0]).abs().max()), float(res[0][0].abs().max()),
      float((res[0][1] - res[1][1]).abs().max()), float(res[0][1].abs().max()))


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def fb(fn):
    def run():
        feats.grad = None; wt.grad = None
        fn().backward(g)
    return run


with torch.no_grad():
    print("forward us: dense %.1f sparse %.1f" % (t(dense_form), t(sparse_form)))
print("fwd+bwd us: dense %.1f sparse %.1f" % (t(fb(dense_form)), t(fb(sparse_form))))
