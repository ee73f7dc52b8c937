import sys
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from glenet_amd import dense_path as dp
dev = torch.device("cuda", 0)
torch.manual_seed(0)
torch.backends.cudnn.benchmark = False
x = torch.randn(2, 64, 200, 176, device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(128, 64, 3, 3, device=dev) / 24).contiguous(memory_format=torch.channels_last)
ref = F.conv2d(x, w, None, 2, 1)
print("MIOpen s2 conv forward differing runs:", sum(int(not torch.equal(F.conv2d(x, w, None, 2, 1), ref)) for _ in range(30)))
bn = torch.nn.BatchNorm2d(128, eps=1e-3, momentum=0.01).to(dev).train()
y0 = dp.BEVBackbone._fused_bn_relu(bn, ref, True)
print("fused BN + ReLU forward differing runs:", sum(int(not torch.equal(dp.BEVBackbone._fused_bn_relu(bn, ref, True), y0)) for _ in range(30)))
# the deconvs and the sparse first layer
from glenet_amd import conv2d as c2
wt = torch.randn(128, 128, 2, 2, device=dev) / 11
xd = torch.randn(2, 128, 100, 88, device=dev).contiguous(memory_format=torch.channels_last)
d0 = c2.deconv(xd, wt)
print("own deconv forward differing runs:", sum(int(not torch.equal(c2.deconv(xd, wt), d0)) for _ in range(30)))
from glenet_amd.spconv import core as sp
B, D, H, W, C = 2, 2, 200, 176, 128
act = (torch.rand(B, 1, H, W, device=dev) < 0.15) & (torch.rand(B, D, H, W, device=dev) < 0.6)
idx = act.nonzero().int().contiguous()
feats = torch.randn(idx.shape[0], C, device=dev)
m = dp.BEVBackbone(256).to(dev).to(memory_format=torch.channels_last).train()
def first():
    st = sp.SparseConvTensor(feats, idx, [D, H, W], B)
    st._ensure_index()
    with torch.enable_grad():
        return m._first_layer_sparse(st).detach()
f0 = first()
print("sparse first layer forward differing runs:", sum(int(not torch.equal(first(), f0)) for _ in range(20)))
def whole():
    st = sp.SparseConvTensor(feats, idx, [D, H, W], B)
    st._ensure_index()
    return m({"encoded_spconv_tensor": st, "spatial_features": None})["spatial_features_2d"].detach()
w0 = whole()
print("whole BEV backbone forward (training mode) differing runs:", sum(int(not torch.equal(whole(), w0)) for _ in range(20)))
