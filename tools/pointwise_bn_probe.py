import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch import nn
import torch.nn.functional as F
dev = torch.device("cuda", 0)
torch.manual_seed(0)
conv = nn.Conv1d(16, 32, 1, bias=False).to(dev)
bn = nn.BatchNorm1d(32).to(dev).train()
rows = torch.randn(5003, 16, device=dev)
x = rows.t().unsqueeze(0)
yv = conv(x)
w = conv.weight.reshape(32, 16)
ym = torch.matmul(w, x)
print("conv out: vendor strides", yv.stride(), "matmul strides", ym.stride(), "max diff", float((yv - ym).abs().max()))
ref = F.batch_norm(yv.double(), None, None, None, None, True, 0.1, 1e-5)
for name, y in (("vendor-conv out", yv), ("matmul out", ym), ("matmul out .contiguous()", ym.contiguous())):
    with torch.backends.cudnn.flags(enabled=True):
        a = F.batch_norm(y, None, None, None, None, True, 0.1, 1e-5)
    with torch.backends.cudnn.flags(enabled=False):
        b = F.batch_norm(y, None, None, None, None, True, 0.1, 1e-5)
    print("%-26s BN vendor err %.2e   BN native err %.2e" % (name, float((a.double() - ref).abs().max()), float((b.double() - ref).abs().max())))
bn.reset_running_stats(); o1 = bn(yv)
bn.reset_running_stats()
with torch.backends.cudnn.flags(enabled=False):
    o2 = bn(ym)
print("module: ", float((o1.double() - ref).abs().max()), float((o2.double() - ref).abs().max()))
