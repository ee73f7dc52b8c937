"""Which of the vendor paths of a 1x1 convolution is fp32-exact (GPU box): F.conv1d (MIOpen), torch.matmul (hipBLASLt / rocBLAS),
both against an fp64 product, on contiguous and on strided (1, C, M) inputs."""
import torch
import torch.nn.functional as F

dev = torch.device("cuda", 0)
torch.manual_seed(0)
for cin, cout, m in ((16, 32, 5003), (32, 32, 110592), (64, 64, 40000), (256, 256, 512)):
    rows = torch.randn(m, cin, device=dev)
    w = torch.randn(cout, cin, device=dev) / cin ** 0.5
    for name, x in (("contiguous", rows.t().contiguous().unsqueeze(0)), ("strided", rows.t().unsqueeze(0))):
        ref = torch.matmul(w.double(), x.double())
        scale = float(ref.abs().max())
        yc = F.conv1d(x, w.unsqueeze(-1))
        ym = torch.matmul(w, x)
        yr = torch.matmul(rows, w.t())          # the row-major product the own pipeline uses
        print("%3d -> %3d, M = %6d, %-10s: conv1d err %.2e   matmul err %.2e   rows @ W^T err %.2e   (of max |y| = %.2f)"
              % (cin, cout, m, name, float((yc.double() - ref).abs().max()) / scale, float((ym.double() - ref).abs().max()) / scale,
                 float((yr.double().t().unsqueeze(0) - ref).abs().max()) / scale, scale), flush=True)
print("torch.backends.cuda.matmul.allow_tf32 =", torch.backends.cuda.matmul.allow_tf32, " cudnn.allow_tf32 =", torch.backends.cudnn.allow_tf32)
print("float32 matmul precision:", torch.get_float32_matmul_precision())
