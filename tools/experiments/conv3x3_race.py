import sys, os
sys.path.insert(0, "/root/repo")
import torch
from glenet_amd import conv2d as c2, _lib
dev = torch.device("cuda", 0)
torch.manual_seed(0)
lib = _lib.load()
side = torch.cuda.Stream()
big = torch.randn(8192, 8192, device=dev)
for (b, cin, cout, h, w) in ((2, 64, 64, 200, 176), (2, 128, 128, 100, 88), (2, 64, 64, 40, 48), (4, 64, 64, 200, 176), (4, 128, 128, 100, 88), (2, 64, 128, 33, 40)):
    x = torch.randn(b, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / 24
    fwd, bwd = c2.packs(wt)
    bn = torch.nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).to(dev).train()
    for form in (2, 1):
        lib.glx_conv3x3_set_grid(0, form << 8)
        ref = c2._run(x, fwd, cout).clone()
        ref_s = [t.clone() for t in c2._run(x, fwd, cout, bn)]
        bad = bad_s = 0
        for it in range(int(os.environ.get('ITERS', '60'))):
            if it % 2 == 0:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        big @ big
            y = c2._run(x, fwd, cout)
            bad += int(not torch.equal(y, ref))
            ys = c2._run(x, fwd, cout, bn)
            bad_s += int(not all(torch.equal(a_, b_) for a_, b_ in zip(ys, ref_s)))
        torch.cuda.synchronize()
        print((b, cin, cout, h, w), "form", form, "runs that differ from the first:", bad, "with stats:", bad_s, flush=True)
