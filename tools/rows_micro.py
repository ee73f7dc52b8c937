"""The pooling MLPs' Conv(k = 1) + BatchNorm (+ ReLU) alone (csrc/glx_rows.hip) against the library formulation it replaces
(_linear_rows + the fused BatchNorm kernels), forward and forward + backward, on the training step's shapes.  Event-timed."""
import copy
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd.pcdet_ops.pointnet2.pointnet2_stack import voxel_pool_modules as vpm  # noqa: E402
from glenet_amd.spconv import core  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows, cin, cout, relu in ((61952, 32, 32, False), (41984, 64, 32, False), (26112, 64, 32, False), (110592, 32, 32, True)):
    torch.manual_seed(0)
    conv, bn = nn.Conv1d(cin, cout, 1, bias=False), nn.BatchNorm1d(cout)
    seq = (nn.Sequential(conv, bn, nn.ReLU()) if relu else nn.Sequential(conv, bn)).to(dev).train()
    old = copy.deepcopy(seq)
    x = torch.randn(rows, cin, device=dev, requires_grad=True)
    cot = torch.randn(rows, cout, device=dev)

    def new_f():
        return vpm.rows_conv_bn(seq, x)

    def old_f():
        w = old[0].weight.reshape(cout, cin)
        return core.fused_train_bn(old[1], vpm.NeighborVoxelSAModuleMSG._linear_rows(x, w, None), relu, None)

    def fb(f):
        def run():
            y = f()
            torch.autograd.grad(y, (x, (seq if f is new_f else old)[0].weight, (seq if f is new_f else old)[1].weight), cot)
        return run

    with torch.no_grad():
        pass
    print("%6d rows %2d -> %2d relu=%d   forward %6.1f us (library form %6.1f)   forward + backward %6.1f us (library form %6.1f)"
          % (rows, cin, cout, relu, timed(new_f), timed(old_f), timed(fb(new_f)), timed(fb(old_f))), flush=True)
