"""How much a chain of tiny kernels on a second stream slows the BEV convolution kernels (persistent grids sized to
the chip: a few occupied slots push the last blocks into a second round).
  python tools/conv_side_load.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import conv2d as c2  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(4, 64, 200, 176, device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 64, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
gy = torch.randn_like(x)
side = torch.cuda.Stream(dev)
small = torch.zeros(4096, device=dev)
big = torch.zeros(1 << 22, device=dev)


def run(fn, load):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if load is not None:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(4000):
                load()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(40):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 40 * 1e3


with torch.no_grad():
    fns = {"conv3x3 forward": lambda: c2.conv3x3(x, w), "weight gradient": lambda: c2.wgrad(x, gy, w)}
    loads = {"alone": None, "tiny kernels beside (4 K floats)": lambda: small.add_(1.0),
             "16 MB elementwise kernels beside": lambda: big.add_(1.0)}
    for fname, fn in fns.items():
        for lname, load in loads.items():
            print("%-18s %-36s %7.1f us" % (fname, lname, run(fn, load)), flush=True)

# ---- the same inside ONE captured graph: convolutions on the capture stream, a chain of small kernels on a forked stream
def graph_ms(n_conv, n_small, fork=True):
    cap = torch.cuda.Stream(dev)
    br = torch.cuda.Stream(dev)
    cap.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g, stream=cap):
        if fork:
            br.wait_stream(cap)
            with torch.cuda.stream(br):
                for _ in range(n_small):
                    small.add_(1.0)
        else:
            for _ in range(n_small):
                small.add_(1.0)
        for _ in range(n_conv):
            c2.conv3x3(x, w)
        if fork:
            cap.wait_stream(br)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 10


print("one graph: 40 convolutions                           %.3f ms" % graph_ms(40, 0))
print("one graph: 600 small kernels                         %.3f ms" % graph_ms(0, 600))
print("one graph: both, small kernels on a forked stream    %.3f ms" % graph_ms(40, 600))
print("one graph: both on one stream                        %.3f ms" % graph_ms(40, 600, fork=False))

# ---- with room left on every CU: 512 (two per CU) and 256 convolution blocks instead of the resident 768
from glenet_amd import _lib  # noqa: E402
for blocks in (512, 256):
    _lib.call_nostream("glx_conv3x3_set_grid", blocks, 0)
    print("%d conv blocks: 40 convolutions %.3f ms, with 600 small kernels on a forked stream %.3f ms"
          % (blocks, graph_ms(40, 0), graph_ms(40, 600)))
_lib.call_nostream("glx_conv3x3_set_grid", 0, 0)
