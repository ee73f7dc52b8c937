"""Stand-alone probe (torch only): the row-major PointNet extractor at BASELINE configs[3]'s size -- (4096 x 512) point rows
through three linear layers to 512 channels (the last activation is exactly 4 GiB), max over the 512 points of an object,
backward -- eager against the same launches recorded into a HIP graph and replayed.  Prints, per replay, how far the
weight gradients are from the eager ones.  Switches (environment): B (objects), WIDE (last width), RED (amax | max | mean),
LIN (linear | bmm)."""
import os
import torch

dev = torch.device("cuda", 0)
B, P = int(os.environ.get("B", 4096)), 512
WIDE, RED, LIN = int(os.environ.get("WIDE", 512)), os.environ.get("RED", "amax"), os.environ.get("LIN", "linear")
torch.manual_seed(0)
x = torch.randn(B * P, 4, device=dev)
ws = [torch.nn.Parameter(torch.randn(o, i, device=dev) / i ** 0.5) for i, o in ((4, 64), (64, 128), (128, WIDE))]


def lin(h, w):
    if LIN == "bmm":
        s = 128
        return torch.bmm(h.view(s, h.shape[0] // s, -1), w.t().unsqueeze(0).expand(s, -1, -1)).view(h.shape[0], -1)
    return torch.nn.functional.linear(h, w)


def step():
    for w in ws:
        w.grad = None
    h = torch.relu(lin(x, ws[0]))
    h = torch.relu(lin(h, ws[1]))
    h = lin(h, ws[2]).view(B, P, -1)
    y = h.amax(dim=1) if RED == "amax" else (h.max(dim=1)[0] if RED == "max" else h.mean(dim=1))
    y.square().mean().backward()
    return [w.grad for w in ws]


side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for _ in range(2):
        want = [g.clone() for g in step()]
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    grads = step()
torch.cuda.synchronize()
for rep in range(6):
    for t in grads:
        t.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    err = [float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(grads, want)]
    print("B=%d WIDE=%d RED=%s LIN=%s replay %d: max relative difference of the three weight gradients from eager:" % (B, WIDE, RED, LIN, rep),
          ["%.2e" % e for e in err], flush=True)
