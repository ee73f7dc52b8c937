"""Minimal repro, torch only (no glenet_amd): a reduction that torch splits over several blocks per output ("global
reduce": staging buffer + semaphores zeroed by a cudaMemsetAsync in front of the kernel, aten/src/ATen/native/cuda/Reduce.cuh)
recorded into a HIP graph returns WRONG values from the second replay on; the same call eagerly, and reductions that need
one block per output, are right.  Found through the CVAE training step (NaN gradients from the third
replay): `h.view(B, 512, 8).amax(dim=1)` of its narrow PointNet extractor.
    python tools/graph_reduce_repro.py            # ROCm 7.2.0, torch 2.10.0+rocm7.0, MI355X: see profiles/r04_graph_reduce_repro.txt"""
import torch

dev = torch.device("cuda", 0)
print(torch.__version__, torch.version.hip, torch.cuda.get_device_name(0))
torch.manual_seed(0)
for shape, dim, op in (((256, 512, 8), 1, "amax"), ((256, 512, 8), 1, "sum"), ((256, 512, 8), 1, "max"), ((256, 512, 512), 1, "amax"),
                       ((64, 4096), 1, "amax"), ((8, 1 << 20), 1, "sum"), ((1 << 22,), 0, "sum")):
    x = torch.randn(shape, device=dev)
    f = {"amax": lambda t: t.amax(dim=dim), "sum": lambda t: t.sum(dim=dim), "max": lambda t: t.max(dim=dim)[0]}[op]
    want = f(x)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        f(x)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        y = f(x)
    errs = []
    for rep in range(4):
        y.fill_(float("nan"))                      # whatever the replay does not write stays NaN
        g.replay()
        torch.cuda.synchronize()
        left = int(torch.isnan(y).sum())
        ok = ~torch.isnan(y)
        errs.append("%d of %d outputs unwritten, max error of the written ones %.3g" % (
            left, y.numel(), float((y[ok] - want[ok]).abs().max()) if bool(ok.any()) else 0.0))
    print("%-5s over dim %d of %-16s" % (op, dim, tuple(shape)), flush=True)
    for rep, e in enumerate(errs):
        print("      replay %d: %s" % (rep, e), flush=True)
