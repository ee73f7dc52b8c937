"""Stand-alone probe (torch only, no glenet_amd): do memset / memcpy NODES of a captured HIP graph survive sizes at and
beyond 4 GiB?  The CVAE training step's widest activation (4096 objects x 512 points x 512 channels x 4 B) is EXACTLY
2^32 bytes; its row-major form replayed as a graph returns NaN gradients while eager launches of the same ops are fine
(tools/cvae_nan_probe2.py).  Eager and recorded fills / copies / memsets of the same buffers are compared."""
import torch

dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev)


def check(tag, nbytes, op):
    """op(dst, src) enqueues the work under test.  Eager result vs graph-replay result."""
    n = nbytes // 4
    src = torch.arange(n, dtype=torch.int32, device=dev).view(torch.float32)     # every element distinct and non-zero after 0
    dst = torch.empty(n, dtype=torch.float32, device=dev)

    def bad(ref):
        return int((dst.view(torch.int32) != ref.view(torch.int32)).sum())
    dst.fill_(7.0)
    op(dst, src)
    torch.cuda.synchronize()
    want = dst.clone()
    g = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        op(dst, src)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        op(dst, src)
    torch.cuda.synchronize()
    out = []
    for rep in range(3):
        dst.fill_(7.0)
        g.replay()
        torch.cuda.synchronize()
        out.append(bad(want))
    print("%-34s %13d bytes: elements differing from the eager result per replay %s" % (tag, nbytes, out), flush=True)
    del g, src, dst, want
    torch.cuda.empty_cache()


for nbytes in (2 ** 32 - 4096, 2 ** 32, 2 ** 32 + 4096, 2 ** 33):
    check("copy_ (memcpy node)", nbytes, lambda d, s: d.copy_(s))
    check("zero_", nbytes, lambda d, s: d.zero_())
    check("hipMemsetAsync via torch.cuda.memset", nbytes, lambda d, s: torch.cuda.cudart().cudaMemsetAsync(d.data_ptr(), 0, d.numel() * 4, torch.cuda.current_stream().cuda_stream)
          if hasattr(torch.cuda.cudart(), "cudaMemsetAsync") else d.zero_())
    check("mul_ (elementwise kernel)", nbytes, lambda d, s: torch.mul(s, 2.0, out=d))
    check("amax over rows (reduce)", nbytes, lambda d, s: d[:512].copy_(s.view(-1, 512).amax(dim=0)))
