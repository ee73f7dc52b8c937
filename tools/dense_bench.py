"""Time dense(): torch zero fill, scatter path, single-pass path (GPU box)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import _lib  # noqa: E402
from glenet_amd._lib import call  # noqa: E402
from glenet_amd.spconv import core as sp  # noqa: E402

dev = torch.device("cuda")
B, C, D, H, W = 4, 128, 2, 200, 176
N = 26000
g = torch.Generator().manual_seed(0)
cells = torch.randperm(B * D * H * W, generator=g)[:N].sort().values
idx = torch.stack([cells // (D * H * W), (cells // (H * W)) % D, (cells // W) % H, cells % W], 1).int().to(dev)
feat = torch.randn(N, C, device=dev)
st = sp.SparseConvTensor(feat, idx, [D, H, W], B)
index = st._ensure_index()
empty_bitmap = torch.zeros_like(index.bitmap)
out = torch.empty((B, C, D, H, W), device=dev)


def timeit(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-44s %7.1f us" % (name, e0.elapsed_time(e1) * 1e3 / n))


timeit("torch zero fill (144 MB)", lambda: out.zero_())
timeit("scatter only", lambda: call("glx_dense_scatter", feat, idx, N, C, B, D, H, W, out, None))
timeit("single pass, real index", lambda: call("glx_dense_from_index", feat, N, C, index.bitmap, index.prefix,
                                               index.rank_to_row, B, D, H, W, out))
timeit("single pass, EMPTY index (pure stores)", lambda: call("glx_dense_from_index", feat, N, C, empty_bitmap,
                                                            index.prefix, index.rank_to_row, B, D, H, W, out))
ref = torch.zeros_like(out)
call("glx_dense_scatter", feat, idx, N, C, B, D, H, W, ref, None)
call("glx_dense_from_index", feat, N, C, index.bitmap, index.prefix, index.rank_to_row, B, D, H, W, out)
torch.cuda.synchronize()
print("equal:", torch.equal(ref, out))
