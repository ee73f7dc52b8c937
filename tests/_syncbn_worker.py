"""Worker of tests/test_dist_gpu.py: the reference's `--sync_bn` option (tools/train.py:119-120:
`torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)`) on a sparse stack built from this package's spconv classes.
Every rank holds one frame; the converted stack's output rows and the summed weight gradients must equal ONE process running
the unconverted stack (fused conv + BatchNorm kernels) on the union batch -- BatchNorm statistics over all ranks' rows.
World size 1 (one GPU): the conversion and the module path alone."""
import json
import os
import sys

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from glenet_amd import dist as gdist
    from glenet_amd import spconv
    from glenet_amd.spconv import core as sp
    rank, local_rank, world = gdist.env_world()
    di = local_rank % torch.cuda.device_count()
    dev = torch.device("cuda", di)
    torch.cuda.set_device(di)
    gdist.init("nccl", device=dev, force=True)
    shape = [9, 40, 36]

    def frame(seed, b):
        rng = np.random.default_rng(100 + seed)
        occ = rng.random(tuple(shape)) < 0.08
        idx = np.argwhere(occ).astype(np.int32)
        idx = np.concatenate([np.full((len(idx), 1), b, np.int32), idx], 1)
        return idx, rng.normal(size=(len(idx), 4)).astype(np.float32)

    def build():
        torch.manual_seed(0)
        m = spconv.SparseSequential(
            spconv.SubMConv3d(4, 16, 3, padding=1, bias=False, indice_key="a"), nn.BatchNorm1d(16, eps=1e-3, momentum=0.01), nn.ReLU(),
            spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False, indice_key="b"), nn.BatchNorm1d(32, eps=1e-3, momentum=0.01),
            nn.ReLU()).to(dev).train()
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, nn.BatchNorm1d):
                    mod.weight.uniform_(0.5, 1.5)
                    mod.bias.normal_(0, 0.2)
        return m

    def run(model, frames):
        idx = np.concatenate([f[0] for f in frames])
        feats = np.concatenate([f[1] for f in frames])
        x = sp.SparseConvTensor(torch.from_numpy(feats).to(dev), torch.from_numpy(idx).to(dev), shape, len(frames))
        y = model(x)
        gen = torch.Generator(device=dev).manual_seed(5)
        cot = torch.randn((200000, 32), device=dev, generator=gen)
        return y, cot

    # ---- this rank: its own frame through the CONVERTED stack (batch index 0 locally)
    sync = nn.SyncBatchNorm.convert_sync_batchnorm(build())
    assert sum(isinstance(m, nn.SyncBatchNorm) for m in sync.modules()) == 2
    mine = frame(rank, 0)
    y, cot = run(sync, [mine])
    # the union's row order is frame 0's rows, then frame 1's: this rank's cotangent rows start behind the lower ranks' rows
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[rank] = y.features.shape[0]
    dist.all_reduce(counts)
    off = int(counts[:rank].sum())
    (y.features * cot[off:off + y.features.shape[0]]).sum().backward()
    grads = [p.grad.detach().clone() for p in sync.parameters()]
    for g in grads:
        dist.all_reduce(g)                       # loss of the union = sum of the ranks' losses
    # ---- one process, unconverted stack (fused kernels), union batch
    ref = build()
    frames = [frame(r, r) for r in range(world)]
    yr, _ = run(ref, frames)
    (yr.features * cot[:yr.features.shape[0]]).sum().backward()
    n = y.features.shape[0]
    rows = yr.features[off:off + n]
    out = dict(rank=rank, world=world, rows=n,
               same_indices=bool(torch.equal(y.indices[:, 1:], yr.indices[off:off + n, 1:])),
               out_err=float((y.features - rows).abs().max()), out_scale=float(rows.abs().max()),
               grad_err=[float((a - p.grad).abs().max()) / (float(p.grad.abs().max()) + 1e-12) for a, p in zip(grads, ref.parameters())],
               stats_err=max(float((a - b).abs().max()) for a, b in zip([m.running_var for m in sync.modules() if isinstance(m, nn.SyncBatchNorm)],
                                                                          [m.running_var for m in ref.modules() if isinstance(m, nn.BatchNorm1d)])))
    print("SYNCBN " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
