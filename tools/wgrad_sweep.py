"""Sparse weight gradient (glx_sconv_wgrad: kernel + slab reduction) on the real layer shapes of the KITTI batch (GPU box):
microseconds per call (torch events around 20 back-to-back calls) -- PAIRS=1 (default) glx_sconv_wgrad_pairs over
per-offset pair lists (and what building the lists costs), PAIRS=0 glx_sconv_wgrad's (slice, offset) blocks -- and the
largest difference from an fp64 contraction over the rule pairs of three offsets."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import _lib, backbone as gb, synth  # noqa: E402
from glenet_amd.spconv import core as sp  # noqa: E402

K = synth.KITTI
dev = torch.device("cuda", 0)
frames = [synth.kitti_frame(i)[0] for i in range(4)]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
bidx = torch.from_numpy(np.concatenate([np.full(len(f), i, np.int32) for i, f in enumerate(frames)])).to(dev)
torch.manual_seed(0)
grid = gb.gv.grid_size_of(K["point_cloud_range"], K["voxel_size"])
model = gb.VoxelBackBone8x(4, grid).to(dev).eval()
calls = []
orig = sp._sconv


def spy(features, weight_kio, bias, nbr, tile_order, n_out, **kw):
    calls.append((features, weight_kio, nbr, n_out, kw.get("rules")))
    return orig(features, weight_kio, bias, nbr, tile_order, n_out, **kw)


sp._sconv = spy
with torch.no_grad():
    bd = gb.voxelize_batch(pts, bidx, 4, K)
    bd = gb.MeanVFE()(bd)
    model(bd)
sp._sconv = orig
print("PAIRS=%s" % os.environ.get("PAIRS", "1"))
print("layer (cin, cout, N_out, K, pairs, subm)".ljust(44) + "us/call   TFLOP/s   max rel err vs fp64 (one offset)")
seen, total = set(), 0.0
for f, w, nbr, n_out, rules in calls:
    Kk, cin, cout = w.shape
    R = rules.pair_count
    key = (cin, cout, n_out, Kk, R)
    if key in seen:
        continue
    seen.add(key)
    gout = torch.randn((n_out, cout), device=dev)
    dW = torch.empty_like(w)
    wsb = _lib.query("glx_sconv_wgrad_workspace_bytes", n_out, Kk, cin, cout)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    subm = 1 if rules.subm else 0
    PAIRS = os.environ.get("PAIRS", "1") != "0"
    if PAIRS:
        plb = _lib.query("glx_pair_lists_bytes", n_out, Kk)
        pl = torch.empty(plb, dtype=torch.uint8, device=dev)
        wsb = _lib.query("glx_sconv_wgrad_pairs_workspace_bytes", n_out, Kk, cin, cout)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

        def build():
            _lib.call("glx_pair_lists_build", nbr, n_out, Kk, None, pl, _lib.size_arg(plb))
        build()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for it in range(20):
            build()
        t1.record()
        torch.cuda.synchronize()
        build_us = t0.elapsed_time(t1) * 1e3 / 20
        meta = pl[:4 * 57].view(torch.int32).cpu().numpy()
        assert meta[Kk] == R, (meta[:Kk + 1], R)

    def run():
        if PAIRS:
            _lib.call("glx_sconv_wgrad_pairs", f, gout, pl, n_out, Kk, cin, cout, dW, ws, _lib.size_arg(wsb))
        else:
            _lib.call("glx_sconv_wgrad", f, f.shape[0], gout, nbr, n_out, Kk, cin, cout, dW, None, subm, ws, _lib.size_arg(wsb))
    run()
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for it in range(20):
            run()
        t1.record()
        torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) * 1e3 / 20)
    us = float(np.median(ts[1:]))
    total += us
    errs = []
    for k in sorted({0, Kk // 2, Kk - 1}):
        col = nbr[:n_out, k].long()
        m = col >= 0
        want = f[col[m]].double().t() @ gout[m].double()
        errs.append(float((dW[k].double() - want).abs().max() / (want.abs().max() + 1e-30)))
    print(("(%d, %d, %d, %d, %d, %d)" % (cin, cout, n_out, Kk, R, subm)).ljust(44)
          + "%7.1f   %7.2f   %.2e" % (us, 2.0 * R * cin * cout / us / 1e6, max(errs))
          + ("   lists %.1f us, CH %d, chunks %d" % (build_us, meta[56], meta[28 + Kk]) if PAIRS else ""), flush=True)
print("sum over the distinct layers: %.1f us" % total)
