"""The block kernel's two arithmetics (fp32 MFMAs | two scaled fp16 pieces and three fp16 MFMAs, GLX_SCONV_ARITH) on the real layer
shapes of the KITTI batch (GPU box): error against an fp64 evaluation of the rule table, and time per launch from HIP events.
SCALES=1 additionally runs inputs whose rows / channels differ by many powers of two."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glenet_amd import _lib, backbone as gb, synth  # noqa: E402
from glenet_amd.spconv import core as sp  # noqa: E402


def _set_variant(v):
    """The tile-shape sweep variants (2-42, 60 / 61) were removed from the product library in round 6 (every one of them measured
    and rejected: profiles/LABBOOK_r01_r04.md, r05_sconv_bound.md); on a library without the knob only the default runs."""
    try:
        fn = _lib.load().glx_sconv_set_variant
    except AttributeError:
        if v not in (-1, None):
            raise SystemExit("this library has no glx_sconv_set_variant: check out a round <= 5 tree for the sweep variants")
        return
    fn(int(v))



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
    calls.append((features, weight_kio, nbr, tile_order, n_out, kw.get("rules")))
    return orig(features, weight_kio, bias, nbr, tile_order, n_out, **kw)


sp._sconv = spy
with torch.no_grad():
    bd = gb.voxelize_batch(pts, bidx, 4, K)
    bd = gb.MeanVFE()(bd)
    model(bd)
sp._sconv = orig


def reference(f, w, nbr, n_out):
    """fp64: out[j] = sum_k f[nbr[j, k]] @ w[k] over the present neighbours; also sum |f| |w| (the error scale)."""
    fd, wd = f.double(), w.double()
    out = torch.zeros(n_out, w.shape[2], dtype=torch.float64, device=dev)
    mag = torch.zeros_like(out)
    for k in range(w.shape[0]):
        idx = nbr[:n_out, k].long()
        ok = idx >= 0
        rows = fd[idx.clamp(min=0)] * ok[:, None]
        out += rows @ wd[k]
        mag += rows.abs() @ wd[k].abs()
    return out, mag


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        t0.record()
        fn()
        t1.record()
        torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) * 1e3)
    return float(np.median(ts))


seen = set()
print("layer (cin, cout, N, pairs)".ljust(36) + "fp32 us   f16x2 us |  max err / sum|f||w|:  fp32      f16x2   | max |f16x2 - fp32| / max |out|")
for f, w, nbr, order, n_out, rules in calls:
    Kk, cin, cout = w.shape
    key = (cin, cout, n_out, Kk)
    if key in seen or cin % 32 or cout < 64:
        continue
    seen.add(key)
    variants = [("layer", f, w)]
    if os.environ.get("SCALES"):
        g = torch.Generator(device=dev).manual_seed(cin + n_out)
        rs = torch.exp2(torch.randint(-20, 21, (f.shape[0], 1), device=dev, generator=g).float())
        cs = torch.exp2(torch.randint(-12, 13, (1, cin), device=dev, generator=g).float())
        variants += [("rows x 2^[-20, 20]", f * rs, w), ("channels x 2^[-12, 12]", f * cs, w),
                     ("filter x 2^-30", f, w * 2.0 ** -30), ("randn", torch.randn_like(f), torch.randn_like(w) / (27 * cin) ** 0.5)]
    for name, fv, wv in variants:
        fv, wv = fv.contiguous(), wv.contiguous()
        packed = sp.pack_weights(wv)
        want, mag = reference(fv, wv, nbr, n_out)
        res, tm = {}, {}
        for arith in (0, 1):
            _lib.call_nostream("glx_sconv_set_arith", arith)
            res[arith] = orig(fv, wv, None, nbr, order, n_out, packed=packed, rules=rules)
            tm[arith] = timed(lambda: orig(fv, wv, None, nbr, order, n_out, packed=packed, rules=rules))
        _lib.call_nostream("glx_sconv_set_arith", 1)
        extra = ""
        for v in [int(x) for x in os.environ.get("VARIANTS", "").split(",") if x]:      # other kernel forms of the f16 x 2 arithmetic
            _set_variant(v)
            got = orig(fv, wv, None, nbr, order, n_out, packed=packed, rules=rules)
            t = timed(lambda: orig(fv, wv, None, nbr, order, n_out, packed=packed, rules=rules))
            _set_variant(-1)
            extra += " | v%d %.1f us, max |d| vs f16x2 %.3g" % (v, t, float((got - res[1]).abs().max() / res[1].abs().max()))
        live = mag > 0
        e = [float(((res[a].double() - want).abs()[live] / mag[live]).max()) for a in (0, 1)]
        d = float((res[1] - res[0]).abs().max() / res[0].abs().max())
        print(("(%d, %d, %d, %d) %s" % (cin, cout, n_out, rules.pair_count if rules is not None else -1, name)).ljust(36)
              + "%7.1f  %7.1f   |  %.3g (2^%.1f)   %.3g (2^%.1f)  | %.3g" % (tm[0], tm[1], e[0], np.log2(e[0] + 1e-300), e[1],
                                                                         np.log2(e[1] + 1e-300), d) + extra)
