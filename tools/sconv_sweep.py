"""Sweep sparse-conv tile variants on the real layer shapes of the KITTI batch (GPU box)."""
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


def ev():
    e = ctypes.c_void_p()
    _lib.call_nostream("glx_event_create", ctypes.byref(e))
    return e


_lib.call_nostream("glx_sconv_set_xcd_group", int(os.environ.get("XCD_GROUP", "0")))
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5,6,7".split(","))]
seen = set()
print("layer(cin,cout,N,R)".ljust(34) + "".join(("v%d" % v).rjust(9) for v in variants) + "   alg GB/s @best")
for f, w, nbr, order, n_out, rules in calls:
    Kk, cin, cout = w.shape
    R = rules.pair_count
    key = (cin, cout, n_out, Kk, R)
    if key in seen or (os.environ.get("ONLY6464") and (cin, cout) != (64, 64)):
        continue
    seen.add(key)
    packed = sp.pack_weights(w)
    row = []
    _set_variant(-1)
    ref_out = orig(f, w, None, nbr, order, n_out, packed=packed)
    for v in variants:
        _set_variant(v)
        ts = []
        try:
            got = orig(f, w, None, nbr, order, n_out, packed=packed)
            if not torch.equal(got, ref_out):
                print("  !! variant %d differs from the default on %s: max |d| %.3g" % (
                    v, key, float((got - ref_out).abs().max())))
            if os.environ.get("WALL"):
                # variants made of several launches: stream time of 20 back-to-back calls (launch gaps included)
                for rep in range(5):
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record()
                    for it in range(20):
                        orig(f, w, None, nbr, order, n_out, packed=packed)
                    t1.record()
                    torch.cuda.synchronize()
                    ts.append(t0.elapsed_time(t1) * 1e3 / 20)
                row.append(float(np.median(ts[1:])))
                continue
            for it in range(12):
                s, e = ev(), ev()
                sp._profile_hook = lambda *a, s=s, e=e: (s, e)       # -> glx_sconv_opts.profile_start / _stop
                orig(f, w, None, nbr, order, n_out, packed=packed)
                sp._profile_hook = None
                ms = ctypes.c_float()
                _lib.call_nostream("glx_event_elapsed_ms", s, e, ctypes.byref(ms))
                ts.append(ms.value * 1e3)
            row.append(float(np.median(ts[2:])))
        except Exception as ex:  # variant does not fit LDS
            row.append(float("nan"))
    _set_variant(-1)
    if os.environ.get("TILE_MAP"):
        # the default kernel with the work-balanced block -> tile map of this rule table
        nt = (n_out + 63) // 64
        tmap = torch.empty(nt, dtype=torch.int32, device=dev)
        wsb = _lib.query("glx_sconv_tile_map_workspace_bytes", n_out)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _lib.call("glx_sconv_tile_map", nbr, order, n_out, Kk, None, tmap, ws, _lib.size_arg(wsb))
        assert sorted(tmap.cpu().tolist()) == list(range(nt))
        ts = []
        for it in range(12):
            s, e = ev(), ev()
            sp._profile_hook = lambda *a, s=s, e=e: (s, e)
            rules = type("R", (), dict(subm=True, _tile_maps=True, tile_map=lambda self, *a: tmap))()
            got = orig(f, w, None, nbr, order, n_out, packed=packed, rules=rules)     # -> glx_sconv_opts.tile_map
            sp._profile_hook = None
            ms = ctypes.c_float()
            _lib.call_nostream("glx_event_elapsed_ms", s, e, ctypes.byref(ms))
            ts.append(ms.value * 1e3)
        assert torch.equal(got, ref_out)
        row.append(float(np.median(ts[2:])))
    best = np.nanmin(row)
    alg = R * (cin + 2 * cout) * 4 + R * 8 + Kk * cin * cout * 4
    print(("(%d,%d,%d,%d)" % (cin, cout, n_out, R)).ljust(34) + "".join(("%.1f" % t).rjust(9) for t in row)
          + "   %.0f" % (alg / best / 1e3))
