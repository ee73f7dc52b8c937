"""Per-block timeline of the sparse-conv block kernel on the real layer shapes (GPU box):
where do the tiles run, how long do they take, how well does chunk count predict time."""
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
TR = int(os.environ.get("TR", "64"))
NW = int(os.environ.get("NW", "4"))
VARIANT = int(os.environ.get("VARIANT", "-1"))   # e.g. VARIANT=20 NW=8 for the column-split tile
_set_variant(VARIANT)
_lib.call_nostream("glx_sconv_set_xcd_group", int(os.environ.get("ABLATE", "0"), 0))   # 0x100 / 0x200: TRACE-build ablations
seen = set()
for f, w, nbr, order, n_out, rules in calls:
    Kk, cin, cout = w.shape
    key = (cin, cout, n_out, Kk)
    if key in seen or cout >= 128 or (os.environ.get("ONLY6464") and (cin, cout) != (64, 64)):
        continue
    seen.add(key)
    packed = sp.pack_weights(w)
    nblk = (n_out + TR - 1) // TR
    REC = 4 + 8 * NW
    trace = torch.zeros((nblk, REC), dtype=torch.int64, device=dev)
    for it in range(3):
        orig(f, w, None, nbr, order, n_out, packed=packed)
    _lib.call_nostream("glx_sconv_set_trace", ctypes.c_void_p(trace.data_ptr()))
    orig(f, w, None, nbr, order, n_out, packed=packed)
    torch.cuda.synchronize()
    _lib.call_nostream("glx_sconv_set_trace", None)
    t = trace.cpu().numpy()
    t0 = t[:, 0].min()
    st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0      # us
    dur = en - st
    hw = t[:, 2] & 0xFFFFFFFF
    xcc = t[:, 2] >> 32
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ch = t[:, 3]
    if os.environ.get("MAPPING"):
        # how the hardware deals blocks to CUs: is block b + 256 on the CU of block b ?
        same = np.mean([cuid[b] == cuid[b + 256] for b in range(nblk - 256)])
        same8 = np.mean([xcc[b] == b % 8 for b in range(nblk)])
        print("  block -> CU: xcc == b %% 8 for %.2f of the blocks; CU(b) == CU(b+256) for %.2f; first 16 CU ids %s"
              % (same8, same, cuid[:16].tolist()))
        # what a perfectly balanced assignment of the same tiles could reach
        order_ = np.argsort(-ch)
        load = np.zeros(256)
        for tix in order_:
            load[np.argmin(load)] += ch[tix]
        print("  chunks per CU now max/mean %.2f; greedy longest-first assignment max/mean %.2f"
              % (np.bincount(cuid, weights=ch).max() / (ch.sum() / 256), load.max() / (ch.sum() / 256)))
        for name, rev in (("snake (alternate direction per round)", lambda r: r % 2 == 1),
                          ("snake, last round by current load", None)):
            sl = np.zeros(256)
            for r0 in range(0, nblk, 256):
                tiles_r = order_[r0:r0 + 256]
                m = len(tiles_r)
                if rev is None and r0 + 256 >= nblk and r0 > 0:
                    cus = np.argsort(sl[:m], kind="stable")
                elif (rev or (lambda r: r % 2 == 1))(r0 // 256):
                    cus = np.arange(m)[::-1]
                else:
                    cus = np.arange(m)
                sl[cus] += ch[tiles_r]
            print("  %s: max/mean %.2f" % (name, sl.max() / (ch.sum() / 256)))
        # longest-first by rounds, CUs that get one block fewer served first in round 0
        sl = np.zeros(256)
        m_last = nblk - (nblk - 1) // 256 * 256
        for r0 in range(0, nblk, 256):
            tiles_r = order_[r0:r0 + 256]
            m = len(tiles_r)
            if r0 == 0:
                cus = np.concatenate([np.arange(m_last, 256), np.arange(0, m_last)])[:m]
            else:
                cus = np.argsort(sl[:m], kind="stable")
            sl[cus] += ch[tiles_r]
        print("  rounds, short CUs first then by load: max/mean %.2f" % (sl.max() / (ch.sum() / 256)))
        # cheapest variant: identity for all rounds but the last, which is dealt by load
        sl = np.zeros(256)
        last0 = (nblk - 1) // 256 * 256
        for b_ in range(last0):
            sl[b_ % 256] += ch[b_]
        tl = np.arange(last0, nblk)
        tl = tl[np.argsort(-ch[tl], kind="stable")]
        cus = np.argsort(sl[:len(tl)], kind="stable")
        sl[cus] += ch[tl]
        print("  identity + last round dealt by load: max/mean %.2f" % (sl.max() / (ch.sum() / 256)))
    print("layer (%d,%d) N=%d K=%d blocks=%d  span %.1f us  chunks total %d (mean %.1f, max %d)"
          % (cin, cout, n_out, Kk, nblk, en.max(), ch.sum(), ch.mean(), ch.max()))
    print("  block start: p50 %.1f p90 %.1f max %.1f us | duration: mean %.1f p50 %.1f p90 %.1f max %.1f us"
          % (np.percentile(st, 50), np.percentile(st, 90), st.max(), dur.mean(), np.percentile(dur, 50),
             np.percentile(dur, 90), dur.max()))
    ids, inv = np.unique(cuid, return_inverse=True)
    per_cu_blocks = np.bincount(inv)
    per_cu_chunks = np.bincount(inv, weights=ch)
    per_cu_end = np.zeros(len(ids))
    np.maximum.at(per_cu_end, inv, en)
    print("  CUs used %d | blocks/CU min %d max %d | chunks/CU mean %.0f max %.0f | CU finish p10 %.1f p50 %.1f p90 %.1f max %.1f us"
          % (len(ids), per_cu_blocks.min(), per_cu_blocks.max(), per_cu_chunks.mean(), per_cu_chunks.max(),
             np.percentile(per_cu_end, 10), np.percentile(per_cu_end, 50), np.percentile(per_cu_end, 90),
             per_cu_end.max()))
    c = np.corrcoef(ch, dur)[0, 1]
    cc = np.corrcoef(per_cu_chunks, per_cu_end)[0, 1]
    print("  corr(chunks, block duration) %.2f | corr(chunks on CU, CU finish) %.2f | us per chunk (CU level) %.3f"
          % (c, cc, (per_cu_end / np.maximum(per_cu_chunks, 1)).mean()))
    wv = t[:, 4:].reshape(nblk, NW, 8).astype(np.float64)
    ph = wv[:, :, :5]                       # cycles: issue | multiply | barrier1 | store | barrier2
    tot = ph.sum(axis=2)
    wch = wv[:, :, 5]
    names = ["issue prefetch", "multiply", "barrier1", "stage store (+wait)", "barrier2"]
    print("  per-wave loop cycles: mean %.0f (%.1f us at 2.4 GHz); chunks per wave mean %.1f"
          % (tot.mean(), tot.mean() / 2400, wch.mean()))
    print("  phase share: " + ", ".join("%s %.0f%%" % (n, 100 * ph[:, :, i].sum() / tot.sum())
                                        for i, n in enumerate(names)))
    nch = max(wch.sum(), 1)
    if os.environ.get("GEMM"):
        print("  setup before the loop: %.0f cycles; steps per block %.1f; per-step cycles %.0f"
              % (wv[:, :, 6].mean(), wv[:, :, 7].mean(), tot.mean() / max(wv[:, :, 7].mean(), 1)))
        # clock64 counts shader cycles, the block start / end are 100 MHz wall ticks: cycles of the loop over
        # its wall time = the clock the CUs ran at (the epilogue after the loop is in the wall time: lower bound)
        setup_us = wv[:, 0, 6] / 22.0 / 100.0
        loop_us = np.maximum(dur - setup_us, 1e-3)
        print("  setup %.1f us of a block's %.1f us; loop cycles / loop wall time = %.2f GHz (p10 %.2f p90 %.2f)"
              % (setup_us.mean(), dur.mean(), (tot[:, 0] / loop_us / 1e3).mean(),
                 np.percentile(tot[:, 0] / loop_us / 1e3, 10), np.percentile(tot[:, 0] / loop_us / 1e3, 90)))
        continue
    print("  multiply cycles per chunk (waves with chunks): %.0f = wait for rows %.0f + MFMA loop %.0f + accumulate %.0f"
          % (ph[:, :, 1].sum() / nch, wv[:, :, 6].sum() / nch, wv[:, :, 7].sum() / nch,
             (ph[:, :, 1].sum() - wv[:, :, 6].sum() - wv[:, :, 7].sum()) / nch))
    nx = len(np.unique(xcc))
    per_x = np.bincount(xcc.astype(int), weights=ch, minlength=8)
    print("  XCDs %d chunks/XCD %s" % (nx, per_x.astype(int).tolist()))
