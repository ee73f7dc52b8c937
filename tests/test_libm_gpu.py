"""The NMS predicate on the device, bit for bit.

1. csrc/glx_libm.h on the DEVICE == the host libm the reference's CPU path calls (sinf / cosf / atanf / atan2f),
   on strided sweeps of all floats, a dense sweep of the heading range and random pairs.
2. The device's rotated overlap / IoU == the reference build's values of tests/golden/nms_pred_ref.npz (cross-library
   identity, see tests/golden/make_golden.py:make_nms_predicate_ref) on every pair outside the margin band, and ==
   the oracle on EVERY pair -- all bits, no "fraction bit-identical" any more.
3. A small census (tools/nms_census.py; the 200-seed run is profiles/r03_nms_census.json): 9000-box frames with a
   constructed near-threshold family at thr 0.8 / 0.7 / 0.1 / 0.01 -- zero IoU bit differences, zero threshold-side
   disagreements, identical keep lists."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from glenet_amd import _lib, synth
from glenet_amd.pcdet_ops.iou3d_nms import iou3d_nms_cuda

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _dev_eval(fn, x, dev, y=None):
    tx = T(x, dev)
    ty = T(y, dev) if y is not None else tx
    out = torch.empty_like(tx)
    _lib.call("glx_libm_eval", {"sin": 0, "cos": 1, "atan": 2, "atan2": 3}[fn], tx, ty, _lib.c_int64(tx.numel()), out)
    return out.cpu().numpy()


def _same_bits(a, b):
    nan = np.isnan(a) & np.isnan(b)
    return int(((a.view(np.uint32) != b.view(np.uint32)) & ~nan).sum())


@pytest.mark.parametrize("fn", ["sin", "cos", "atan"])
def test_device_trig_equals_host_libm(dev, fn):
    oracle.lib().orc_set_threads(max(1, len(os.sched_getaffinity(0))))
    try:
        u = np.arange(0, 0x7f800000, 61, dtype=np.uint32)                       # every 61st finite float, both signs
        x = np.concatenate([u, u | np.uint32(0x80000000)]).view(np.float32)
        assert _same_bits(_dev_eval(fn, x, dev), oracle.libm_eval(fn, x)) == 0
        lo, hi = np.float32(-7.0).view(np.uint32), np.float32(7.0).view(np.uint32)
        h = np.concatenate([np.arange(np.float32(2.0 ** -20).view(np.uint32), hi, 3, dtype=np.uint32),
                            np.arange(np.uint32(0x80000000) | np.float32(2.0 ** -20).view(np.uint32), lo, 3, dtype=np.uint32)])
        h = h.view(np.float32)                                                     # the heading range, every 3rd float
        assert len(h) > 1.2e8 and _same_bits(_dev_eval(fn, h, dev), oracle.libm_eval(fn, h)) == 0
    finally:
        oracle.lib().orc_set_threads(1)


def test_device_atan2_equals_host_libm(dev):
    oracle.lib().orc_set_threads(max(1, len(os.sched_getaffinity(0))))
    try:
        rng = np.random.default_rng(5)
        n = 1 << 25
        y = rng.uniform(-8, 8, n).astype(np.float32)
        x = rng.uniform(-8, 8, n).astype(np.float32)
        y[::7] = (y[::7] * 1e-4).astype(np.float32)                             # near the axes
        x[3::11] = (x[3::11] * 1e-5).astype(np.float32)
        y[5::1001] = 0.0
        x[6::1003] = 0.0
        x[7::997] = 1.0
        assert _same_bits(_dev_eval("atan2", y, dev, x), oracle.libm_eval("atan2", y, x)) == 0
        a = rng.integers(0, 1 << 32, 1 << 24, dtype=np.uint64).astype(np.uint32).view(np.float32)   # any bit patterns
        b = rng.integers(0, 1 << 32, 1 << 24, dtype=np.uint64).astype(np.uint32).view(np.float32)
        assert _same_bits(_dev_eval("atan2", a, dev, b), oracle.libm_eval("atan2", a, b)) == 0
    finally:
        oracle.lib().orc_set_threads(1)


def test_nms_predicate_iou_equals_the_reference_build(dev):
    import nms_pred_util as u
    g = u.load()
    safe = u.margin_safe(g["a7"], g["b7"])
    a, b = T(g["a7"], dev), T(g["b7"], dev)
    ov = torch.zeros(a.shape[0], b.shape[0], device=dev)
    iou = torch.zeros_like(ov)
    iou3d_nms_cuda.boxes_overlap_bev_gpu(a, b, ov)
    iou3d_nms_cuda.boxes_iou_bev_gpu(a, b, iou)
    ov, iou = ov.cpu().numpy(), iou.cpu().numpy()
    assert np.array_equal(ov.view(np.uint32)[safe], g["overlap"].view(np.uint32)[safe])      # reference-executed values
    assert np.array_equal(iou.view(np.uint32)[safe], g["iou"].view(np.uint32)[safe])
    assert np.array_equal(ov.view(np.uint32), oracle.boxes_overlap_bev(g["a7"], g["b7"]).view(np.uint32))   # every pair
    assert ((g["overlap"] > 0) & safe).sum() > 2000


@pytest.mark.parametrize("seed", [0, 1])
def test_rotated_iou_all_bits_vs_oracle(dev, seed):
    rng = np.random.default_rng(seed)
    a, b = synth.random_boxes(rng, 1200, near_dup=0.6), synth.random_boxes(rng, 900, near_dup=0.6)
    b[:300] = a[:300] + rng.normal(0, 0.1, (300, 7)).astype(np.float32)
    a[7, 3] = 0.0                                                                 # a degenerate box
    b[11] = a[11]                                                                 # coincident boxes
    oracle.lib().orc_set_threads(max(1, len(os.sched_getaffinity(0))))
    try:
        ref = oracle.boxes_iou_bev(a, b)
    finally:
        oracle.lib().orc_set_threads(1)
    got = torch.zeros(len(a), len(b), device=dev)
    iou3d_nms_cuda.boxes_iou_bev_gpu(T(a, dev), T(b, dev), got)
    got = got.cpu().numpy()
    assert (ref > 0.01).sum() > 2000
    assert _same_bits(got, ref) == 0


def test_nms_census_small(dev):
    import nms_census
    oracle.lib().orc_set_threads(max(1, len(os.sched_getaffinity(0))))
    try:
        rep = nms_census.census(range(3), n=9000, twins_per_thr=400)
    finally:
        oracle.lib().orc_set_threads(1)
    assert rep["frames"] == 3 and rep["candidate_pairs"] > 500000 and rep["pairs_total"] == 3 * 9000 * 8999 // 2
    assert min(rep["near_threshold_pairs_1e-6"].values()) > 100, rep            # the constructed family is there
    assert rep["noncandidate_nonzero_on_device"] == 0
    assert rep["iou_bits_differ"] == 0, rep
    assert sum(rep["threshold_side_disagreements"].values()) == 0
    assert sum(rep["keep_lists_differ"].values()) == 0 and rep["keep_lists_compared"] == 12
