"""Shared by the CPU and GPU tests of the NMS predicate: which pairs of tests/golden/nms_pred_ref.npz the
cross-library identity covers (see make_golden.make_nms_predicate_ref)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load():
    return np.load(os.path.join(GOLD, "nms_pred_ref.npz"))


def _corners(b):
    """(n, 4, 2) rotated corners in float64, heading convention of iou3d_nms ((cos, -sin; sin, cos))."""
    b = b.astype(np.float64)
    hx, hy = b[:, 3] / 2, b[:, 4] / 2
    loc = np.stack([np.stack([-hx, -hy], 1), np.stack([hx, -hy], 1), np.stack([hx, hy], 1), np.stack([-hx, hy], 1)], 1)
    c, s = np.cos(b[:, 6]), np.sin(b[:, 6])
    x = loc[..., 0] * c[:, None] - loc[..., 1] * s[:, None] + b[:, None, 0]
    y = loc[..., 0] * s[:, None] + loc[..., 1] * c[:, None] + b[:, None, 1]
    return np.stack([x, y], -1)


def _outside_distance(box, pts):
    """Chebyshev 'how far outside box' of pts (m, k, 2) for every box (n,): (n, m, k); negative = inside."""
    b = box.astype(np.float64)
    dx = pts[None, ..., 0] - b[:, None, None, 0]
    dy = pts[None, ..., 1] - b[:, None, None, 1]
    c, s = np.cos(-b[:, 6])[:, None, None], np.sin(-b[:, 6])[:, None, None]
    rx = dx * c - dy * s
    ry = dx * s + dy * c
    return np.maximum(np.abs(rx) - b[:, None, None, 3] / 2, np.abs(ry) - b[:, None, None, 4] / 2)


def margin_safe(a7, b7, lo=-1e-4, hi=1.2e-2):
    """(n, m) bool: no corner of either box within (lo, hi) of the other's boundary -- outside that band the
    inside tests of the two libraries (margins 1e-5 and 1e-2) give the same answer."""
    ca, cb = _corners(a7), _corners(b7)
    d_ab = _outside_distance(a7, cb)                       # corners of b against boxes a: (n, m, 4)
    d_ba = _outside_distance(b7, ca).transpose(1, 0, 2)    # corners of a against boxes b -> (n, m, 4)
    bad = ((d_ab > lo) & (d_ab < hi)).any(-1) | ((d_ba > lo) & (d_ba < hi)).any(-1)
    return ~bad
