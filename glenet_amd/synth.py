"""Deterministic synthetic KITTI- / Waymo-shaped LiDAR frames.

No dataset is available offline, so benches and tests draw frames from this generator:
64 beams, each ray hits the ground plane or one of K random car boxes, plus clutter.
numpy only (host side data source; not part of the measured path).
"""
import numpy as np

KITTI = dict(point_cloud_range=[0.0, -40.0, -3.0, 70.4, 40.0, 1.0], voxel_size=[0.05, 0.05, 0.1],
             max_points=5, max_voxels_train=16000, max_voxels_test=40000, num_features=4)
WAYMO = dict(point_cloud_range=[-75.2, -75.2, -2.0, 75.2, 75.2, 4.0], voxel_size=[0.1, 0.1, 0.15],
             max_points=5, max_voxels_train=150000, max_voxels_test=150000, num_features=5)


def _ray_box_hits(origins_dirs, boxes):
    """Nearest hit distance of rays (R,3 unit dirs from the sensor origin) with rotated boxes
    (K,7).  Slab method in each box frame.  Returns t (R,) with inf where no hit."""
    d = origins_dirs
    t_best = np.full(len(d), np.inf)
    for bx in boxes:
        c, s = np.cos(-bx[6]), np.sin(-bx[6])
        rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        o = rot @ (-bx[:3])
        dd = d @ rot.T
        half = bx[3:6] / 2
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (-half - o) / dd
            t2 = (half - o) / dd
        tmin = np.nanmax(np.minimum(t1, t2), axis=1)
        tmax = np.nanmin(np.maximum(t1, t2), axis=1)
        hit = (tmax >= tmin) & (tmin > 0)
        t_best = np.where(hit & (tmin < t_best), tmin, t_best)
    return t_best


def make_boxes(rng, k, x_range, y_range, z_center, size=(3.9, 1.6, 1.56)):
    b = np.zeros((k, 7))
    b[:, 0] = rng.uniform(*x_range, k)
    b[:, 1] = rng.uniform(*y_range, k)
    b[:, 3:6] = np.asarray(size) * rng.uniform(0.9, 1.1, (k, 3))
    b[:, 2] = z_center + b[:, 5] / 2
    b[:, 6] = rng.uniform(-np.pi, np.pi, k)
    return b


def kitti_frame(frame_id, num_points=20000, num_boxes=15):
    """(P,4) float32 [x,y,z,intensity], shuffled; and the (K,7) car boxes."""
    rng = np.random.default_rng(1000 + frame_id)
    elev = np.deg2rad(np.linspace(-24.8, 2.0, 64))
    az = np.deg2rad(np.arange(-45.0, 45.0, 0.16))
    azj = az[None, :] + np.deg2rad(rng.uniform(-0.04, 0.04, (64, len(az))))
    el = np.repeat(elev[:, None], len(az), 1)
    d = np.stack([np.cos(el) * np.cos(azj), np.cos(el) * np.sin(azj), np.sin(el)], -1).reshape(-1, 3)
    ground_z = -1.73
    boxes = make_boxes(rng, num_boxes, (5, 65), (-30, 30), ground_z)
    with np.errstate(divide="ignore"):
        t_ground = np.where(d[:, 2] < 0, ground_z / d[:, 2], np.inf)
    t = np.minimum(t_ground, _ray_box_hits(d, boxes))
    ok = np.isfinite(t) & (t < 120)
    t = t[ok] + rng.normal(0, 0.02, ok.sum())
    pts = d[ok] * t[:, None]
    r = KITTI["point_cloud_range"]
    inside = ((pts[:, 0] >= r[0]) & (pts[:, 0] < r[3]) & (pts[:, 1] >= r[1]) & (pts[:, 1] < r[4]) &
              (pts[:, 2] >= r[2]) & (pts[:, 2] < r[5]))
    pts = pts[inside]
    n_clutter = int(0.05 * num_points)
    clutter = rng.uniform(r[0:3], r[3:6], (n_clutter, 3))
    pts = np.concatenate([pts, clutter])
    perm = rng.permutation(len(pts))
    if len(pts) >= num_points:
        pts = pts[perm[:num_points]]
    else:
        extra = rng.integers(0, len(pts), num_points - len(pts))
        pts = np.concatenate([pts[perm], pts[extra] + rng.normal(0, 0.01, (len(extra), 3))])
    inten = rng.uniform(0, 1, (num_points, 1))
    out = np.concatenate([pts, inten], 1).astype(np.float32)
    out = out[rng.permutation(num_points)]   # data_processor.shuffle_points
    return out, boxes.astype(np.float32)


def gt_uncertainty(frame_id, num_boxes=15):
    """(K,7) label variances of a frame's boxes, U(0.01, 0.2) (the `gt_uncertaintys` GLENet's
    CVAE stage writes next to the ground truth)."""
    rng = np.random.default_rng(7000 + frame_id)
    return rng.uniform(0.01, 0.2, (num_boxes, 7)).astype(np.float32)


def waymo_frame(frame_id, num_points=180000, num_boxes=60):
    """(P,5) float32 [x,y,z,intensity,elongation]; Waymo-shaped range, full 360 degrees."""
    rng = np.random.default_rng(5000 + frame_id)
    elev = np.deg2rad(np.linspace(-17.6, 2.4, 64))
    az = np.deg2rad(np.arange(-180.0, 180.0, 0.125))
    azj = az[None, :] + np.deg2rad(rng.uniform(-0.03, 0.03, (64, len(az))))
    el = np.repeat(elev[:, None], len(az), 1)
    d = np.stack([np.cos(el) * np.cos(azj), np.cos(el) * np.sin(azj), np.sin(el)], -1).reshape(-1, 3)
    sensor_h = 2.0
    boxes = make_boxes(rng, num_boxes, (-70, 70), (-70, 70), -sensor_h)
    with np.errstate(divide="ignore"):
        t_ground = np.where(d[:, 2] < 0, -sensor_h / d[:, 2], np.inf)
    t = np.minimum(t_ground, _ray_box_hits(d, boxes))
    ok = np.isfinite(t) & (t < 75)
    t = t[ok] + rng.normal(0, 0.02, ok.sum())
    pts = d[ok] * t[:, None]
    pts[:, 2] += sensor_h
    boxes[:, 2] += sensor_h
    r = WAYMO["point_cloud_range"]
    inside = ((pts[:, 0] >= r[0]) & (pts[:, 0] < r[3]) & (pts[:, 1] >= r[1]) & (pts[:, 1] < r[4]) &
              (pts[:, 2] >= r[2]) & (pts[:, 2] < r[5]))
    pts = pts[inside]
    clutter = rng.uniform(r[0:3], r[3:6], (int(0.03 * num_points), 3))
    pts = np.concatenate([pts, clutter])
    perm = rng.permutation(len(pts))
    if len(pts) >= num_points:
        pts = pts[perm[:num_points]]
    else:
        extra = rng.integers(0, len(pts), num_points - len(pts))
        pts = np.concatenate([pts[perm], pts[extra] + rng.normal(0, 0.01, (len(extra), 3))])
    feat = rng.uniform(0, 1, (num_points, 2))
    out = np.concatenate([pts, feat], 1).astype(np.float32)
    return out[rng.permutation(num_points)], boxes.astype(np.float32)


def random_boxes(rng, n, xy_range=40.0, near_dup=0.3):
    """(n,7) boxes for IoU / NMS tests: a mix of independent boxes and jittered duplicates
    so that overlaps cover the whole (0,1) range."""
    b = np.zeros((n, 7), np.float32)
    b[:, 0] = rng.uniform(0, xy_range, n)
    b[:, 1] = rng.uniform(-xy_range / 2, xy_range / 2, n)
    b[:, 2] = rng.uniform(-2, 0, n)
    b[:, 3] = rng.uniform(3.2, 4.6, n)
    b[:, 4] = rng.uniform(1.4, 1.9, n)
    b[:, 5] = rng.uniform(1.3, 1.8, n)
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    ndup = int(n * near_dup)
    if ndup and n > 1:
        src = rng.integers(0, n, ndup)
        dst = rng.choice(n, ndup, replace=False)
        b[dst] = b[src]
        b[dst, 0:2] += rng.normal(0, 0.3, (ndup, 2)).astype(np.float32)
        b[dst, 3:6] *= rng.uniform(0.9, 1.1, (ndup, 3)).astype(np.float32)
        b[dst, 6] += rng.normal(0, 0.1, ndup).astype(np.float32)
    return b


def cvae_objects(n, seed=2000, num_points=512, with_labels=False):
    """Synthetic CVAE crops (BASELINE configs[3]): per object `num_points` samples (with replacement) of a
    car-box surface in object coordinates (points relative to the box centre, box turned by a random yaw), normalised
    as the reference dataset does -- (x - mean) / diag, (y - mean) / diag, (z - mean) / 1.56 with
    diag = sqrt(3.9^2 + 1.6^2), transposed to (4, P) (cvae_uncertainty/dataset.py:360-397) -- plus an intensity.
    with_labels: also gt_boxes (n, 7) = [-mean / diag, -mean / diag, -mean / 1.56, log(dx / 3.9), log(dy / 1.6),
    log(dz / 1.56), yaw] and gt_boxes_input (n, 8) = the same with the yaw as (sin, cos) (dataset.py:406-424)."""
    rng = np.random.default_rng(seed)
    size = np.array([3.9, 1.6, 1.56]) * rng.uniform(0.9, 1.1, (n, 1, 3))
    face = rng.integers(0, 3, (n, num_points))
    p = rng.uniform(-0.5, 0.5, (n, num_points, 3))
    sign = rng.choice([-0.5, 0.5], (n, num_points))
    for a in range(3):
        p[..., a] = np.where(face == a, sign, p[..., a])
    p = p * size
    yaw = rng.uniform(-np.pi, np.pi, (n, 1))
    c, s = np.cos(yaw), np.sin(yaw)
    x, y = p[..., 0] * c - p[..., 1] * s, p[..., 0] * s + p[..., 1] * c
    diag = np.sqrt(3.9 ** 2 + 1.6 ** 2)
    mx, my, mz = x.mean(1, keepdims=True), y.mean(1, keepdims=True), p[..., 2].mean(1, keepdims=True)
    pts = np.stack([(x - mx) / diag, (y - my) / diag, (p[..., 2] - mz) / 1.56, rng.uniform(0, 1, (n, num_points))], 1)
    pts = pts.astype(np.float32)                               # (n, 4, P)
    if not with_labels:
        return pts
    box7 = np.concatenate([-mx / diag, -my / diag, -mz / 1.56, np.log(size[:, 0, 0:1] / 3.9), np.log(size[:, 0, 1:2] / 1.6),
                           np.log(size[:, 0, 2:3] / 1.56), yaw], 1).astype(np.float32)
    box8 = np.concatenate([box7[:, :6], np.sin(box7[:, 6:7]), np.cos(box7[:, 6:7])], 1).astype(np.float32)
    return pts, box8, box7
