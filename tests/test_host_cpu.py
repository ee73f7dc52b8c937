"""libglenet_host.so (include/glenet_host.h): the three entry points the reference calls from forked DataLoader
workers -- boxes_iou_bev_cpu, points_in_boxes_cpu, the hard voxelizer behind spconv.utils.VoxelGeneratorV2 /
Point2VoxelCPU3d -- plus the iou3d library's CPU twins, as host C++ that never touches the GPU runtime.
Checked bit for bit against the oracle and, for the iou3d convention, against the fixture produced by the
reference's own compiled iou3d_cpu.cpp (tests/golden/iou3d_ref.npz)."""
import ctypes
import multiprocessing as mp
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import oracle
from glenet_amd import _host, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _boxes(seed, n, **kw):
    return synth.random_boxes(np.random.default_rng(seed), n, **kw)


def test_host_library_exports_its_header_and_links_no_gpu_runtime():
    hdr = open(os.path.join(ROOT, "include", "glenet_host.h")).read()
    names = sorted(set(re.findall(r"\b(glxh_[a-z0-9_]+)\s*\(", hdr)))
    lib = _host.load()
    assert len(names) >= 6
    for n in names:
        assert hasattr(lib, n), n
    assert lib.glxh_abi_version() >= 1
    deps = subprocess.run(["ldd", _host.LIB_PATH], capture_output=True, text=True).stdout
    assert not re.search(r"amdhip|hsa-runtime|libtorch|libc10", deps), deps


def test_boxes_iou_bev_bit_exact_vs_oracle_including_degenerate_boxes():
    a, b = _boxes(1, 300, near_dup=0.6), _boxes(2, 200, near_dup=0.6)
    b[:60] = a[:60]                                    # coincident boxes (self IoU is not exactly 1)
    a[10:20, 3:5] = 0                                  # zero-area
    a[20:25, 6] = 0
    b[60:65] = a[20:25]
    b[60:65, 0] += a[20:25, 3]                         # touching edges (axis aligned)
    got = _host.boxes_iou_bev(a, b)
    want = oracle.boxes_iou_bev(a, b)
    assert got.dtype == np.float32 and np.array_equal(got, want, equal_nan=True)
    assert _host.boxes_iou_bev(a[:0], b).shape == (0, 200)


def test_iou3d_library_cpu_twins_vs_reference_build_golden():
    g = np.load(os.path.join(GOLD, "iou3d_ref.npz"))
    for kind in ("random", "degenerate"):
        a, b = g["%s_a" % kind], g["%s_b" % kind]
        assert np.array_equal(_host.iou3d_boxes_bev(a, b, iou=False), g["%s_overlap" % kind], equal_nan=True)
        assert np.array_equal(_host.iou3d_boxes_bev(a, b, iou=True), g["%s_iou" % kind], equal_nan=True)


def test_points_in_boxes_bit_exact_vs_oracle_with_margin_cases():
    rng = np.random.default_rng(3)
    boxes = _boxes(4, 40)
    pts = rng.uniform([0, -20, -3], [40, 20, 1], (6000, 3)).astype(np.float32)
    # points on / just off the faces, inside the 1e-2 margin and at |dz| == dz/2
    for i in range(20):
        c, s = np.cos(boxes[i, 6]), np.sin(boxes[i, 6])
        for k, off in enumerate((boxes[i, 3] / 2 + 0.009, boxes[i, 3] / 2 + 0.011, boxes[i, 3] / 2)):
            pts[100 * i + k] = [boxes[i, 0] + off * c, boxes[i, 1] + off * s, boxes[i, 2] + boxes[i, 5] / 2]
    got = _host.points_in_boxes(boxes, pts)
    want = oracle.points_in_boxes_cpu(pts, boxes)
    assert got.dtype == np.int32 and np.array_equal(got, want)
    assert got.sum() > 0


@pytest.mark.parametrize("max_voxels,max_points", [(16000, 5), (700, 3), (40000, 1)])
def test_host_voxelizer_bit_exact_vs_oracle(max_voxels, max_points):
    K = synth.KITTI
    pts, _ = synth.kitti_frame(4)
    pts[:50, 0] = 70.4                                   # on / outside the range boundary
    pts[50:60, 2] = np.float32(1.0)
    v, c, n = _host.voxelize_hard(pts, K["voxel_size"], K["point_cloud_range"], max_points, max_voxels)
    ov, oc, on = oracle.voxelize_hard(pts, K["voxel_size"], K["point_cloud_range"], max_points, max_voxels)
    assert np.array_equal(v, ov) and np.array_equal(c, oc) and np.array_equal(n, on)
    e = _host.voxelize_hard(pts[:0], K["voxel_size"], K["point_cloud_range"], 5, 100)
    assert e[0].shape == (0, 5, 4) and e[1].shape == (0, 3)


def test_reference_facing_wrappers_take_host_tensors():
    """The drop-in modules under the reference's import names route the *_cpu calls to the host library."""
    from glenet_amd.pcdet_ops.iou3d_nms import iou3d_nms_cuda, iou3d_nms_utils
    from glenet_amd.pcdet_ops.roiaware_pool3d import roiaware_pool3d_utils
    from glenet_amd.spconv import utils as sputils
    a, b = torch.from_numpy(_boxes(5, 30)), torch.from_numpy(_boxes(6, 20))
    out = torch.zeros(30, 20)
    assert iou3d_nms_cuda.boxes_iou_bev_cpu(a, b, out) == 1
    assert np.array_equal(out.numpy(), oracle.boxes_iou_bev(a.numpy(), b.numpy()))
    got = iou3d_nms_utils.boxes_bev_iou_cpu(a.numpy(), b.numpy())          # numpy in, numpy out (:52-68)
    assert isinstance(got, np.ndarray) and np.array_equal(got, out.numpy())
    pts = torch.rand(500, 3) * 40
    idx = roiaware_pool3d_utils.points_in_boxes_cpu(pts, a)
    assert idx.shape == (30, 500) and np.array_equal(idx.numpy(), oracle.points_in_boxes_cpu(pts.numpy(), a.numpy()))
    K = synth.KITTI
    gen = sputils.Point2VoxelCPU3d(K["voxel_size"], K["point_cloud_range"], 4, 16000, 5)
    frame = synth.kitti_frame(7)[0]
    v, c, n = gen.point_to_voxel(frame)
    ov, oc, on = oracle.voxelize_hard(frame, K["voxel_size"], K["point_cloud_range"], 5, 16000)
    assert np.array_equal(v.numpy(), ov) and np.array_equal(c.numpy(), oc) and np.array_equal(n.numpy(), on)


def _forked_worker(q, seed):
    try:
        a, b = _boxes(seed, 64), _boxes(seed + 1, 48)
        pts = np.random.default_rng(seed).uniform([0, -20, -3], [40, 20, 1], (2000, 3)).astype(np.float32)
        K = synth.KITTI
        v, c, n = _host.voxelize_hard(synth.kitti_frame(seed, num_points=4000)[0], K["voxel_size"],
                                      K["point_cloud_range"], 5, 16000)
        q.put((seed, float(_host.boxes_iou_bev(a, b).sum()), int(_host.points_in_boxes(a, pts).sum()), len(c)))
    except Exception as e:       # noqa: BLE001
        q.put((seed, repr(e)))


def _run_forked(seeds):
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    ps = [ctx.Process(target=_forked_worker, args=(q, s)) for s in seeds]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return res


def _expected(seed):
    a, b = _boxes(seed, 64), _boxes(seed + 1, 48)
    pts = np.random.default_rng(seed).uniform([0, -20, -3], [40, 20, 1], (2000, 3)).astype(np.float32)
    K = synth.KITTI
    c = oracle.voxelize_hard(synth.kitti_frame(seed, num_points=4000)[0], K["voxel_size"], K["point_cloud_range"], 5, 16000)[1]
    return (seed, float(oracle.boxes_iou_bev(a, b).sum()), int(oracle.points_in_boxes_cpu(pts, a).sum()), len(c))


def test_forked_workers_call_the_host_entry_points():
    """DataLoader-style: fork()ed children (the parent has torch loaded) compute concurrently."""
    _host.load()
    assert _run_forked([11, 12, 13]) == [_expected(s) for s in (11, 12, 13)]


@pytest.mark.gpu
def test_forked_workers_while_the_parent_owns_the_gpu(dev):
    """The reference's situation: the training process holds a HIP context and forks its DataLoader workers,
    which call boxes_bev_iou_cpu / points_in_boxes_cpu / the voxel generator."""
    x = torch.randn(1 << 20, device=dev)
    assert float((x * 2).sum()) == float((x * 2).sum())
    _host.load()
    assert _run_forked([21, 22]) == [_expected(s) for s in (21, 22)]
    assert torch.isfinite(x.sum())                     # the parent's context is still alive
