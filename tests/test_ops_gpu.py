"""GPU parity of the pcdet.ops mirrors (rotated IoU / NMS / voting NMS, point-box operators,
queries and grouping) against the CPU oracle and the reference's golden vectors."""
import copy
import os

import numpy as np
import pytest
import torch

import oracle
from glenet_amd import synth
from glenet_amd.pcdet_ops.iou3d import iou3d_cuda, iou3d_utils
from glenet_amd.pcdet_ops.iou3d_nms import iou3d_nms_cuda, iou3d_nms_utils
from glenet_amd.pcdet_ops.pointnet2.pointnet2_stack import pointnet2_utils, voxel_pool_modules, voxel_query_utils
from glenet_amd.pcdet_ops.roiaware_pool3d import roiaware_pool3d_utils
from glenet_amd.pcdet_ops.roipoint_pool3d import roipoint_pool3d_utils
from glenet_amd.spconv import core as sp

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def ulp_report(a, b):
    same = (a.view(np.uint32) == b.view(np.uint32)).mean()
    return same, np.abs(a - b).max()


# ------------------------------------------------------------------ rotated IoU
@pytest.mark.parametrize("kind", ["random", "axis", "dup", "degenerate"])
def test_iou3d_lib_vs_reference_golden(dev, kind):
    """HIP kernel vs the output of the reference's compiled iou3d_cpu.cpp (committed fixture)."""
    g = np.load(os.path.join(GOLD, "iou3d_ref.npz"))
    a, b = T(g[kind + "_a"], dev), T(g[kind + "_b"], dev)
    ov = torch.zeros(a.shape[0], b.shape[0], device=dev)
    iou = torch.zeros_like(ov)
    iou3d_cuda.boxes_overlap_bev_gpu(a, b, ov)
    iou3d_cuda.boxes_iou_bev_gpu(a, b, iou)
    # fp32 geometry with double-rounded trig vs glibc float trig: a few ulp on areas of O(1..25)
    np.testing.assert_allclose(ov.cpu().numpy(), g[kind + "_overlap"], rtol=1e-5, atol=2e-5)
    fin = np.isfinite(g[kind + "_iou"])
    np.testing.assert_allclose(iou.cpu().numpy()[fin], g[kind + "_iou"][fin], rtol=1e-5, atol=2e-6)
    same, _ = ulp_report(ov.cpu().numpy(), g[kind + "_overlap"])
    assert same > 0.9, "only %.3f of overlaps bit-identical" % same


def test_boxes_iou_bev_and_iou3d_vs_oracle(dev):
    rng = np.random.default_rng(0)
    a, b = synth.random_boxes(rng, 300, near_dup=0.5), synth.random_boxes(rng, 77, near_dup=0.5)
    b[:40] = a[:40] + rng.normal(0, 0.1, (40, 7)).astype(np.float32)
    ref = oracle.boxes_iou_bev(a, b)
    got = iou3d_nms_utils.boxes_iou_bev(T(a, dev), T(b, dev)).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-6)
    same, mx = ulp_report(got, ref)
    assert same > 0.98 and (ref > 0.01).sum() > 30
    got3 = iou3d_nms_utils.boxes_iou3d_gpu(T(a, dev), T(b, dev)).cpu().numpy()
    np.testing.assert_allclose(got3, oracle.boxes_iou3d(a, b), rtol=1e-5, atol=2e-6)
    # host entry point (numpy in / numpy out, like GT sampling calls it)
    got_cpu = iou3d_nms_utils.boxes_bev_iou_cpu(a[:20], b[:20])
    np.testing.assert_allclose(got_cpu, ref[:20, :20], rtol=1e-5, atol=2e-6)
    # empty inputs
    e = iou3d_nms_utils.boxes_iou_bev(T(a[:0], dev), T(b, dev))
    assert e.shape == (0, 77)


def test_aligned_iou3d(dev):
    rng = np.random.default_rng(1)
    a = synth.random_boxes(rng, 200)
    b = a + rng.normal(0, 0.15, a.shape).astype(np.float32)
    got, got_bev = iou3d_utils.boxes_aligned_iou3d_gpu(T(a, dev), T(b, dev), need_bev=True)
    abev = iou3d_utils.boxes3d_to_bev_torch(torch.from_numpy(a)).numpy()
    bbev = iou3d_utils.boxes3d_to_bev_torch(torch.from_numpy(b)).numpy()
    ov = oracle.iou3d_boxes_aligned_overlap_bev(abev, bbev)
    hmin = np.maximum(a[:, 2] - a[:, 5] / 2, b[:, 2] - b[:, 5] / 2)
    hmax = np.minimum(a[:, 2] + a[:, 5] / 2, b[:, 2] + b[:, 5] / 2)
    o3 = ov[:, 0] * np.clip(hmax - hmin, 0, None)
    ref = o3 / np.clip(a[:, 3] * a[:, 4] * a[:, 5] + b[:, 3] * b[:, 4] * b[:, 5] - o3, 1e-7, None)
    np.testing.assert_allclose(got.cpu().numpy()[:, 0], ref, rtol=1e-4, atol=1e-5)
    assert (ref > 0.3).sum() > 50


# ------------------------------------------------------------------ NMS
@pytest.mark.parametrize("n,thr,pre", [(9000, 0.8, None), (2048, 0.7, 1000), (500, 0.1, None), (65, 0.01, None),
                                       (64, 0.5, None), (1, 0.5, None)])
def test_nms_keep_bit_identical(dev, n, thr, pre):
    rng = np.random.default_rng(n)
    boxes = synth.random_boxes(rng, n, xy_range=60.0 if n > 3000 else 25.0, near_dup=0.6)
    scores = (rng.permutation(n).astype(np.float32) + 1) / n          # distinct: sort order is unique
    ref = oracle.nms_gpu(boxes, scores, thr, pre_maxsize=pre)
    got, _ = iou3d_nms_utils.nms_gpu(T(boxes, dev), T(scores, dev), thr, pre_maxsize=pre)
    assert got.dtype == torch.int64
    assert np.array_equal(got.cpu().numpy(), ref)
    assert 0 < len(ref) <= n
    refn = oracle.nms_gpu(boxes, scores, thr, normal=True)
    gotn, _ = iou3d_nms_utils.nms_normal_gpu(T(boxes, dev), T(scores, dev), thr)
    assert np.array_equal(gotn.cpu().numpy(), refn)


def test_nms_raw_extension_signature(dev):
    """iou3d_nms_cuda.nms_gpu(boxes_sorted (cuda), keep (CPU int64), thr) -> count."""
    rng = np.random.default_rng(5)
    boxes = synth.random_boxes(rng, 700, near_dup=0.5)
    keep = torch.zeros(700, dtype=torch.int64)
    num = iou3d_nms_cuda.nms_gpu(T(boxes, dev), keep, 0.3)
    assert np.array_equal(keep[:num].numpy(), oracle.nms_sorted(boxes, 0.3))
    assert iou3d_nms_cuda.nms_gpu(T(boxes[:0], dev), keep, 0.3) == 0


@pytest.mark.parametrize("normal", [False, True])
def test_nms_batched_frames_and_early_stop(dev, normal):
    """glx_nms_batch: every frame's keep list == the oracle's for that frame (bit-identical), and with
    max_keep the kept prefix is unchanged while the sweep may stop early."""
    rng = np.random.default_rng(404)
    F, n, thr = 5, 1500, 0.55
    boxes = np.stack([synth.random_boxes(rng, n, xy_range=8.0 + 10 * f, near_dup=0.6) for f in range(F)])
    full_k, full_n = iou3d_nms_cuda.nms_device_batch(T(boxes, dev), thr, normal=normal)
    cut_k, cut_n = iou3d_nms_cuda.nms_device_batch(T(boxes, dev), thr, normal=normal, max_keep=100)
    for f in range(F):
        ref = oracle.nms_sorted(boxes[f], thr, normal=normal)
        m = int(full_n[f])
        assert m == len(ref) and np.array_equal(full_k[f, :m].cpu().numpy(), ref)
        c = int(cut_n[f])
        assert c >= min(100, m) and c <= m
        assert np.array_equal(cut_k[f, :min(100, m)].cpu().numpy(), ref[:100])
    assert len({int(v) for v in full_n}) > 1                     # frames differ
    k0, n0 = iou3d_nms_cuda.nms_device_batch(T(boxes[:, :0], dev), thr)
    assert n0.tolist() == [0] * F


def test_nms_full_size_properties(dev):
    """Size-independent checks at the training size (9000 proposals): kept set is mutually
    non-overlapping above thr and idempotent."""
    rng = np.random.default_rng(11)
    boxes = synth.random_boxes(rng, 9000, xy_range=70.0, near_dup=0.7)
    scores = torch.from_numpy(rng.random(9000).astype(np.float32)).to(dev)
    b = T(boxes, dev)
    keep, _ = iou3d_nms_utils.nms_gpu(b, scores, 0.8, pre_maxsize=9000)
    kb = b[keep]
    m = iou3d_nms_utils.boxes_iou_bev(kb, kb)
    m.fill_diagonal_(0)
    assert float(m.max()) <= 0.8
    keep2, _ = iou3d_nms_utils.nms_gpu(kb, scores[keep], 0.8)
    assert keep2.numel() == keep.numel()


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_variance_voting_nms_vs_reference_golden(dev, case):
    g = np.load(os.path.join(GOLD, "nms_func_ref.npz"))
    boxes, scores = g["c%d_boxes" % case], g["c%d_scores" % case]
    var = T(g["c%d_var" % case], dev) if ("c%d_var" % case) in g else None
    thr, sthr = g["c%d_params" % case]
    keep, _, new_boxes = iou3d_nms_utils.new_nms_gpu(T(boxes, dev), T(scores, dev), float(thr),
                                                     score_threshold=float(sthr), variance=var)
    assert np.array_equal(np.asarray(keep), g["c%d_keep" % case])
    # voted boxes: fp32 weighted means of O(10) values; tolerance = north_star's 1e-4
    np.testing.assert_allclose(new_boxes, g["c%d_new_boxes" % case], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("sthr", [0.1, 0.0])
def test_variance_voting_nms_larger_vs_oracle(dev, sthr):
    """sthr 0.0 is what class_agnostic_nms passes: suppressed boxes stay in the loop with score 0 and
    are voted too (the parallel tail of the device path); the whole new_boxes array is compared."""
    rng = np.random.default_rng(3)
    n = 1500
    boxes = synth.random_boxes(rng, n, xy_range=30.0, near_dup=0.7)
    scores = (rng.permutation(n).astype(np.float32) + 1) / n
    var = rng.uniform(0.01, 0.3, (n, 7)).astype(np.float32)
    rk, rb = oracle.new_nms_gpu(boxes, scores, 0.1, sthr, var)
    keep, _, nb = iou3d_nms_utils.new_nms_gpu(T(boxes, dev), T(scores, dev), 0.1, score_threshold=sthr,
                                              variance=T(var, dev))
    assert np.array_equal(np.asarray(keep), rk)
    np.testing.assert_allclose(nb, rb, rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ point / box operators
def _scene(rng, n_pts=6000, n_box=24):
    boxes = synth.random_boxes(rng, n_box, xy_range=20.0, near_dup=0.0)
    pts = rng.uniform([-2, -12, -3], [22, 12, 1], (n_pts, 3)).astype(np.float32)
    # put a share of the points inside boxes (incl. exactly on faces / centres)
    for i in range(n_box):
        k = rng.integers(0, n_pts, 60)
        local = rng.uniform(-0.5, 0.5, (60, 3)) * boxes[i, 3:6]
        c, s = np.cos(boxes[i, 6]), np.sin(boxes[i, 6])
        pts[k, 0] = boxes[i, 0] + local[:, 0] * c - local[:, 1] * s
        pts[k, 1] = boxes[i, 1] + local[:, 0] * s + local[:, 1] * c
        pts[k, 2] = boxes[i, 2] + local[:, 2]
    pts[0] = boxes[0, :3]
    pts[1] = boxes[0, :3] + [0, 0, boxes[0, 5] / 2]
    return boxes, pts.astype(np.float32)


def test_points_in_boxes(dev):
    rng = np.random.default_rng(2)
    boxes, pts = _scene(rng)
    B = 2
    bb = np.stack([boxes, np.roll(boxes, 3, 0)])
    pp = np.stack([pts, pts[::-1].copy()])
    got = roiaware_pool3d_utils.points_in_boxes_gpu(T(pp, dev), T(bb, dev)).cpu().numpy()
    ref = oracle.points_in_boxes_gpu(pp, bb)
    assert np.array_equal(got, ref) and (ref >= 0).sum() > 500
    got_cpu = roiaware_pool3d_utils.points_in_boxes_cpu(pts, boxes)       # numpy in/out, MARGIN 1e-2
    assert np.array_equal(got_cpu, oracle.points_in_boxes_cpu(pts, boxes))


@pytest.mark.parametrize("method", ["max", "avg"])
def test_roiaware_pool3d(dev, method):
    rng = np.random.default_rng(4)
    boxes, pts = _scene(rng, n_pts=5000, n_box=16)
    feat = rng.normal(size=(len(pts), 9)).astype(np.float32)
    pooled, argmax, pidx = oracle.roiaware_pool3d_forward(boxes, pts, feat, (3, 2, 2), 6, method)
    mod = roiaware_pool3d_utils.RoIAwarePool3d((3, 2, 2), max_pts_each_voxel=6)
    f = T(feat, dev).requires_grad_(True)
    out = mod(T(boxes, dev), T(pts, dev), f, pool_method=method)
    assert np.array_equal(out.detach().cpu().numpy(), pooled)      # copies / same-order sums: exact
    assert (pidx[..., 0] > 0).sum() > 50 and (pidx[..., 0] == 5).any()     # cap max_pts-1 exercised
    g = rng.normal(size=pooled.shape).astype(np.float32)
    out.backward(T(g, dev))
    gref = oracle.roiaware_pool3d_backward(pidx, argmax, g, len(pts), method)
    np.testing.assert_allclose(f.grad.cpu().numpy(), gref, rtol=1e-5, atol=1e-5)   # float atomics


def test_roipoint_pool3d(dev):
    rng = np.random.default_rng(6)
    boxes, pts = _scene(rng, n_pts=4000, n_box=12)
    boxes[5, :3] = [500, 500, 500]                                    # an empty box
    B = 2
    pp = np.stack([pts, pts[::-1].copy()])
    ff = rng.normal(size=(B, len(pts), 5)).astype(np.float32)
    bb = np.stack([boxes, boxes])
    pooled, empty = oracle.roipoint_pool3d(pp, ff, bb, [1.0, 1.0, 1.0], 128)
    mod = roipoint_pool3d_utils.RoIPointPool3d(num_sampled_points=128, pool_extra_width=[1.0, 1.0, 1.0])
    gp, ge = mod(T(pp, dev), T(ff, dev), T(bb, dev))
    assert np.array_equal(ge.cpu().numpy(), empty) and empty[0, 5] == 1
    assert np.array_equal(gp.cpu().numpy(), pooled)
    # a box with fewer than 128 points wraps around
    mod2 = roipoint_pool3d_utils.RoIPointPool3d(num_sampled_points=512, pool_extra_width=[0.0, 0.0, 0.0])
    p2, e2 = oracle.roipoint_pool3d(pp, ff, bb, [0.0, 0.0, 0.0], 512)
    gp2, ge2 = mod2(T(pp, dev), T(ff, dev), T(bb, dev))
    assert np.array_equal(gp2.cpu().numpy(), p2)


def _voxel_scene(rng, B, Z, Y, X, density):
    occ = rng.random((B, Z, Y, X)) < density
    idx = np.argwhere(occ).astype(np.int32)                      # frames contiguous, sorted
    xyz = ((idx[:, [3, 2, 1]] + 0.5) * np.array([0.1, 0.1, 0.2])).astype(np.float32)
    cnt = np.bincount(idx[:, 0], minlength=B).astype(np.int32)
    return idx, xyz, cnt


def test_voxel_query_dense_and_index(dev):
    rng = np.random.default_rng(7)
    B, Z, Y, X = 2, 11, 40, 36
    idx, xyz, cnt = _voxel_scene(rng, B, Z, Y, X, 0.08)
    v2p = oracle.generate_voxel2pinds(idx, B, [Z, Y, X])
    M = 600
    qc = np.stack([np.repeat(np.arange(B), M // B), rng.integers(-1, Z + 1, M), rng.integers(0, Y, M),
                   rng.integers(0, X, M)], 1).astype(np.int32)
    qc[:, 1] = np.clip(qc[:, 1], 0, Z - 1)
    q = ((qc[:, [3, 2, 1]] + rng.random((M, 3))) * np.array([0.1, 0.1, 0.2])).astype(np.float32)
    # x windows of 9 / 5 / 3 cells (row-wise bitmap scan), 31 cells (widest row-wise case, rows crossing
    # bitmap words and the grid border) and 35 cells (per-cell scan)
    for rng_, radius, ns in [((4, 4, 4), 0.4, 16), ((1, 2, 2), 0.25, 4), ((0, 1, 1), 0.05, 8),
                             ((1, 1, 15), 0.9, 12), ((1, 0, 17), 1.2, 20)]:
        ref, ref_e = oracle.voxel_query(rng_, radius, ns, xyz, q, qc, v2p)
        got, got_e = voxel_query_utils.voxel_query(rng_, radius, ns, T(xyz, dev), T(q, dev), T(qc, dev), T(v2p, dev))
        assert np.array_equal(got.cpu().numpy(), ref) and np.array_equal(got_e.cpu().numpy(), ref_e)
        # same answer from the sparse tensor's own index (rows in shuffled order)
        perm = rng.permutation(len(idx))
        inv = np.argsort(perm)
        st = sp.SparseConvTensor(torch.zeros(len(idx), 1, device=dev), T(idx[perm], dev), [Z, Y, X], B)
        got2, e2 = voxel_query_utils.voxel_query_sparse(rng_, radius, ns, T(xyz[perm], dev), T(q, dev), T(qc, dev), st)
        assert np.array_equal(e2.cpu().numpy(), ref_e)
        assert np.array_equal(perm[got2.cpu().numpy()][~ref_e], ref[~ref_e])
        assert inv is not None
    assert ref_e.any() and (~ref_e).any()


def test_ball_query_and_grouping(dev):
    rng = np.random.default_rng(8)
    cnt = np.array([700, 0, 500], np.int32)                      # a frame without points
    xyz = rng.uniform(0, 4, (cnt.sum(), 3)).astype(np.float32)
    ncnt = np.array([60, 0, 40], np.int32)
    q = rng.uniform(-0.5, 4.5, (ncnt.sum(), 3)).astype(np.float32)
    feat = rng.normal(size=(cnt.sum(), 32)).astype(np.float32)
    idx, empty = oracle.ball_query(0.45, 16, xyz, cnt, q, ncnt)
    gi, ge = pointnet2_utils.ball_query(0.45, 16, T(xyz, dev), T(cnt, dev), T(q, dev), T(ncnt, dev))
    assert np.array_equal(gi.cpu().numpy(), idx) and np.array_equal(ge.cpu().numpy(), empty)
    assert empty.any() and not empty.all()
    ref = oracle.group_points(feat, cnt, idx, ncnt)
    f = T(feat, dev).requires_grad_(True)
    out = pointnet2_utils.grouping_operation(f, T(cnt, dev), gi, T(ncnt, dev))
    assert np.array_equal(out.detach().cpu().numpy(), ref)
    g = rng.normal(size=ref.shape).astype(np.float32)
    out.backward(T(g, dev))
    # row 0 of a frame collects every slot of every empty ball (hundreds of terms, |sum of |terms|| ~ 1e3):
    # fp32 accumulation order (float atomics) moves its entries by ~1e-5 absolute
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.group_points_grad(g, idx, ncnt, cnt, cnt.sum()),
                               rtol=1e-5, atol=2e-4)
    # QueryAndGroup module: relative xyz + features, empty balls zeroed
    qg = pointnet2_utils.QueryAndGroup(0.45, 16, use_xyz=True)
    nf, _ = qg(T(xyz, dev), T(cnt, dev), T(q, dev), T(ncnt, dev), T(feat, dev))
    assert nf.shape == (100, 35, 16)
    assert float(nf[T(empty, dev)].abs().max()) == 0.0


def test_roi_grid_pool_module_dense_equals_sparse(dev):
    """NeighborVoxelSAModuleMSG: identical output from the dense-map path and the index path."""
    rng = np.random.default_rng(9)
    B, Z, Y, X = 2, 5, 24, 20
    idx, xyz, cnt = _voxel_scene(rng, B, Z, Y, X, 0.15)
    feats = rng.normal(size=(len(idx), 16)).astype(np.float32)
    M = 2 * 54
    qc_xyz = np.stack([np.repeat(np.arange(B), M // B), rng.integers(0, X, M), rng.integers(0, Y, M),
                       rng.integers(0, Z, M)], 1).astype(np.int32)      # [b, x, y, z] as the head passes it
    q = ((qc_xyz[:, 1:4] + rng.random((M, 3))) * np.array([0.1, 0.1, 0.2])).astype(np.float32)
    torch.manual_seed(0)
    mod = voxel_pool_modules.NeighborVoxelSAModuleMSG(query_ranges=[[2, 2, 2]], radii=[0.4], nsamples=[8],
                                                      mlps=[[16, 16, 24]]).to(dev).eval()
    st = sp.SparseConvTensor(T(feats, dev), T(idx, dev), [Z, Y, X], B)
    v2p = T(oracle.generate_voxel2pinds(idx, B, [Z, Y, X]), dev)
    args = (T(xyz, dev), T(cnt, dev), T(q, dev), torch.tensor([M // B] * B, dtype=torch.int32, device=dev),
            T(qc_xyz, dev), T(feats, dev))
    with torch.no_grad():
        a = mod(*args, v2p)
        b = mod(*args, st)
    assert a.shape == (M, 24) and torch.equal(a, b)


@pytest.mark.parametrize("mlps", [[[16, 32, 32]], [[16, 48, 64], [16, 16, 24]]])
def test_roi_grid_pool_fused_aggregation_matches_module_path(dev, mlps):
    """Inference fast path (k_voxel_pool_agg: grouping + position MLP + ReLU + max + output MLP in
    one kernel, BatchNorms folded) vs the module's own tensor-op path that follows
    voxel_pool_modules.py:88-108; empty balls present; tolerance 1e-5 relative (folded affine)."""
    rng = np.random.default_rng(31)
    B, Z, Y, X = 2, 5, 24, 20
    idx, xyz, cnt = _voxel_scene(rng, B, Z, Y, X, 0.08)
    feats = rng.normal(size=(len(idx), 16)).astype(np.float32)
    M = 2 * 301
    qc_xyz = np.stack([np.repeat(np.arange(B), M // B), rng.integers(0, X, M), rng.integers(0, Y, M),
                       rng.integers(0, Z, M)], 1).astype(np.int32)
    q = ((qc_xyz[:, 1:4] + rng.random((M, 3))) * np.array([0.1, 0.1, 0.2])).astype(np.float32)
    torch.manual_seed(3)
    n = len(mlps)
    mod = voxel_pool_modules.NeighborVoxelSAModuleMSG(
        query_ranges=[[1, 1, 1], [2, 2, 2]][:n], radii=[0.25, 0.4][:n], nsamples=[16, 8][:n],
        mlps=mlps).to(dev)
    for m in mod.modules():                       # non-trivial running statistics and affine
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    mod.eval()
    st = sp.SparseConvTensor(T(feats, dev), T(idx, dev), [Z, Y, X], B)
    args = (T(xyz, dev), T(cnt, dev), T(q, dev), torch.tensor([M // B] * B, dtype=torch.int32, device=dev),
            T(qc_xyz, dev), T(feats, dev))
    with torch.no_grad():
        assert mod._fusable(args[-1])
        fused = mod(*args, st)
        v2p = T(oracle.generate_voxel2pinds(idx, B, [Z, Y, X]), dev)
        fused_dense = mod(*args, v2p)
        mod.USE_FUSED = False
        plain = mod(*args, st)
        _, _, empty = mod.groupers[0](args[4][:, [0, 3, 2, 1]].contiguous(), args[0], args[1], args[2],
                                      args[3], T(feats, dev), st)
    assert bool(empty.any()) and not bool(empty.all())
    assert fused.shape == plain.shape == (M, sum(m[-1] for m in mlps))
    assert torch.equal(fused, fused_dense)
    np.testing.assert_allclose(fused.cpu().numpy(), plain.cpu().numpy(), rtol=1e-5, atol=1e-5)
    with torch.enable_grad():                     # training / autograd keeps the tensor-op path
        mod.USE_FUSED = True
        assert not mod._fusable(args[-1])


def test_roi_grid_pool_harness_vs_reference_formulation(dev):
    """REGRESSION test (the parity of this stage against the reference's OWN VoxelRCNNHead.roi_grid_pool /
    NeighborVoxelSAModuleMSG, executed, is tests/test_reference_step_gpu.py: pooled features of a training step and an
    inference pass): glenet_amd.roi_grid.RoIGridPool == that data flow re-typed here with its own pieces -- dense
    voxel->point map per scale, grid coords by float floor division, per-batch counts (voxelrcnn_head.py:106-191) -- on
    two scales of a synthetic scene with other shapes than the golden's."""
    from glenet_amd import roi_grid as rg
    rng = np.random.default_rng(21)
    B, vs, pcr = 2, [0.1, 0.1, 0.2], [0.0, 0.0, 0.0, 4.0, 4.8, 2.0]
    shapes = {"x_conv1": (10, 48, 40), "x_conv2": (5, 24, 20)}
    strides = {"x_conv1": 1, "x_conv2": 2}
    feats, tensors = {}, {}
    for name, (Z, Y, X) in shapes.items():
        idx, _, _ = _voxel_scene(rng, B, Z, Y, X, 0.12)
        f = rng.normal(size=(len(idx), 8)).astype(np.float32)
        feats[name] = (idx, f)
        tensors[name] = sp.SparseConvTensor(T(f, dev), T(idx, dev), [Z, Y, X], B)
    cfg = {n: dict(mlps=[[8, 12]], query_ranges=[[2, 2, 2]], radii=[0.3 * strides[n]], nsamples=[8])
           for n in shapes}
    torch.manual_seed(4)
    pool = rg.RoIGridPool({n: 8 for n in shapes}, cfg, 3, vs, pcr).to(dev).eval()
    rois = np.concatenate([rng.uniform([0.5, 0.5, 0.4], [3.5, 4.3, 1.6], (B, 5, 3)),
                           rng.uniform(0.4, 1.2, (B, 5, 3)), rng.uniform(-3, 3, (B, 5, 1))], -1).astype(np.float32)
    with torch.no_grad():
        fused = pool(T(rois, dev), tensors, strides, B)          # inference fast path (3 kernels / scale)
        pool.USE_FUSED = False
        for layer in pool.roi_grid_pool_layers:
            layer.USE_FUSED = False
        got = pool(T(rois, dev), tensors, strides, B)
        # reference formulation
        grid, _ = rg.global_grid_points_of_roi(T(rois, dev), 3)
        grid = grid.view(B, -1, 3)
        gc = torch.cat([(grid[..., i:i + 1] - pcr[i]) // vs[i] for i in range(3)], -1)
        bidx = torch.zeros(B, gc.shape[1], 1, device=dev)
        for b in range(B):
            bidx[b] = b
        cnt_new = torch.full((B,), gc.shape[1], dtype=torch.int32, device=dev)
        want = []
        for k, name in enumerate(shapes):
            idx, f = feats[name]
            st = tensors[name]
            xyz = rg.get_voxel_centers(st.indices[:, 1:4], strides[name], vs, pcr)
            cnt = torch.tensor([(idx[:, 0] == b).sum() for b in range(B)], dtype=torch.int32, device=dev)
            v2p = T(oracle.generate_voxel2pinds(idx, B, list(shapes[name])), dev)
            coords = torch.cat([bidx, gc // strides[name]], -1).int()
            o = pool.roi_grid_pool_layers[k](xyz=xyz.contiguous(), xyz_batch_cnt=cnt,
                                             new_xyz=grid.reshape(-1, 3).contiguous(), new_xyz_batch_cnt=cnt_new,
                                             new_coords=coords.view(-1, 4).contiguous(), features=st.features,
                                             voxel2point_indices=v2p)
            want.append(o.view(-1, 27, o.shape[-1]))
        want = torch.cat(want, -1)
    assert got.shape == (B * 5, 27, 24) and torch.equal(got, want)
    assert float(got.abs().max()) > 0
    # fast path: same arithmetic with folded BatchNorms; sin/cos may differ by an ulp
    np.testing.assert_allclose(fused.cpu().numpy(), got.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_roi_grid_points_and_floor_division_coords(dev):
    """glx_roi_grid_points vs get_global_grid_points_of_roi restated in torch (<= 2 ulp: sin/cos)
    and its voxel coordinates vs torch's own float floor division `//` on the SAME points, exactly
    (voxelrcnn_head.py:128-134).  Zero-size RoIs centred on multiples of the voxel size put every
    grid point on a cell boundary, where floor(a / b) and a // b part ways."""
    import ctypes
    from glenet_amd import _lib, roi_grid as rg
    rng = np.random.default_rng(77)
    vs, pcr = [0.05, 0.05, 0.1], [0.0, -40.0, -3.0]
    G, B, R = 6, 2, 300
    rois = np.concatenate([rng.uniform([0, -40, -3], [70.4, 40, 1], (B * R, 3)),
                           rng.uniform(0.3, 5.0, (B * R, 3)), rng.uniform(-4, 4, (B * R, 1))], -1).astype(np.float32)
    k = rng.integers(0, 1400, (200, 3))
    rois[:200, 0:3] = (k * np.array(vs) + np.array(pcr)).astype(np.float32)      # on cell boundaries
    rois[:200, 3:6] = 0
    rois[200:260, 0] = rng.uniform(-3, 0, 60)                                    # outside the range
    r = T(rois, dev)
    f3 = ctypes.c_float * 3
    xyz = torch.empty((B * R * G ** 3, 3), dtype=torch.float32, device=dev)
    coords = torch.empty((B * R * G ** 3, 4), dtype=torch.int32, device=dev)
    _lib.call("glx_roi_grid_points", r, B * R, 7, R, G, f3(*pcr), f3(*vs), xyz, coords)
    want_xyz, _ = rg.global_grid_points_of_roi(r, G)
    np.testing.assert_allclose(xyz.cpu().numpy(), want_xyz.reshape(-1, 3).cpu().numpy(), rtol=0, atol=2e-5)
    assert torch.equal(xyz[:200 * G ** 3], want_xyz[:200].reshape(-1, 3))      # no rotation involved
    want_c = torch.stack([(xyz[:, i] - pcr[i]) // vs[i] for i in range(3)], -1).int()
    assert torch.equal(coords[:, [3, 2, 1]], want_c)
    assert torch.equal(coords[:, 0], torch.arange(B * R * G ** 3, device=dev).int() // (R * G ** 3))
    assert int((want_c < 0).sum()) > 0
    # at every stride the in-kernel integer floor division == torch's `//` on the float coords
    for stride in (2, 4, 8):
        a = (want_c.float() // stride).int()
        b = torch.div(want_c, stride, rounding_mode="floor")
        assert torch.equal(a, b)


def test_farthest_point_sampling_bit_identical(dev):
    """Stacked + batched FPS vs the oracle (same tie rule as the reference's 1024-thread tree);
    duplicate points force ties; one frame > 16 K points takes the global-temp path."""
    rng = np.random.default_rng(13)
    cnt = [5000, 700, 17000]
    xyz = rng.normal(size=(sum(cnt), 3)).astype(np.float32)
    xyz[5000 + 10:5000 + 200] = xyz[5000 + 5]                      # exact duplicates in frame 1
    xyz[100:2100] = np.round(xyz[100:2100], 1)                      # many equal distances in frame 0
    npoint = [256, 300, 128]
    want = oracle.stack_farthest_point_sample(xyz, cnt, npoint)
    got = pointnet2_utils.stack_farthest_point_sample(T(xyz, dev), torch.tensor(cnt, dtype=torch.int32, device=dev),
                                                      npoint).cpu().numpy()
    assert np.array_equal(got, want)
    small = [5000, 700, 6000]
    xs = xyz[:sum(small)]
    want = oracle.stack_farthest_point_sample(xs, small, 64)
    got = pointnet2_utils.stack_farthest_point_sample(T(xs, dev), torch.tensor(small, dtype=torch.int32, device=dev), 64)
    assert np.array_equal(got.cpu().numpy(), want)
    xb = rng.normal(size=(3, 2048, 3)).astype(np.float32)
    gotb = pointnet2_utils.farthest_point_sample(T(xb, dev), 100).cpu().numpy()
    for b in range(3):
        assert np.array_equal(gotb[b], oracle.stack_farthest_point_sample(xb[b], [2048], 100))


def test_three_nn_and_interpolate(dev):
    rng = np.random.default_rng(14)
    ucnt, kcnt = [700, 1300], [900, 2500]
    unknown = rng.uniform(-5, 5, (sum(ucnt), 3)).astype(np.float32)
    known = rng.uniform(-5, 5, (sum(kcnt), 3)).astype(np.float32)
    known[5] = known[6] = known[7] = known[8]                      # equal distances: ascending index wins
    feats = rng.normal(size=(sum(kcnt), 24)).astype(np.float32)
    d_ref, i_ref = oracle.three_nn(unknown, ucnt, known, kcnt)
    d, i = pointnet2_utils.three_nn(T(unknown, dev), torch.tensor(ucnt, dtype=torch.int32, device=dev),
                                    T(known, dev), torch.tensor(kcnt, dtype=torch.int32, device=dev))
    assert np.array_equal(i.cpu().numpy(), i_ref)
    np.testing.assert_array_equal(d.cpu().numpy(), d_ref)
    w = 1.0 / (d_ref + 1e-8)
    w = (w / w.sum(1, keepdims=True)).astype(np.float32)
    f = T(feats, dev).requires_grad_(True)
    out = pointnet2_utils.three_interpolate(f, i, T(w, dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), oracle.three_interpolate(feats, i_ref, w),
                               rtol=1e-6, atol=1e-6)
    g = rng.normal(size=out.shape).astype(np.float32)
    out.backward(T(g, dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.three_interpolate_grad(g, i_ref, w, len(feats)),
                               rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("cin,P,B,widths", [(4, 512, 9, (64, 128, 512)), (5, 200, 3, (64, 128, 512)),
                                              (4, 128, 1, (64, 128, 512)), (4, 512, 7, (8, 8, 8)),
                                              (5, 300, 2, (16, 8, 12))])
@pytest.mark.parametrize("f16x2", [True, False])
def test_fused_pointnet_feat_matches_torch_modules(dev, cin, P, B, widths, f16x2, monkeypatch):
    """glx_pointnet_feat / glx_pointnet_feat_f16x2 (one MFMA kernel) == the unfused Conv1d/BatchNorm1d/ReLU/max modules in
    fp32, also when P is not a multiple of the 128-point pass and for 5 point features."""
    from glenet_amd import dense_path as dp
    if not f16x2 and widths[2] != 512:
        pytest.skip("the narrow extractor has one form")
    monkeypatch.setattr(dp.PointFeat, "F16X2", f16x2)
    torch.manual_seed(cin * 100 + P)
    m = dp.PointFeat(cin, widths).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    for bn in (m.bn1, m.bn2, m.bn3):
        bn.running_mean.copy_((torch.randn(bn.num_features, generator=g) * 0.2).to(dev))
        bn.running_var.copy_((torch.rand(bn.num_features, generator=g) + 0.5).to(dev))
        bn.weight.data.copy_((torch.rand(bn.num_features, generator=g) - 0.3).to(dev))   # some negative scales
        bn.bias.data.copy_((torch.randn(bn.num_features, generator=g) * 0.1).to(dev))
    x = torch.randn(B, cin, P, device=dev)
    with torch.no_grad():
        fused = m(x)
        z = torch.relu(m.bn1(m.conv1(x)))
        z = torch.relu(m.bn2(m.conv2(z)))
        ref = m.bn3(m.conv3(z)).max(dim=2)[0]
    assert fused.shape == (B, widths[2])
    np.testing.assert_allclose(fused.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    # the module falls back to the torch path when gradients are needed
    assert m(x.requires_grad_(True)).requires_grad


def test_pointnet_feat_f16x2_against_fp64_with_wild_scales(dev):
    """The f16 x 2 form of the wide extractor carries a power of two per weight row and per point: output channels whose
    folded weights are 2^-12 .. 2^12 apart, an all-zero weight row, points far from the origin next to points at it and
    objects whose second layer is dead (every activation 0) all come out within 2^-17 of the output's scale of the
    fp64 result -- the fp32-MFMA form is held to the same bound beside it."""
    from glenet_amd import dense_path as dp
    torch.manual_seed(5)
    B, cin, P = 6, 4, 333
    m = dp.PointFeat(cin, (64, 128, 512)).to(dev).eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for bn in (m.bn1, m.bn2, m.bn3):
            bn.running_mean.copy_((torch.randn(bn.num_features, generator=g) * 0.2).to(dev))
            bn.running_var.copy_((torch.rand(bn.num_features, generator=g) + 0.5).to(dev))
            bn.weight.copy_((torch.rand(bn.num_features, generator=g) - 0.3).to(dev))
            bn.bias.copy_((torch.randn(bn.num_features, generator=g) * 0.1).to(dev))
        m.conv2.weight.mul_(torch.exp2(torch.randint(-12, 13, (128, 1, 1), generator=g).float()).to(dev))
        m.conv3.weight.mul_(torch.exp2(torch.randint(-12, 13, (512, 1, 1), generator=g).float()).to(dev))
        m.conv2.weight[9].zero_()
        m.conv3.weight[77].zero_()
    x = torch.randn(B, cin, P, device=dev)
    x[1] *= 300.0                                   # a far object
    x[2, :, ::2] = 0.0                              # points at the origin between the others
    x[3] *= 1e-6
    md = copy.deepcopy(m).double()
    with torch.no_grad():
        xd = x.double()
        z = torch.relu(md.bn1(md.conv1(xd)))
        z = torch.relu(md.bn2(md.conv2(z)))
        y = md.bn3(md.conv3(z))
        ref = y.max(dim=2)[0]
        scale = y.abs().amax(dim=2)                 # per object and channel: what the sums were made of
        # the contraction's own scale: |W3| |h2| per object and channel, at its largest over the points
        w3 = (md.conv3.weight[:, :, 0] * (md.bn3.weight / torch.sqrt(md.bn3.running_var + md.bn3.eps))[:, None]).abs()
        mag = torch.einsum("oc,bcp->bop", w3, z.abs()).amax(dim=2) + scale
    for f16x2 in (True, False):
        dp.PointFeat.F16X2 = f16x2
        try:
            with torch.no_grad():
                got = m(x).double()
        finally:
            dp.PointFeat.F16X2 = True
        err = ((got - ref).abs() / mag.clamp_min(1e-30)).max().item()
        assert err < 2.0 ** -17, (f16x2, err)


@pytest.mark.parametrize("cout,cin", [(128, 64), (512, 128), (64, 128), (128, 128)])
def test_f16x2_pack_kernel_equals_the_tensor_statements(dev, cout, cin):
    """glx_f16x2_pack (one launch) == the image dense_path.PointFeat._f16x2_image builds with tensor statements on the host, bit for
    bit: rows 2^-20 .. 2^20 apart, a zero row, an exact power of two as a row's maximum; a transposed (strided) source, a per-row
    factor and a constant."""
    from glenet_amd import dense_path as dp
    g = torch.Generator().manual_seed(cout + cin)
    w = torch.randn(cout, cin, generator=g) * torch.exp2(torch.randint(-20, 21, (cout, 1), generator=g).float())
    w[3] = 0
    w[5] = 0.25 * torch.sign(w[5])
    for src, rs, sc in ((w, None, 1.0), (w.t().contiguous().t(), torch.where(torch.rand(cout, generator=g) > 0.5, 1.0, -1.0), 1.0),
                        (w, None, -1.0)):
        img_h, ew_h = dp.PointFeat._f16x2_image(src, rs, sc)                       # host: tensor statements
        img_d, ew_d = dp.PointFeat._f16x2_image(src.to(dev) if src.is_contiguous() else src.t().contiguous().to(dev).t(),
                                                None if rs is None else rs.to(dev), sc)
        assert img_d.shape == img_h.shape and img_d.dtype == torch.float16
        assert torch.equal(ew_d.cpu(), ew_h)
        assert torch.equal(img_d.cpu().view(torch.int16), img_h.view(torch.int16))


@pytest.mark.parametrize("rows", [70001, 31, 8192 * 256])
def test_rows128_moments_against_fp64(dev, rows):
    """glx_rows128_moments: x^T x (fp64 out) and the column sums of a (rows, 128) matrix in one pass, bf16 x 3 products: against
    fp64 products to 5e-6 of sqrt(G_ii G_jj) (fp32 sums over a block's <= 8192 rows of non-negative values: 2.3e-6 measured at
    2.1 M rows), symmetric, twice the same bits; row counts that are
    not a multiple of the 32-row step, fewer rows than one step, and the CVAE's 2.1 M."""
    import ctypes
    from glenet_amd import _lib
    g = torch.Generator(device=dev).manual_seed(rows % 1000)
    x = torch.relu(torch.randn(rows, 128, device=dev, generator=g) + 0.3) * torch.exp2(torch.randint(-6, 7, (1, 128), device=dev, generator=g).float())
    n = _lib.query("glx_rows128_moments_workspace_bytes")
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    outs = []
    for _ in range(2):
        G = torch.full((128, 128), float("nan"), dtype=torch.float64, device=dev)
        H = torch.full((128,), float("nan"), device=dev)
        _lib.call("glx_rows128_moments", x, ctypes.c_longlong(rows), G, H, None, ws, _lib.size_arg(n))
        outs.append((G, H))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    G, H = outs[0]
    want, hw = torch.zeros(128, 128, dtype=torch.float64, device=dev), torch.zeros(128, dtype=torch.float64, device=dev)
    for lo in range(0, rows, 1 << 18):                     # fp64 in chunks (memory)
        xd = x[lo:lo + (1 << 18)].double()
        want += xd.t() @ xd
        hw += xd.sum(0)
    dg = want.diagonal().sqrt()
    assert torch.equal(G, G.t())
    assert float(((G - want).abs() / (dg[:, None] * dg[None, :]).clamp_min(1e-300)).max()) < 5e-6
    assert float(((H.double() - hw).abs() / hw.abs().clamp_min(1e-30)).max()) < 5e-6


def test_rows128_affine_f16x2_against_fp64(dev):
    """glx_rows128_affine_f16x2 (the dense part of the 128 -> 512 layer's input gradient): y = init + x W^T on a row count that
    is not a multiple of the 32-row trip, weight rows 2^-10 .. 2^10 apart, rows of very different magnitudes and an all-zero row:
    within 2^-17 of the contraction's own scale per element."""
    import ctypes
    from glenet_amd import _lib, dense_path as dp
    g = torch.Generator(device=dev).manual_seed(21)
    rows = 5003
    x = torch.randn(rows, 128, device=dev, generator=g) * torch.exp2(torch.randint(-8, 9, (rows, 1), device=dev, generator=g).float())
    x[17] = 0
    w = torch.randn(128, 128, device=dev, generator=g) * torch.exp2(torch.randint(-10, 11, (128, 1), device=dev, generator=g).float())
    init = torch.randn(128, device=dev, generator=g)
    wh, ew = dp.PointFeat._f16x2_image(w)
    y = torch.full((rows, 128), float("nan"), device=dev)
    _lib.call("glx_rows128_affine_f16x2", x, ctypes.c_longlong(rows), wh, ew, init, y, None)
    want = init.double() + x.double() @ w.double().t()
    mag = x.abs().double() @ w.abs().double().t() + init.abs().double()
    assert torch.isfinite(y).all()
    assert float(((y.double() - want).abs() / mag.clamp_min(1e-300)).max()) < 2.0 ** -17
    assert torch.equal(y[17], init)                                       # the zero row: exactly init
    y2 = torch.empty_like(y)
    _lib.call("glx_rows128_affine_f16x2", x, ctypes.c_longlong(rows), wh, ew, None, y2, None)      # init == NULL
    assert float(((y2.double() - (want - init.double())).abs() / mag).max()) < 2.0 ** -17


@pytest.mark.parametrize("cin,P,B", [(4, 17, 3), (4, 64, 2), (3, 65, 5), (8, 129, 1), (4, 512, 300), (5, 191, 259)])
def test_pointnet_feat_f16x2_register_resident_form_against_the_streamed_one(dev, cin, P, B):
    """The two kernels behind glx_pointnet_feat_f16x2(_pair) -- W3 in registers with the points through LDS (the default) and W3
    streamed through an LDS ring -- on objects of one, two and many half-passes of 64 points (P = 17: one, mostly padding; 65: the
    second holds one point), fewer objects than CUs and more (a block walks over several objects, 300 and 259 are not multiples of
    anything), 3 .. 8 point features (8: two k-steps in layer 1), with and without the narrow extractor riding along: against the
    modules in fp64 to 2^-17 of the contraction's scale, and against each other to the rounding of layer 1 (fp32 MFMA sums in one form,
    an FMA chain in the other)."""
    from glenet_amd import _lib, dense_path as dp
    torch.manual_seed(cin * 1000 + P)
    m = dp.CVAE(cin, 8).to(dev).eval()
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_((torch.randn(mod.num_features, generator=g) * 0.2).to(dev))
                mod.running_var.copy_((torch.rand(mod.num_features, generator=g) + 0.5).to(dev))
                mod.weight.copy_((torch.rand(mod.num_features, generator=g) - 0.3).to(dev))
                mod.bias.copy_((torch.randn(mod.num_features, generator=g) * 0.1).to(dev))
        fe = m.x_encoder.fe
        pts = torch.randn(B, cin, P, device=dev)
        pts[0] *= 50.0
        w1, b1, _, b2, _, b3 = fe._packed()
        w2h, e2, w3h, e3 = fe._packed_f16()
        narrow, _ = m._sample_pack()
        out = {}
        lib = _lib.load()
        for form in (1, 0):
            before = lib.glx_pointnet_feat_set_form(form)
            try:
                f512, f8 = torch.full((B, 512), 7.0, device=dev), torch.full((B, 8), 7.0, device=dev)
                _lib.call("glx_pointnet_feat_f16x2_pair", pts, B, cin, P, w1, b1, w2h, e2, b2, w3h, e3, b3, f512, narrow, f8)
                alone = torch.full((B, 512), 7.0, device=dev)
                _lib.call("glx_pointnet_feat_f16x2", pts, B, cin, P, w1, b1, w2h, e2, b2, w3h, e3, b3, alone)
                torch.cuda.synchronize()
            finally:
                lib.glx_pointnet_feat_set_form(before)
            assert torch.equal(alone, f512), form                  # the narrow extractor does not disturb the wide one
            out[form] = (f512.double(), f8.double())
        assert before == 1                                          # the register-resident form is the default
        fd = copy.deepcopy(fe).double()
        xd = pts.double()
        z = torch.relu(fd.bn1(fd.conv1(xd)))
        z = torch.relu(fd.bn2(fd.conv2(z)))
        y = fd.bn3(fd.conv3(z))
        ref = y.max(dim=2)[0]
        w3 = (fd.conv3.weight[:, :, 0] * (fd.bn3.weight / torch.sqrt(fd.bn3.running_var + fd.bn3.eps))[:, None]).abs()
        mag = torch.einsum("oc,bcp->bop", w3, z.abs()).amax(dim=2) + y.abs().amax(dim=2)
        nd = copy.deepcopy(m.obj_encoder.fe).double()
        zn = torch.relu(nd.bn1(nd.conv1(xd)))
        zn = torch.relu(nd.bn2(nd.conv2(zn)))
        ref8 = nd.bn3(nd.conv3(zn)).max(dim=2)[0]
    for form in (1, 0):
        err = ((out[form][0] - ref).abs() / mag.clamp_min(1e-30)).max().item()
        assert err < 2.0 ** -17, (form, err)
        np.testing.assert_allclose(out[form][1].cpu().numpy(), ref8.cpu().numpy(), rtol=2e-5, atol=2e-5)
    between = ((out[1][0] - out[0][0]).abs() / mag.clamp_min(1e-30)).max().item()
    assert between < 2.0 ** -19, between


@pytest.mark.parametrize("P,B,pre", [(17, 3, False), (64, 2, True), (65, 5, True), (512, 300, True), (191, 259, False)])
def test_pointmax_forward_f16x2_register_resident_form_equals_the_streamed_one(dev, P, B, pre):
    """The two kernels behind glx_pointmax_forward_f16x2 (W3 in registers with the rows through LDS -- the default --, and W3 streamed
    through an LDS ring): the same maxima AND the same points bit for bit, on one-, two- and many-half-pass objects, fewer and more
    objects than CUs, with and without the BatchNorm of the layer in front applied on load, duplicated rows (ties go to the lower
    point) and a constant object; the maxima against an fp64 product to 2^-17 of the contraction's scale."""
    from glenet_amd import _lib, dense_path as dp
    torch.manual_seed(P * 7 + B)
    h2 = torch.randn(B * P, 128, device=dev)
    h2[: P] = h2[0]                                    # object 0: every row the same (all ties)
    if B > 1 and P > 3:
        h2[P + 3] = h2[P + 1]                          # object 1: a duplicated row
    w = torch.randn(512, 128, device=dev) * torch.exp2(torch.randint(-6, 7, (512, 1), device=dev).float())
    w3h, e3 = dp.PointFeat._f16x2_image(w)
    coef = torch.cat([torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.3]) if pre else None
    lib = _lib.load()
    out = {}
    for form in (1, 0):
        before = lib.glx_pointnet_feat_set_form(form)
        try:
            v = torch.full((B, 512), 3.0, device=dev)
            a = torch.full((B, 512), -1, device=dev, dtype=torch.int32)
            _lib.call("glx_pointmax_forward_f16x2", h2, B, P, w3h, e3, v, a, coef)
            torch.cuda.synchronize()
        finally:
            lib.glx_pointnet_feat_set_form(before)
        out[form] = (v, a)
    assert torch.equal(out[1][0], out[0][0]) and torch.equal(out[1][1], out[0][1])
    v, a = out[1]
    hd = h2.double()
    if pre:
        hd = torch.relu(hd * coef[:128].double() + coef[128:].double())
    y = (hd @ w.double().t()).view(B, P, 512)
    ref = y.max(dim=1)[0]
    mag = (hd.abs() @ w.double().abs().t()).view(B, P, 512).amax(dim=1)
    assert float(((v.double() - ref).abs() / mag.clamp_min(1e-30)).max()) < 2.0 ** -17
    assert int(a.min()) >= 0 and int(a.max()) < P
    picked = torch.gather(y, 1, a.long()[:, None, :])[:, 0]          # the value at the reported point is the maximum (to the same bound)
    assert float(((picked - ref).abs() / mag.clamp_min(1e-30)).max()) < 2.0 ** -16
    assert int(a[0].max()) == 0                                      # all rows equal: the lowest point


@pytest.mark.parametrize("bins,cin", [(2, 4), (3, 5)])
def test_cvae_two_launch_sampler_equals_the_modules(dev, bins, cin):
    """CVAE.sample's fused path (glx_pointnet_feat_f16x2_pair + glx_cvae_sample_tail) against the same model module by module
    (eval-mode BatchNorm with non-trivial running statistics, 2 and 3 direction bins, 4 and 5 point features, an object count
    that is not a multiple of the tail kernel's 16 objects per block): boxes to 1e-4, the heading modulo its bin period; and the
    narrow extractor riding along in the wide one's launch against its own kernel."""
    from glenet_amd import _lib, dense_path as dp
    torch.manual_seed(bins * 10 + cin)
    m = dp.CVAE(cin, 8, num_dir_bins=bins).to(dev).eval()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_((torch.randn(mod.num_features, generator=g) * 0.2).to(dev))
                mod.running_var.copy_((torch.rand(mod.num_features, generator=g) + 0.5).to(dev))
                mod.weight.copy_((torch.rand(mod.num_features, generator=g) + 0.5).to(dev))
                mod.bias.copy_((torch.randn(mod.num_features, generator=g) * 0.1).to(dev))
        B, P = 37, 300
        pts = torch.randn(B, cin, P, device=dev)
        eps = torch.randn(B, 8, device=dev)
        assert m._sample_fusable(pts)
        got = m.sample(pts, eps)
        dp.CVAE.FUSED_SAMPLER = False
        try:
            want = m.sample(pts, eps)
        finally:
            dp.CVAE.FUSED_SAMPLER = True
        f8_small = m.obj_encoder.fe(pts)                                   # glx_pointnet_feat_small
        fe = m.x_encoder.fe
        w1, b1, _, b2, _, b3 = fe._packed()
        w2h, e2, w3h, e3 = fe._packed_f16()
        narrow, _ = m._sample_pack()
        f512, f8 = torch.empty(B, 512, device=dev), torch.empty(B, 8, device=dev)
        _lib.call("glx_pointnet_feat_f16x2_pair", pts.contiguous(), B, cin, P, w1, b1, w2h, e2, b2, w3h, e3, b3, f512, narrow, f8)
        f512_alone = fe(pts)                                               # glx_pointnet_feat_f16x2
    assert got.shape == (B, 7 + bins)
    d = (got - want).abs()
    period = 2 * np.pi / bins
    d[:, 6] = torch.minimum(d[:, 6], (d[:, 6] - period).abs())
    assert float(d.max()) < 1e-4, float(d.max())
    np.testing.assert_allclose(f8.cpu().numpy(), f8_small.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f512.cpu().numpy(), f512_alone.cpu().numpy(), rtol=0, atol=0)     # the same kernel body: bitwise


def test_group_points_gather_backward_matches_atomic_scatter(dev):
    """The gather form of the grouping gradient (no float atomics; chosen when many grid points share
    few rows) == the oracle's scatter, incl. rows nobody references (0), a frame without queries and
    channel counts that are not a power of two."""
    rng = np.random.default_rng(17)
    for C in (32, 3, 48):
        cnt = np.array([300, 50, 200], np.int32)
        ncnt = np.array([4000, 0, 2500], np.int32)
        ns = 16
        idx = np.concatenate([rng.integers(0, max(int(c * 0.3), 1), (m, ns)) for c, m in zip(cnt, ncnt)]).astype(np.int32)
        g = rng.normal(size=(int(ncnt.sum()), C, ns)).astype(np.float32)
        want = oracle.group_points_grad(g, idx, ncnt, cnt, int(cnt.sum()))
        feats = T(rng.normal(size=(int(cnt.sum()), C)).astype(np.float32), dev).requires_grad_(True)
        assert int(ncnt.sum()) * ns >= pointnet2_utils.GATHER_MIN_REFS_PER_ROW * int(cnt.sum())
        out = pointnet2_utils.grouping_operation(feats, T(cnt, dev), T(idx, dev), T(ncnt, dev))
        out.backward(T(g, dev))
        np.testing.assert_allclose(feats.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * np.abs(want).max())
        assert float(feats.grad[cnt[0]:cnt[0] + cnt[1]].abs().max()) == 0.0      # the frame without queries


def test_roi_grid_pool_row_major_training_path_equals_conv_formulation(dev):
    """Training-mode NeighborVoxelSAModuleMSG: the row-major path (1x1 convs as matrix products,
    BatchNorm on rows) == the module's conv formulation that mirrors voxel_pool_modules.py:88-108 --
    outputs, input / parameter gradients and the BatchNorm running statistics."""
    import copy
    rng = np.random.default_rng(41)
    B, Z, Y, X = 2, 5, 24, 20
    idx, xyz, cnt = _voxel_scene(rng, B, Z, Y, X, 0.08)
    feats = rng.normal(size=(len(idx), 16)).astype(np.float32)
    M = 2 * 304                                   # M * 16 rows divisible by 128: the split-K product runs too
    qc_xyz = np.stack([np.repeat(np.arange(B), M // B), rng.integers(0, X, M), rng.integers(0, Y, M),
                       rng.integers(0, Z, M)], 1).astype(np.int32)
    q = ((qc_xyz[:, 1:4] + rng.random((M, 3))) * np.array([0.1, 0.1, 0.2])).astype(np.float32)
    torch.manual_seed(3)
    a = voxel_pool_modules.NeighborVoxelSAModuleMSG(query_ranges=[[1, 1, 1], [2, 2, 2]], radii=[0.25, 0.4],
                                                    nsamples=[16, 8], mlps=[[16, 32, 32], [16, 16, 24]]).to(dev).train()
    b = copy.deepcopy(a)
    b.USE_ROW_MAJOR = False
    voxel_pool_modules.NeighborVoxelSAModuleMSG.SPLITK_MIN_ROWS = 1024
    st = sp.SparseConvTensor(T(feats, dev), T(idx, dev), [Z, Y, X], B)
    outs = []
    for mod in (a, b):
        f = T(feats, dev).requires_grad_(True)
        # the conv formulation's (1, C, M, ns) 1x1 convolutions run on torch's own kernels, not MIOpen's: MIOpen's
        # backward for these shapes faulted ("Memory access fault by GPU") in full-suite runs, depending on where the
        # allocator had put the tensors -- a vendor kernel reading past a buffer, in the mirror only (the product
        # path, USE_ROW_MAJOR, has no convolution call)
        with torch.backends.cudnn.flags(enabled=mod.USE_ROW_MAJOR):
            out = mod(T(xyz, dev), T(cnt, dev), T(q, dev), torch.tensor([M // B] * B, dtype=torch.int32, device=dev),
                      T(qc_xyz, dev), f, st)
            (out * torch.linspace(0.5, 1.5, out.shape[1], device=dev)).square().mean().backward()
        outs.append((out.detach(), f.grad))
    np.testing.assert_allclose(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy(), rtol=1e-3, atol=1e-6)
    for (n, p), (_, q2) in zip(a.named_parameters(), b.named_parameters()):
        np.testing.assert_allclose(p.grad.cpu().numpy(), q2.grad.cpu().numpy(), rtol=2e-3,
                                   atol=2e-5 * max(1e-3, float(q2.grad.abs().max())), err_msg=n)
    for (n, u), (_, v) in zip(a.named_buffers(), b.named_buffers()):
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=n)
    voxel_pool_modules.NeighborVoxelSAModuleMSG.SPLITK_MIN_ROWS = 1 << 13


def test_group_rows_and_gradient_match_indexing(dev):
    """GroupRows (row-major grouping on raw query rows): out == features[idx] with empty balls zeroed
    (bit-exact: copies), gradient == index_add of the incoming rows (gather form, fp32 sum order
    differs -> rtol 1e-5).  C = 32 (feature width) and C = 3 (coordinates, non-power-of-two)."""
    rng = np.random.default_rng(5)
    for n, m, ns, c in ((700, 300, 16, 32), (50, 129, 8, 3), (4000, 64, 16, 16)):
        feats = rng.normal(size=(n, c)).astype(np.float32)
        idx = rng.integers(0, n, (m, ns)).astype(np.int32)
        empty = rng.random(m) < 0.3
        idx[empty, 0] = -1
        idx[empty, 1:] = rng.integers(-5, n, (int(empty.sum()), ns - 1))        # unspecified slots
        f = T(feats, dev).requires_grad_(True)
        out = voxel_pool_modules.GroupRows.apply(f, T(idx, dev))
        want = feats[np.where(empty[:, None], 0, idx)] * (~empty)[:, None, None]
        assert np.array_equal(out.detach().cpu().numpy(), want.astype(np.float32))
        g = rng.normal(size=(m, ns, c)).astype(np.float32)
        out.backward(T(g, dev))
        ref = np.zeros((n, c), np.float64)
        live = ~empty
        np.add.at(ref, idx[live].reshape(-1), g[live].reshape(-1, c).astype(np.float64))
        np.testing.assert_allclose(f.grad.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)


def test_group_rows_edge_cases(dev):
    """No query rows, and queries that are all empty balls: zeros out, zero gradient (no reference is
    bucketed, the gather has nothing to read)."""
    feats = torch.randn(40, 8, device=dev, requires_grad=True)
    out = voxel_pool_modules.GroupRows.apply(feats, torch.zeros((0, 4), dtype=torch.int32, device=dev))
    assert out.shape == (0, 4, 8)
    idx = torch.full((33, 4), -1, dtype=torch.int32, device=dev)
    idx[:, 1:] = 7                                                  # unspecified slots of empty balls
    out = voxel_pool_modules.GroupRows.apply(feats, idx)
    assert float(out.detach().abs().max()) == 0.0
    out.backward(torch.ones_like(out))
    assert float(feats.grad.abs().max()) == 0.0


def test_relu_add_max_matches_tensor_ops(dev):
    """ReluAddMax == relu(a + b).max(dim=1)[0], forward bit-exact, gradients of both inputs equal to
    autograd's (random data: no ties; rows that are negative everywhere give max 0 and no gradient)."""
    torch.manual_seed(2)
    for m, ns, c in ((300, 16, 32), (17, 8, 24), (1, 1, 3)):
        a = torch.randn(m, ns, c, device=dev)
        a[::7] -= 10.0                                              # all-negative rows
        b = torch.randn(m, ns, c, device=dev)
        a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        out = voxel_pool_modules.ReluAddMax.apply(a1, b1)
        ref = torch.relu(a2 + b2).max(dim=1)[0]
        assert torch.equal(out, ref)
        w = torch.randn_like(ref)
        (out * w).sum().backward()
        (ref * w).sum().backward()
        assert torch.equal(a1.grad, a2.grad) and torch.equal(b1.grad, b2.grad)
        assert float(a1.grad[::7].abs().max()) == 0.0


def _roi_scene(dev, rng, channels=8, mid=32):
    from glenet_amd import roi_grid as rg
    B, vs, pcr = 2, [0.1, 0.1, 0.2], [0.0, 0.0, 0.0, 4.0, 4.8, 2.0]
    shapes = {"x_conv1": (10, 48, 40), "x_conv2": (5, 24, 20)}
    strides = {"x_conv1": 1, "x_conv2": 2}
    feats = {}
    for name, (Z, Y, X) in shapes.items():
        idx, _, _ = _voxel_scene(rng, B, Z, Y, X, 0.12)
        feats[name] = (idx, rng.normal(size=(len(idx), channels)).astype(np.float32), [Z, Y, X])
    cfg = {n: dict(mlps=[[mid, mid]], query_ranges=[[2, 2, 2]], radii=[0.3 * strides[n]], nsamples=[16]) for n in shapes}
    torch.manual_seed(4)
    pool = rg.RoIGridPool({n: channels for n in shapes}, cfg, 3, vs, pcr).to(dev)
    for m in pool.modules():                      # non-trivial affine parameters
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    rois = np.concatenate([rng.uniform([0.5, 0.5, 0.4], [3.5, 4.3, 1.6], (B, 6, 3)), rng.uniform(0.4, 1.2, (B, 6, 3)),
                           rng.uniform(-3, 3, (B, 6, 1))], -1).astype(np.float32)
    rois[1, 5, :3] = [30.0, 30.0, 5.0]            # an RoI outside the scene: empty balls
    return pool, feats, strides, rois, B


def _run_pool(pool, feats, strides, rois, B, dev, pad_rows=0):
    tensors, leaves = {}, {}
    for name, (idx, f, shape) in feats.items():
        leaf = T(f, dev).requires_grad_(True)
        leaves[name] = leaf
        if pad_rows:      # shape-static form: capacity rows with garbage behind the live count
            n = len(idx)
            fp = torch.cat([leaf, torch.full((pad_rows, f.shape[1]), float("nan"), device=dev)])
            ip = torch.cat([T(idx, dev), torch.zeros((pad_rows, 4), dtype=torch.int32, device=dev)])
            st0 = sp.SparseConvTensor(leaf.detach(), T(idx, dev), shape, B)
            index = st0._ensure_index()
            cnt = torch.tensor([n], dtype=torch.int32, device=dev)
            st = sp.SparseConvTensor(fp, ip, shape, B, count=cnt)
            st._index = sp.CellIndex(index.grid, index.bitmap, index.flags, index.prefix, index.rank_to_row,
                                          index.row_to_rank, n + pad_rows, count=cnt, unique=cnt)
            tensors[name] = st
        else:
            tensors[name] = sp.SparseConvTensor(leaf, T(idx, dev), shape, B)
    out = pool(T(rois, dev), tensors, strides, B)
    w = torch.linspace(0.5, 1.5, out.shape[-1], device=dev)
    pool.zero_grad(set_to_none=True)
    ((out * w).square().mean() + out.mean()).backward()
    return (out.detach(), {k: v.grad.clone() for k, v in leaves.items()},
            {n: p.grad.clone() for n, p in pool.named_parameters()},
            {n: b.clone() for n, b in pool.named_buffers()})


@pytest.mark.parametrize("training", [True, False])
def test_fused_position_pool_equals_the_unfused_training_path(dev, training):
    """PosPool (csrc/glx_roipool.hip: position conv + BatchNorm from the moments of the offsets + add + ReLU + max,
    no (M, ns, C) tensor) == the row-major formulation with torch's BatchNorm on the materialised tensors:
    outputs, gradients of the sparse features, of every parameter, and the running statistics; batch statistics
    (training) and running statistics (eval with gradients)."""
    import copy
    rng = np.random.default_rng(77)
    pool, feats, strides, rois, B = _roi_scene(dev, rng)
    pool.train(training)
    state = copy.deepcopy(pool.state_dict())
    res = []
    for fused in (True, False):
        pool.load_state_dict(state)
        pool.USE_POS_POOL = fused
        res.append(_run_pool(pool, feats, strides, rois, B, dev))
    pool.USE_POS_POOL = True
    (o1, g1, p1, b1), (o2, g2, p2, b2) = res
    assert float(o1.abs().max()) > 0 and float((o1[-1] - o1[-2]).abs().max()) > 0
    np.testing.assert_allclose(o1.cpu().numpy(), o2.cpu().numpy(), rtol=2e-4, atol=2e-5)
    for k in g1:
        np.testing.assert_allclose(g1[k].cpu().numpy(), g2[k].cpu().numpy(), rtol=2e-3, atol=2e-6 + 2e-4 * float(g2[k].abs().max()))
    for k in p1:
        np.testing.assert_allclose(p1[k].cpu().numpy(), p2[k].cpu().numpy(), rtol=5e-3,
                                   atol=1e-6 + 5e-4 * float(p2[k].abs().max()), err_msg=k)
    for k in b1:
        np.testing.assert_allclose(b1[k].cpu().numpy(), b2[k].cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)


def test_position_pool_backward_forms_agree(dev):
    """glx_pos_pool_backward with the feature gradient summed in the per-block LDS table (default) and with one global atomic per
    (point, channel): the same gradients up to the order of the float sums."""
    from glenet_amd import _lib
    rng = np.random.default_rng(81)
    pool, feats, strides, rois, B = _roi_scene(dev, rng)
    pool.train()
    import copy
    state = copy.deepcopy(pool.state_dict())
    res = []
    for form in (1, 0):
        pool.load_state_dict(state)
        old = _lib.load().glx_pos_pool_set_backward_form(form)
        try:
            res.append(_run_pool(pool, feats, strides, rois, B, dev))
        finally:
            _lib.load().glx_pos_pool_set_backward_form(old)
    (o1, g1, p1, b1), (o2, g2, p2, b2) = res
    assert torch.equal(o1, o2)
    for k in g1:
        assert float(g1[k].abs().max()) > 0
        np.testing.assert_allclose(g1[k].cpu().numpy(), g2[k].cpu().numpy(), rtol=1e-4, atol=1e-6 + 1e-5 * float(g2[k].abs().max()))
    for k in p1:
        np.testing.assert_allclose(p1[k].cpu().numpy(), p2[k].cpu().numpy(), rtol=1e-3,
                                   atol=1e-6 + 1e-4 * float(p2[k].abs().max()), err_msg=k)


def test_position_pool_with_the_output_mlp_in_the_same_launch(dev):
    """glx_pos_pool_forward_out (k_rp_forward<C, true>: the layer's output Conv1d and its BatchNorm's batch statistics formed
    while a point's pooled row is in registers) against the pooling launch + GEMM + statistics pass: outputs, gradients of
    the sparse features and of every parameter, running statistics (voxel_pool_modules.py:105-108)."""
    import copy
    from glenet_amd.pcdet_ops.pointnet2.pointnet2_stack import voxel_pool_modules as vpm
    rng = np.random.default_rng(79)
    pool, feats, strides, rois, B = _roi_scene(dev, rng)
    pool.train()
    state = copy.deepcopy(pool.state_dict())
    res, used = [], []
    orig, was = vpm.pos_pool_out, vpm.POS_POOL_OUT
    vpm.pos_pool_out = lambda *a: (used.append(1), orig(*a))[1]
    try:
        for on in (True, False):
            pool.load_state_dict(state)
            vpm.POS_POOL_OUT = on
            res.append(_run_pool(pool, feats, strides, rois, B, dev))
    finally:
        vpm.POS_POOL_OUT, vpm.pos_pool_out = was, orig
    assert len(used) == len(pool.roi_grid_pool_layers)              # every scale took the fused launch, once
    (o1, g1, p1, b1), (o2, g2, p2, b2) = res
    np.testing.assert_allclose(o1.cpu().numpy(), o2.cpu().numpy(), rtol=1e-4, atol=1e-5)
    for k in g1:
        np.testing.assert_allclose(g1[k].cpu().numpy(), g2[k].cpu().numpy(), rtol=1e-3, atol=1e-6 + 1e-4 * float(g2[k].abs().max()))
    for k in p1:
        np.testing.assert_allclose(p1[k].cpu().numpy(), p2[k].cpu().numpy(), rtol=1e-3,
                                   atol=1e-6 + 1e-4 * float(p2[k].abs().max()), err_msg=k)
    for k in b1:
        np.testing.assert_allclose(b1[k].cpu().numpy(), b2[k].cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)


def test_roi_grid_training_path_on_shape_static_tensors(dev):
    """The sync-free training path (grid-point kernel + index query + live-row handling of mlps_in) on
    capacity-sized sparse tensors whose padding rows hold NaN == the exact-shape tensors, and == the generic
    module path (voxel centres / coordinates / counts built with tensor ops as voxelrcnn_head.py:106-191 does)."""
    import copy
    rng = np.random.default_rng(78)
    pool, feats, strides, rois, B = _roi_scene(dev, rng)
    pool.train()
    state = copy.deepcopy(pool.state_dict())
    exact = _run_pool(pool, feats, strides, rois, B, dev)
    pool.load_state_dict(state)
    static = _run_pool(pool, feats, strides, rois, B, dev, pad_rows=37)
    pool.load_state_dict(state)
    pool.USE_ROWS = False
    try:
        generic = _run_pool(pool, feats, strides, rois, B, dev)
    finally:
        pool.USE_ROWS = True
    for other, tol in ((static, 1e-5), (generic, 2e-4)):
        np.testing.assert_allclose(exact[0].cpu().numpy(), other[0].cpu().numpy(), rtol=tol, atol=tol)
        for k in exact[1]:
            n = exact[1][k].shape[0]
            np.testing.assert_allclose(exact[1][k].cpu().numpy(), other[1][k][:n].cpu().numpy(), rtol=5e-3,
                                       atol=1e-6 + 5e-4 * float(exact[1][k].abs().max()))
        for k in exact[2]:
            np.testing.assert_allclose(exact[2][k].cpu().numpy(), other[2][k].cpu().numpy(), rtol=5e-3,
                                       atol=1e-6 + 5e-4 * float(exact[2][k].abs().max()), err_msg=k)
        for k in exact[3]:
            np.testing.assert_allclose(exact[3][k].cpu().numpy(), other[3][k].cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
    assert torch.isfinite(static[0]).all()
    # the scales' neighbour queries as ONE launch (glx_roi_grid_query_multi) == a launch per scale, bit for bit
    pool.load_state_dict(state)
    was = pool.GROUPED_QUERY
    pool.GROUPED_QUERY = not was
    try:
        other = _run_pool(pool, feats, strides, rois, B, dev)
    finally:
        pool.GROUPED_QUERY = was
    assert torch.equal(exact[0], other[0])
    for k in exact[1]:
        np.testing.assert_allclose(exact[1][k].cpu().numpy(), other[1][k].cpu().numpy(), rtol=1e-4,
                                   atol=1e-6 + 1e-5 * float(exact[1][k].abs().max()), err_msg=k)


# ------------------------------------------------------------------ the iou3d library's remaining exports, soft-NMS
def _corner_boxes(rng, n):
    c = synth.random_boxes(rng, n, xy_range=12.0, near_dup=0.6)                      # [x,y,z,dx,dy,dz,ry]
    lo = c[:, :3] - c[:, 3:6] / 2
    hi = c[:, :3] + c[:, 3:6] / 2
    return np.concatenate([lo, hi, c[:, 6:7]], 1).astype(np.float32)                 # [x1,y1,z1,x2,y2,z2,ry]


def test_iou3d_library_iou3d_and_nms_exports(dev):
    """boxes_iou3d_{gpu,cpu}, nms_gpu, nms_3d_gpu, nms_normal_gpu of pcdet.ops.iou3d.iou3d_cuda (iou3d.cpp:98-266)
    against the oracle's restatement (its rotated overlap is pinned by the reference-built iou3d_cpu fixture)."""
    rng = np.random.default_rng(31)
    a, b = _corner_boxes(rng, 180), _corner_boxes(rng, 75)
    b[:30] = a[:30] + rng.normal(0, 0.08, (30, 7)).astype(np.float32)
    a[5, 5] = a[5, 2]                                                                 # a flat box: zero height overlap
    ref = oracle.iou3d_boxes_iou3d(a, b)
    got = torch.zeros(len(a), len(b), device=dev)
    iou3d_cuda.boxes_iou3d_gpu(T(a, dev), T(b, dev), got)
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-5, atol=2e-6)
    assert (ref > 0.2).sum() > 20 and float(got[5].abs().max()) == 0.0
    got_cpu = torch.zeros(len(a), len(b))
    iou3d_cuda.boxes_iou3d_cpu(torch.from_numpy(a), torch.from_numpy(b), got_cpu)
    np.testing.assert_allclose(got_cpu.numpy(), ref, rtol=1e-5, atol=2e-6)
    # NMS entry points: boxes in score order, host int64 keep, count returned
    n = 400
    c = _corner_boxes(rng, n)
    for kind, entry, boxes in (("bev", iou3d_cuda.nms_gpu, c[:, [0, 1, 3, 4, 6]]), ("3d", iou3d_cuda.nms_3d_gpu, c),
                               ("normal", iou3d_cuda.nms_normal_gpu, c[:, [0, 1, 3, 4, 6]])):
        boxes = np.ascontiguousarray(boxes)
        want = oracle.iou3d_nms_sorted(boxes, 0.3, kind)
        keep = torch.zeros(n, dtype=torch.int64)
        num = entry(T(boxes, dev), keep, 0.3)
        assert num == len(want) and 0 < num < n, kind
        assert np.array_equal(keep[:num].numpy(), want), kind
    # the wrappers of iou3d_utils (centre boxes + scores)
    centre = synth.random_boxes(rng, 300, xy_range=10.0, near_dup=0.6)
    scores = (rng.permutation(300).astype(np.float32) + 1) / 300
    order = np.argsort(-scores, kind="stable")
    want = order[oracle.iou3d_nms_sorted(iou3d_utils.boxes3d_to_bev_3d_torch(torch.from_numpy(centre)).numpy()[order],
                                         0.25, "3d")]
    got = iou3d_utils.nms_3d_gpu(T(centre, dev), T(scores, dev), 0.25)
    assert np.array_equal(got.cpu().numpy(), want)
    got = iou3d_utils.nms_normal_gpu(T(np.ascontiguousarray(c[:300, [0, 1, 3, 4, 6]]), dev), T(scores, dev), 0.25)
    want = order[oracle.iou3d_nms_sorted(np.ascontiguousarray(c[:300, [0, 1, 3, 4, 6]])[order], 0.25, "normal")]
    assert np.array_equal(got.cpu().numpy(), want)
    assert iou3d_utils.nms_gpu(T(centre, dev), T(scores, dev), 0.25).numel() > 0


@pytest.mark.parametrize("mode,with_var", [("gaussian", True), ("gaussian", False), ("linear", True)])
def test_softnms_vs_oracle(dev, mode, with_var):
    """softnms_gpu (iou3d_nms_utils.py:292-356: score decay by IoU with the current top box, optional variance
    voting of its first six coordinates) against the statement-by-statement restatement."""
    rng = np.random.default_rng(41)
    n = 160
    boxes = synth.random_boxes(rng, n, xy_range=9.0, near_dup=0.7)
    scores = (rng.permutation(n).astype(np.float32) + 1) / n
    var = (rng.random((n, 7)).astype(np.float32) * 0.5 + 0.05) if with_var else None
    keep_ref, boxes_ref = oracle.softnms_gpu(boxes, scores, 0.5, score_threshold=0.1, soft_mode=mode, variance=var)
    tb, ts = T(boxes, dev), T(scores, dev)
    keep, _, new_boxes = iou3d_nms_utils.softnms_gpu(tb, ts, 0.5, score_threshold=0.1, soft_mode=mode,
                                                    variance=None if var is None else T(var, dev))
    assert np.array_equal(keep.cpu().numpy(), keep_ref) and 0 < len(keep_ref) < n
    np.testing.assert_allclose(new_boxes.cpu().numpy(), boxes_ref, rtol=1e-5, atol=1e-5)
    assert new_boxes.data_ptr() == tb.data_ptr()                                      # in place, like the reference
    if with_var:
        assert np.abs(boxes_ref[:, :6] - boxes[:, :6]).max() > 1e-3                   # the vote moved boxes


def test_seed_rois_kernel_equals_the_tensor_statements(dev):
    """glx_seed_rois (bench / test helper: ground truth + offset into the first proposal slots) == the torch.where
    statements it replaces, rows without a live ground truth untouched."""
    from glenet_amd import _lib
    g = torch.Generator(device=dev).manual_seed(5)
    B, R, G = 3, 40, 9
    rois = torch.rand((B, R, 7), device=dev, generator=g)
    labels = torch.randint(1, 4, (B, R), device=dev, generator=g)
    gt = torch.rand((B, G, 8), device=dev, generator=g)
    gt[..., 7] = torch.randint(0, 3, (B, G), device=dev, generator=g).float()
    gt[1] = 0
    off = torch.tensor([0.3, -0.2, 0.05, 0.1, 0.0, -0.1, 0.07], device=dev)
    want_r, want_l = rois.clone(), labels.clone()
    has = gt[:, :, 7:8] > 0
    want_r[:, :G, :7] = torch.where(has, gt[:, :, :7] + off, want_r[:, :G, :7])
    want_l[:, :G] = torch.where(has[..., 0], gt[:, :, 7].long(), want_l[:, :G])
    _lib.call("glx_seed_rois", rois, labels, gt, off, B, R, 7, G, 8)
    assert torch.equal(rois, want_r) and torch.equal(labels, want_l)


def test_nms_early_termination_gives_the_full_sweeps_prefix(dev):
    """glx_nms_batch with max_keep on long lists (round 4): a pass over the first 2048 boxes, then the full pass whose
    kernels return at once for frames that already hold max_keep survivors.  Three kinds of frame in one batch -- (0) spread
    boxes: the prefix suffices; (1) 9000 boxes drawn around 300 clusters: fewer than 512 survive at all, the full pass
    runs; (2) a list whose first 2048 boxes are near-duplicates of 200 boxes and whose tail is spread: the 512th survivor
    lies behind the prefix -- every frame's keep[:min(num, 512)] equals the full sweep's and the oracle's."""
    from glenet_amd.pcdet_ops.iou3d_nms import iou3d_nms_cuda
    rng = np.random.default_rng(17)
    n, thr, cap = 9000, 0.8, 512

    def clustered(k, m, jitter):
        base = synth.random_boxes(rng, k, xy_range=70.0, near_dup=0.0)
        b = base[rng.integers(0, k, m)].copy()
        b[:, :2] += rng.normal(0, jitter, (m, 2)).astype(np.float32)
        b[:, 6] += rng.normal(0, 0.02, m).astype(np.float32)
        return b
    f0 = synth.random_boxes(rng, n, xy_range=70.0, near_dup=0.1)
    f1 = clustered(300, n, 0.03)
    f2 = np.concatenate([clustered(200, 2048, 0.03), synth.random_boxes(rng, n - 2048, xy_range=70.0, near_dup=0.1)])
    boxes = np.stack([f0, f1, f2]).astype(np.float32)
    full_k, full_n = iou3d_nms_cuda.nms_device_batch(T(boxes, dev), thr)
    cut_k, cut_n = iou3d_nms_cuda.nms_device_batch(T(boxes, dev), thr, max_keep=cap)
    full_k, full_n, cut_k, cut_n = (t.cpu().numpy() for t in (full_k, full_n, cut_k, cut_n))
    for f in range(3):
        want = oracle.nms_sorted(boxes[f], thr)
        assert full_n[f] == len(want) and np.array_equal(full_k[f, :full_n[f]], want)
        m = min(cap, len(want))
        assert cut_n[f] >= m and np.array_equal(cut_k[f, :m], np.asarray(want[:m])), f
    # the three frames are the three cases
    assert full_k[0, cap - 1] < 2048 and full_n[1] < cap and full_k[2, cap - 1] >= 2048


@pytest.mark.parametrize("shape", [(1, 32, 5003), (1, 16, 777, 16), (1, 64, 1), (1, 7, 70001)])
def test_channel_major_batchnorm_matches_fp64(dev, shape):
    """glx_bn_cm_train_forward / _backward (spconv.core.StackedBN): training-mode BatchNorm of a stacked (1, C, ...) tensor
    (the BatchNorm1d / 2d inputs of voxel_pool_modules.py:70-130) against an fp64 evaluation of torch.nn.BatchNorm: output,
    input gradient, gamma / beta gradients, running statistics (unbiased variance), num_batches_tracked; twice in a row
    (bitwise reproducible: fixed summation order)."""
    import torch
    from torch import nn
    from glenet_amd.spconv import core
    torch.manual_seed(sum(shape))
    c = shape[1]
    cls = nn.BatchNorm1d if len(shape) == 3 else nn.BatchNorm2d
    bn = cls(c, eps=1e-3, momentum=0.01).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    ref = cls(c, eps=1e-3, momentum=0.01).double().train()
    ref.load_state_dict({k: v.double().cpu() if v.dtype.is_floating_point else v.cpu() for k, v in bn.state_dict().items()})
    x = torch.randn(*shape, device=dev) * 2.0 + 0.7
    g = torch.randn(*shape, device=dev)
    xd = x.double().cpu().requires_grad_(True)
    if x.numel() // c > 1:
        ref(xd).backward(g.double().cpu())
    outs = []
    for _ in range(2):
        x1 = x.clone().requires_grad_(True)
        bn.zero_grad(set_to_none=True)
        y = core.stacked_train_bn(bn, x1)
        assert y is not None and y.shape == x.shape
        y.backward(g)
        outs.append((y.detach().clone(), x1.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    if x.numel() // c > 1:
        yd = torch.nn.functional.batch_norm(x.double().cpu(), None, None, ref.weight.detach(), ref.bias.detach(), True, 0.0, 1e-3)
        y, dx, dgam, dbet = outs[0]
        scale = float(yd.abs().max())
        assert float((y.double().cpu() - yd).abs().max()) <= 2e-6 * scale + 1e-6
        assert float((dx.double().cpu() - xd.grad).abs().max()) <= 2e-5 * float(xd.grad.abs().max()) + 1e-6
        assert float((dgam.double().cpu() - ref.weight.grad).abs().max()) <= 2e-5 * float(ref.weight.grad.abs().max()) + 1e-5
        assert float((dbet.double().cpu() - ref.bias.grad).abs().max()) <= 2e-5 * float(ref.bias.grad.abs().max()) + 1e-5
        assert int(bn.num_batches_tracked) == 2
        # two updates of the running statistics with the same batch statistics
        m1 = ref.running_mean.clone()           # after one reference update from (0, 1)
        mean = x.double().cpu().transpose(0, 1).reshape(c, -1).mean(1)
        var = x.double().cpu().transpose(0, 1).reshape(c, -1).var(1, unbiased=True)
        want_m = 0.99 * (0.99 * 0 + 0.01 * mean) + 0.01 * mean
        want_v = 0.99 * (0.99 * 1 + 0.01 * var) + 0.01 * var
        assert float((bn.running_mean.double().cpu() - want_m).abs().max()) <= 1e-6
        assert float((bn.running_var.double().cpu() - want_v).abs().max()) <= 1e-5
        assert float((m1 - 0.01 * mean).abs().max()) <= 1e-12
    # eval mode, image batches and CPU tensors are not this path's
    assert core.stacked_train_bn(bn.eval(), x) is None
    assert core.stacked_train_bn(bn.train(), torch.randn(2, c, 5, device=dev) if len(shape) == 3 else torch.randn(2, c, 5, 5, device=dev)) is None
