"""Empty and degenerate inputs through the pcdet.ops / spconv mirrors: the reference's callers guard most of
these (`if box_scores_nms.shape[0] > 0`, model_nms_utils.py:14; empty frames are dropped by the dataset), the
operators here return correctly shaped empty results instead of launching an empty grid."""
import numpy as np
import pytest
import torch

from glenet_amd import synth
from glenet_amd.pcdet_ops.iou3d_nms import iou3d_nms_utils
from glenet_amd.pcdet_ops.pointnet2.pointnet2_stack import pointnet2_utils
from glenet_amd.pcdet_ops.roiaware_pool3d import roiaware_pool3d_utils
from glenet_amd.pcdet_ops.roipoint_pool3d import roipoint_pool3d_utils
from glenet_amd.spconv import core as sp

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_iou_and_nms_with_no_boxes(dev):
    rng = np.random.default_rng(0)
    b = T(synth.random_boxes(rng, 5), dev)
    e = torch.zeros((0, 7), device=dev)
    assert tuple(iou3d_nms_utils.boxes_iou3d_gpu(e, b).shape) == (0, 5)
    assert tuple(iou3d_nms_utils.boxes_iou3d_gpu(b, e).shape) == (5, 0)
    assert tuple(iou3d_nms_utils.boxes_iou_bev(e, e).shape) == (0, 0)
    keep, _ = iou3d_nms_utils.nms_gpu(e, torch.zeros(0, device=dev), 0.7)
    assert keep.numel() == 0 and keep.dtype == torch.int64
    keep, _ = iou3d_nms_utils.nms_normal_gpu(e, torch.zeros(0, device=dev), 0.7)
    assert keep.numel() == 0
    # one box: kept
    keep, _ = iou3d_nms_utils.nms_gpu(b[:1], torch.ones(1, device=dev), 0.7)
    assert keep.tolist() == [0]
    # all boxes identical: exactly the best-scored one survives
    same = b[:1].repeat(300, 1)
    sc = torch.rand(300, device=dev)
    keep, _ = iou3d_nms_utils.nms_gpu(same, sc, 0.5)
    assert keep.tolist() == [int(sc.argmax())]


def test_points_in_boxes_with_no_points_or_no_boxes(dev):
    rng = np.random.default_rng(1)
    boxes = T(synth.random_boxes(rng, 4), dev)[None]                    # (1, 4, 7)
    pts = torch.rand((1, 50, 3), device=dev)
    out = roiaware_pool3d_utils.points_in_boxes_gpu(pts, boxes[:, :0])
    assert tuple(out.shape) == (1, 50) and bool((out == -1).all())
    out = roiaware_pool3d_utils.points_in_boxes_gpu(pts[:, :0], boxes)
    assert tuple(out.shape) == (1, 0)


@pytest.mark.parametrize("method", ["max", "avg"])
def test_roiaware_pool_with_no_rois_or_no_points(dev, method):
    rng = np.random.default_rng(2)
    rois = T(synth.random_boxes(rng, 3), dev)
    pts = torch.rand((40, 3), device=dev) * 10
    feat = torch.rand((40, 8), device=dev, requires_grad=True)
    pool = roiaware_pool3d_utils.RoIAwarePool3d(out_size=4, max_pts_each_voxel=16)
    out = pool(rois[:0], pts, feat, pool_method=method)
    assert tuple(out.shape) == (0, 4, 4, 4, 8)
    out = pool(rois, pts[:0], feat[:0], pool_method=method)
    assert tuple(out.shape) == (3, 4, 4, 4, 8) and float(out.detach().abs().sum()) == 0.0
    # boxes that contain no point at all: zeros, and a zero gradient
    far = rois.clone()
    far[:, :3] += 1000.0
    out = pool(far, pts, feat, pool_method=method)
    assert float(out.detach().abs().sum()) == 0.0
    out.sum().backward()
    assert float(feat.grad.abs().sum()) == 0.0


def test_roipoint_pool_with_empty_boxes(dev):
    rng = np.random.default_rng(3)
    pts = torch.rand((2, 100, 3), device=dev) * 5
    feat = torch.rand((2, 100, 6), device=dev)
    boxes = T(synth.random_boxes(rng, 6), dev).view(2, 3, 7).clone()
    boxes[..., :3] += 1000.0                                            # nothing inside any box
    pool = roipoint_pool3d_utils.RoIPointPool3d(num_sampled_points=16, pool_extra_width=0.5)
    pooled, empty = pool(pts, feat, boxes)
    assert tuple(pooled.shape) == (2, 3, 16, 9) and tuple(empty.shape) == (2, 3)
    assert bool((empty == 1).all()) and float(pooled.abs().sum()) == 0.0


def test_ball_query_with_no_queries_and_unreachable_queries(dev):
    xyz = torch.rand((60, 3), device=dev)
    cnt = torch.tensor([30, 30], dtype=torch.int32, device=dev)
    new_xyz = torch.rand((0, 3), device=dev)
    new_cnt = torch.tensor([0, 0], dtype=torch.int32, device=dev)
    idx, empty = pointnet2_utils.ball_query(0.2, 8, xyz, cnt, new_xyz, new_cnt)
    assert tuple(idx.shape) == (0, 8) and tuple(empty.shape) == (0,)
    # queries far from every point: flagged empty, indices zeroed like the reference (pointnet2_utils.py:34-36)
    far = torch.rand((4, 3), device=dev) + 100.0
    idx, empty = pointnet2_utils.ball_query(0.2, 8, xyz, cnt, far, torch.tensor([2, 2], dtype=torch.int32, device=dev))
    assert bool(empty.all()) and bool((idx == 0).all())


def test_farthest_point_sampling_degenerate_sets(dev):
    # every point identical: the sampler still returns npoint valid indices, the first one is 0
    xyz = torch.ones((1, 64, 3), device=dev)
    idx = pointnet2_utils.farthest_point_sample(xyz, 8)
    assert tuple(idx.shape) == (1, 8) and int(idx[0, 0]) == 0 and bool(((idx >= 0) & (idx < 64)).all())
    # as many samples as points: a permutation
    xyz = torch.rand((1, 32, 3), device=dev)
    idx = pointnet2_utils.farthest_point_sample(xyz, 32)
    assert sorted(idx[0].tolist()) == list(range(32))


def test_sparse_convs_on_an_empty_active_set(dev):
    torch.manual_seed(0)
    feats = torch.zeros((0, 16), device=dev)
    coords = torch.zeros((0, 4), dtype=torch.int32, device=dev)
    x = sp.SparseConvTensor(feats, coords, [8, 16, 16], 2)
    subm = sp.SubMConv3d(16, 32, 3, padding=1, bias=False, indice_key="e1").to(dev)
    down = sp.SparseConv3d(32, 32, 3, stride=2, padding=1, bias=False, indice_key="e2").to(dev)
    y = down(subm(x))
    assert tuple(y.features.shape) == (0, 32) and y.spatial_shape == [4, 8, 8]
    d = y.dense()
    assert tuple(d.shape) == (2, 32, 4, 8, 8) and float(d.detach().abs().sum()) == 0.0


def test_sparse_conv_single_voxel_and_isolated_voxels(dev):
    """One active voxel: a submanifold conv sees only the centre tap; isolated voxels do not mix."""
    torch.manual_seed(1)
    conv = sp.SubMConv3d(16, 16, 3, padding=1, bias=False, indice_key="s").to(dev)
    f = torch.randn((3, 16), device=dev)
    c = torch.tensor([[0, 1, 1, 1], [0, 5, 9, 9], [1, 1, 1, 1]], dtype=torch.int32, device=dev)
    y = conv(sp.SparseConvTensor(f, c, [8, 16, 16], 2))
    centre = conv.weight.detach().reshape(27, 16, 16)[13]       # (kz, ky, kx, Cin, Cout): the centre tap
    np.testing.assert_allclose(y.features.detach().cpu().numpy(), (f @ centre).cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_voxelizer_with_no_points_and_with_points_outside_the_range(dev):
    from glenet_amd import backbone as gb
    K = synth.KITTI
    empty = gb.voxelize_batch(torch.zeros((0, 4), device=dev), torch.zeros(0, dtype=torch.int32, device=dev), 2, K)
    assert empty["voxels"].shape[0] == 0 and tuple(empty["voxel_coords"].shape) == (0, 4)
    assert empty["voxel_offset"].tolist() == [0, 0, 0]
    assert tuple(gb.MeanVFE()(dict(empty))["voxel_features"].shape) == (0, 4)
    from glenet_amd import voxelize as gv
    f, c = gv.dynamic_voxelize_mean(torch.zeros((0, 4), device=dev), K["voxel_size"], K["point_cloud_range"])
    assert f.shape[0] == 0 and c.shape[0] == 0
    # every point outside point_cloud_range: dropped, like points_to_voxel's bounds test
    pts = torch.tensor([[-5.0, 0.0, 0.0, 0.1], [10.0, 100.0, 0.0, 0.2], [10.0, 0.0, 9.0, 0.3]], device=dev)
    out = gb.voxelize_batch(pts, torch.zeros(3, dtype=torch.int32, device=dev), 1, K)
    assert out["voxels"].shape[0] == 0
    # one inside: one voxel holding that point, zero padded to max_points
    pts = torch.cat([pts, torch.tensor([[10.0, 0.0, 0.0, 0.4]], device=dev)])
    out = gb.voxelize_batch(pts, torch.zeros(4, dtype=torch.int32, device=dev), 1, K)
    assert out["voxels"].shape[0] == 1 and int(out["voxel_num_points"][0]) == 1
    assert out["voxels"][0, 0].tolist() == pts[3].tolist() and float(out["voxels"][0, 1:].abs().sum()) == 0.0
