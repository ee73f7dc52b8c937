"""pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda on the HIP kernels: the nine exports of
pointnet2_batch/src/pointnet2_api.cpp:10-24 against the oracle's line-by-line restatement of
pointnet2_batch/src/*.cu (parity unpinned upstream: CUDA-only, no reference test).  Index outputs bit-identical,
copies exact, interpolation 1e-6, atomic gradients 1e-5; autograd Functions of the mirror module checked against
plain torch indexing."""
import numpy as np
import pytest
import torch

import oracle
from glenet_amd.pcdet_ops.pointnet2.pointnet2_batch import pointnet2_batch_cuda as ext
from glenet_amd.pcdet_ops.pointnet2.pointnet2_batch import pointnet2_utils as pu

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _cloud(seed, B, N, clump=True):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-4, 4, (B, N, 3)).astype(np.float32)
    if clump and N >= 64:
        x[:, : N // 4] = (x[:, : N // 4] * 0.1).astype(np.float32)      # dense region: balls overflow nsample
    return x


@pytest.mark.parametrize("B,N,m,radius,nsample", [(2, 1000, 300, 0.8, 16), (3, 4096, 1024, 0.4, 32),
                                                  (1, 70, 5, 0.5, 100), (2, 16384, 257, 1.5, 64)])
def test_ball_query_bit_identical(dev, B, N, m, radius, nsample):
    xyz = _cloud(1, B, N)
    new_xyz = _cloud(2, B, m, clump=False)
    new_xyz[:, 0] = 100.0                                                 # a ball without points keeps its zero row
    ref = oracle.batch_ball_query(radius, nsample, xyz, new_xyz)
    got = pu.ball_query(radius, nsample, T(xyz, dev), T(new_xyz, dev))
    assert got.dtype == torch.int32 and tuple(got.shape) == (B, m, nsample)
    assert np.array_equal(got.cpu().numpy(), ref)
    assert (ref[:, 0] == 0).all()
    # raw export: a caller-filled row survives an empty ball (the kernel never writes it)
    idx = torch.full((B, m, nsample), 7, dtype=torch.int32, device=dev)
    assert ext.ball_query_wrapper(B, N, m, radius, nsample, T(new_xyz, dev), T(xyz, dev), idx) == 1
    assert (idx[:, 0] == 7).all()


@pytest.mark.parametrize("B,N,m", [(2, 5000, 512), (1, 20000, 300), (3, 1024, 1024), (2, 600, 64), (2, 37, 20),
                                   (1, 1, 1), (2, 2, 2)])
def test_farthest_point_sampling_bit_identical(dev, B, N, m):
    xyz = _cloud(3, B, N)
    ref = oracle.batch_farthest_point_sample(xyz, m)
    got = pu.farthest_point_sample(T(xyz, dev), m)
    assert np.array_equal(got.cpu().numpy(), ref)
    assert (ref[:, 0] == 0).all()


@pytest.mark.parametrize("N", [8, 100, 600, 1024, 1500, 3000])
def test_farthest_point_sampling_ties_follow_the_reference_block_size(dev, N):
    """Points on a small integer lattice: many exactly equal distances, so the winner is decided by the tie rule
    of the block size opt_n_threads(N) picks (cuda_utils.h:9-13), which changes with N below 1024."""
    rng = np.random.default_rng(N)
    xyz = rng.integers(0, 4, (2, N, 3)).astype(np.float32)
    m = min(N, 40)
    ref = oracle.batch_farthest_point_sample(xyz, m)
    got = pu.farthest_point_sample(T(xyz, dev), m)
    assert np.array_equal(got.cpu().numpy(), ref)
    # the stacked module's batched export is the same entry point
    from glenet_amd.pcdet_ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as st
    idx = torch.empty((2, m), dtype=torch.int32, device=dev)
    temp = torch.full((2, N), 1e10, device=dev)
    st.farthest_point_sampling_wrapper(2, N, m, T(xyz, dev), temp, idx)
    assert np.array_equal(idx.cpu().numpy(), ref)


@pytest.mark.parametrize("B,n,m", [(2, 3000, 700), (1, 257, 5000), (3, 64, 2), (2, 10, 1)])
def test_three_nn_bit_identical(dev, B, n, m):
    unknown = _cloud(4, B, n, clump=False)
    known = _cloud(5, B, m, clump=False)
    if m >= 8:
        known[:, 5] = known[:, 2]                                          # equal distances: ascending index wins
    d_ref, i_ref = oracle.batch_three_nn(unknown, known)
    d, i = pu.three_nn(T(unknown, dev), T(known, dev))
    assert np.array_equal(i.cpu().numpy(), i_ref)
    assert np.array_equal(d.cpu().numpy(), d_ref)                         # inf where m < 3, as upstream's 1e40 narrows


@pytest.mark.parametrize("B,C,N,npoint,nsample", [(2, 16, 4096, 512, 32), (1, 3, 1000, 77, 5), (3, 67, 300, 40, 16)])
def test_group_and_gather_points_exact_and_adjoint(dev, B, C, N, npoint, nsample):
    rng = np.random.default_rng(6)
    feats = rng.normal(size=(B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, npoint, nsample)).astype(np.int32)
    idx[:, :, 1] = idx[:, :, 0]                                            # repeated rows: the scatter must add
    ref = oracle.batch_group_points(feats, idx)
    f = T(feats, dev).requires_grad_(True)
    out = pu.grouping_operation(f, T(idx, dev))
    assert np.array_equal(out.detach().cpu().numpy(), ref)
    go = rng.normal(size=ref.shape).astype(np.float32)
    out.backward(T(go, dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.batch_group_points_grad(go, idx, N), rtol=1e-5, atol=1e-5)
    # gather = grouping with one sample per point
    idx1 = rng.integers(0, N, (B, npoint)).astype(np.int32)
    f2 = T(feats, dev).requires_grad_(True)
    g = pu.gather_operation(f2, T(idx1, dev))
    assert np.array_equal(g.detach().cpu().numpy(), oracle.batch_group_points(feats, idx1))
    assert torch.equal(g.detach(), torch.gather(f2.detach(), 2, T(idx1, dev).long().unsqueeze(1).expand(-1, C, -1)))
    go1 = rng.normal(size=(B, C, npoint)).astype(np.float32)
    g.backward(T(go1, dev))
    np.testing.assert_allclose(f2.grad.cpu().numpy(), oracle.batch_group_points_grad(go1, idx1, N), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,C,m,n", [(2, 32, 500, 2000), (1, 5, 3, 100), (3, 128, 1024, 333)])
def test_three_interpolate_and_gradient(dev, B, C, m, n):
    rng = np.random.default_rng(7)
    feats = rng.normal(size=(B, C, m)).astype(np.float32)
    idx = rng.integers(0, m, (B, n, 3)).astype(np.int32)
    w = rng.uniform(0, 1, (B, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    ref = oracle.batch_three_interpolate(feats, idx, w)
    f = T(feats, dev).requires_grad_(True)
    out = pu.three_interpolate(f, T(idx, dev), T(w, dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    go = rng.normal(size=ref.shape).astype(np.float32)
    out.backward(T(go, dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.batch_three_interpolate_grad(go, idx, w, m),
                               rtol=1e-5, atol=1e-5)


def test_query_and_group_composition(dev):
    """QueryAndGroup (pointnet2_utils.py:228-265) = ball query + two groupings + centring, against the same
    statement written with torch indexing on the oracle's ball-query result."""
    B, N, m, ns, C = 2, 2048, 128, 16, 8
    xyz = _cloud(8, B, N)
    rng = np.random.default_rng(9)
    feats = rng.normal(size=(B, C, N)).astype(np.float32)
    sel = oracle.batch_farthest_point_sample(xyz, m)
    new_xyz = np.take_along_axis(xyz, sel[:, :, None].astype(np.int64), 1)
    out = pu.QueryAndGroup(0.6, ns)(T(xyz, dev), T(new_xyz, dev), T(feats, dev))
    idx = oracle.batch_ball_query(0.6, ns, xyz, new_xyz)
    gx = oracle.batch_group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - new_xyz.transpose(0, 2, 1)[..., None]
    gf = oracle.batch_group_points(feats, idx)
    assert tuple(out.shape) == (B, 3 + C, m, ns)
    assert np.array_equal(out.cpu().numpy(), np.concatenate([gx, gf], 1))
    all_ = pu.GroupAll()(T(xyz, dev), None, T(feats, dev))
    assert tuple(all_.shape) == (B, 3 + C, 1, N)


def test_host_tensors_are_refused(dev):
    from glenet_amd import _lib
    with pytest.raises(_lib.GlxError):
        ext.ball_query_wrapper(1, 4, 1, 1.0, 2, torch.zeros(1, 1, 3), torch.zeros(1, 4, 3),
                               torch.zeros(1, 1, 2, dtype=torch.int32))
