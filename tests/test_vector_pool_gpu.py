"""VectorPool family of PV-RCNN++: the four remaining pointnet2_stack_cuda exports and the two
autograd wrappers (pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py:306-452) against the oracle's restatement
of vector_pool_gpu.cu.  Index outputs bit-identical (with the segment order both define: ascending new point),
pooled sums bit-identical (same summation order), the atomic gradient to 1e-5; the host retry loop of both wrappers
is exercised with buffers that start too small."""
import numpy as np
import pytest
import torch

import oracle
from glenet_amd.pcdet_ops.pointnet2.pointnet2_stack import pointnet2_stack_cuda as ext
from glenet_amd.pcdet_ops.pointnet2.pointnet2_stack import pointnet2_utils as pu

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _scene(seed, n=(1500, 900, 0, 1200), m=(96, 64, 8, 80), c=16):
    rng = np.random.default_rng(seed)
    sx = rng.uniform(0, 6, (sum(n), 3)).astype(np.float32)
    sx[:1300] = (sx[:1300] * 0.05 + 3).astype(np.float32)             # a dense clump: > 1000 neighbours for some points
    sf = rng.normal(size=(sum(n), c)).astype(np.float32)
    nx = rng.uniform(0.5, 5.5, (sum(m), 3)).astype(np.float32)
    nx[0] = [3.1, 3.1, 3.1]
    nx[1] = [50, 50, 50]                                               # no neighbour at all
    return sx, sf, nx, np.array(n, np.int32), np.array(m, np.int32)


@pytest.mark.parametrize("neighbor_type,nsample", [(0, -1), (1, -1), (0, 7), (1, 1500)])
def test_local_neighbor_lists_and_three_nn_bit_identical(dev, neighbor_type, nsample):
    sx, sf, nx, n, m = _scene(1)
    M = len(nx)
    for avg in (1, 64):                                                # 1: the buffer overflows (truncation rule)
        stack, start_len, cum = oracle.query_stacked_local_neighbor_idxs(sx, n, nx, m, avg, 0.8, nsample, neighbor_type)
        g_stack = torch.zeros(max(avg * M, 1), dtype=torch.int32, device=dev)
        g_sl = torch.zeros((M, 2), dtype=torch.int32, device=dev)
        g_cum = torch.zeros(1, dtype=torch.int32, device=dev)
        ext.query_stacked_local_neighbor_idxs_wrapper_stack(T(sx, dev), T(n, dev), T(nx, dev), T(m, dev), g_stack, g_sl,
                                                            g_cum, avg, 0.8, nsample, neighbor_type)
        assert int(g_cum) == cum and np.array_equal(g_sl.cpu().numpy(), start_len)
        keep = min(cum, avg * M)
        assert np.array_equal(g_stack.cpu().numpy()[:keep], stack[:keep])
    assert start_len[1, 1] == 0 and start_len[:, 1].max() == (1000 if nsample < 0 else min(nsample, 1000))
    # the autograd wrapper with its retry loop (starts at avg 2) == the oracle's
    G = 8
    rng = np.random.default_rng(2)
    centers = (nx[:, None, :] + rng.uniform(-0.4, 0.4, (M, G, 3))).astype(np.float32)
    d, idx, avg = oracle.three_nn_for_vector_pool_by_two_step(sx, n, nx, centers, m, 0.4, nsample, neighbor_type, 2, G, 2.0)
    gd, gi, gavg = pu.three_nn_for_vector_pool_by_two_step(T(sx, dev), T(n, dev), T(nx, dev), T(centers, dev), T(m, dev),
                                                           0.4, nsample, neighbor_type, 2, G, 2.0)
    assert int(gavg) == avg and np.array_equal(gi.cpu().numpy(), idx)
    assert np.array_equal(gd.cpu().numpy(), d)
    assert (idx[1] == -1).all() and np.isinf(d[1]).all()


@pytest.mark.parametrize("pooling_type,neighbor_type,nsample,grid,c_each", [
    (0, 0, -1, (3, 3, 3), 8), (0, 1, -1, (2, 2, 2), 16), (0, 0, 20, (2, 3, 2), 4), (1, 0, -1, (3, 3, 3), 8),
    (1, 1, 5, (2, 2, 2), 16)])
def test_vector_pool_forward_and_gradient(dev, pooling_type, neighbor_type, nsample, grid, c_each):
    sx, sf, nx, n, m = _scene(3)
    want = oracle.vector_pool(sx, n, sf, nx, m, grid, 0.9, c_each, True, num_mean_points_per_grid=3, nsample=nsample,
                              neighbor_type=neighbor_type, pooling_type=pooling_type)
    feats = T(sf, dev).requires_grad_(True)
    got = pu.vector_pool_with_voxel_query_op(T(sx, dev), T(n, dev), feats, T(nx, dev), T(m, dev), *grid, 0.9, c_each, True,
                                             3, nsample, neighbor_type, pooling_type)
    nf, nl, mean, pc = got
    assert int(mean) == want[2] and np.array_equal(pc.cpu().numpy(), want[3])
    assert np.array_equal(nf.detach().cpu().numpy(), want[0])          # same summation order -> same bits
    assert np.array_equal(nl.cpu().numpy(), want[1])
    assert float(nf[1].abs().sum()) == 0 and int(pc[1].sum()) == 0     # the isolated point
    g_out = torch.randn(nf.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    nf.backward(g_out)
    if pooling_type == 0:
        # the gradient kernel itself carries the 1 / count of the average (vector_pool_gpu.cu:457-459)
        ref = oracle.vector_pool_grad(g_out.cpu().numpy(), want[3], want[4], len(sx), sf.shape[1])
        np.testing.assert_allclose(feats.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
        assert float(feats.grad.abs().sum()) > 0


def test_vector_pool_raw_extension_rows(dev):
    """vector_pool_wrapper itself: rows of grouped_idxs (ascending new point, ascending k inside) and the
    nothing-is-written rule when the row buffer is too small."""
    sx, sf, nx, n, m = _scene(4)
    M, G, cg = len(nx), 8, 4
    bufs = lambda rows: (torch.zeros((M, G * cg), device=dev), torch.zeros((M, 3 * G), device=dev),
                         torch.zeros((M, G), dtype=torch.int32, device=dev),
                         torch.zeros((rows, 3), dtype=torch.int32, device=dev))
    nf, nl, pc, gi = bufs(10)
    cum = ext.vector_pool_wrapper(T(sx, dev), T(n, dev), T(sf, dev), T(nx, dev), T(m, dev), nf, nl, pc, gi, 2, 2, 2, 0.9,
                                  1, 10, -1, 0, 0)
    assert cum > 10 and float(nf.abs().sum()) == 0 and int(pc.sum()) == 0
    nf, nl, pc, gi = bufs(cum)
    cum2 = ext.vector_pool_wrapper(T(sx, dev), T(n, dev), T(sf, dev), T(nx, dev), T(m, dev), nf, nl, pc, gi, 2, 2, 2, 0.9,
                                   1, cum, -1, 0, 0)
    assert cum2 == cum
    nf_o = np.zeros((M, G * cg), np.float32); nl_o = np.zeros((M, 3 * G), np.float32)
    pc_o = np.zeros((M, G), np.int32); gi_o = np.zeros((cum, 3), np.int32)
    import ctypes
    f = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    total = oracle.lib().orc_vector_pool(f(sx), f(sf), f(n), f(nx), f(m), len(n), M, sf.shape[1], G * cg, 2, 2, 2,
                                         ctypes.c_float(0.9), 1, cum, -1, 0, 0, f(nf_o), f(nl_o), f(pc_o), f(gi_o))
    assert total == cum and np.array_equal(gi.cpu().numpy(), gi_o)
    assert np.array_equal(nf.cpu().numpy(), nf_o) and np.array_equal(pc.cpu().numpy(), pc_o)
