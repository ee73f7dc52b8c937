"""Drop-in for `pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda`
(pointnet2_batch/src/pointnet2_api.cpp:10-24): the nine exports with the reference's positional signatures
(sizes first, caller-allocated outputs), on the HIP kernels of csrc/glx_points_batch.hip and the batch front
ends of the stacked FPS / three-NN kernels.  B equal frames, features channel-major (B, C, N), indices local
to the frame.  The four wrappers that return `int` upstream return 1; the three_* ones return None."""
from .... import _lib
from ...._lib import call


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    """ball_query.cpp:28-38: new_xyz (B, m, 3), xyz (B, n, 3), idx (B, m, nsample) int32 zero-filled by the caller."""
    _lib.check_cuda(new_xyz, xyz, idx)
    call("glx_batch_ball_query", b, n, m, float(radius), nsample, new_xyz, xyz, idx)
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    """group_points.cpp:33-44: points (B, C, n), idx (B, npoints, nsample), out (B, C, npoints, nsample)."""
    _lib.check_cuda(points, idx, out)
    call("glx_batch_group_points", b, c, n, npoints, nsample, points, idx, out)
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    """group_points.cpp:20-30: grad_points (B, C, n) zero-filled by the caller."""
    _lib.check_cuda(grad_out, idx, grad_points)
    call("glx_batch_group_points_grad", b, c, n, npoints, nsample, grad_out, idx, grad_points)
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    """sampling.cpp:17-25: points (B, C, n), idx (B, npoints), out (B, C, npoints)."""
    _lib.check_cuda(points, idx, out)
    call("glx_batch_gather_points", b, c, n, npoints, points, idx, out)
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    """sampling.cpp:28-37: grad_points (B, C, n) zero-filled by the caller."""
    _lib.check_cuda(grad_out, idx, grad_points)
    call("glx_batch_gather_points_grad", b, c, n, npoints, grad_out, idx, grad_points)
    return 1


def farthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    """sampling.cpp:40-49: points (B, n, 3), temp (B, n) filled with 1e10, idx (B, m) int32."""
    _lib.check_cuda(points, temp, idx)
    call("glx_batch_fps", b, n, m, points, temp, idx)
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    """interpolate.cpp:21-31: unknown (B, n, 3), known (B, m, 3) -> dist2 (B, n, 3), idx (B, n, 3)."""
    _lib.check_cuda(unknown, known, dist2, idx)
    call("glx_batch_three_nn", b, n, m, unknown, known, dist2, idx)


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    """interpolate.cpp:34-46: points (B, c, m), idx / weight (B, n, 3), out (B, c, n)."""
    _lib.check_cuda(points, idx, weight, out)
    call("glx_batch_three_interpolate", b, c, m, n, points, idx, weight, out)


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    """interpolate.cpp:49-61: grad_out (B, c, n), grad_points (B, c, m) zero-filled by the caller."""
    _lib.check_cuda(grad_out, idx, weight, grad_points)
    call("glx_batch_three_interpolate_grad", b, c, n, m, grad_out, idx, weight, grad_points)
