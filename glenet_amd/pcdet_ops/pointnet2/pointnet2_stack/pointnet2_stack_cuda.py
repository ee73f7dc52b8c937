"""Drop-in for `pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda`
(pointnet2_stack/src/pointnet2_api.cpp:12-31): every export of the module -- the entry points on GLENet's
path and the PV-RCNN(++) set-abstraction family (FPS, three_nn / three_interpolate, vector pool)."""
from .... import _lib
from ...._lib import call


def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
    _lib.check_cuda(new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
    call("glx_ball_query", B, M, float(radius), nsample, new_xyz, new_xyz_batch_cnt, xyz,
         xyz_batch_cnt, idx)
    return 1


def voxel_query_wrapper(M, Z, Y, X, nsample, radius, z_range, y_range, x_range, new_xyz, xyz,
                        new_coords, point_indices, idx):
    _lib.check_cuda(new_xyz, xyz, new_coords, point_indices, idx)
    call("glx_voxel_query", M, Z, Y, X, nsample, float(radius), z_range, y_range, x_range, new_xyz,
         xyz, new_coords, point_indices, idx)
    return 1


def voxel_query_index_wrapper(M, Z, Y, X, nsample, radius, z_range, y_range, x_range, new_xyz, xyz,
                              new_coords, bitmap, prefix, rank_to_row, idx):
    """Same query against a SparseConvTensor's cell index (no dense map)."""
    _lib.check_cuda(new_xyz, xyz, new_coords, bitmap, prefix, idx)
    call("glx_voxel_query_index", M, Z, Y, X, nsample, float(radius), z_range, y_range, x_range,
         new_xyz, xyz, new_coords, bitmap, prefix, rank_to_row, idx)
    return 1


def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
    _lib.check_cuda(features, features_batch_cnt, idx, idx_batch_cnt, out)
    call("glx_group_points", B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out)
    return 1


def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt,
                              grad_features):
    _lib.check_cuda(grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features)
    call("glx_group_points_grad", B, M, C, N, nsample, grad_out, idx, idx_batch_cnt,
         features_batch_cnt, grad_features)
    return 1


def group_points_grad_gather_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt,
                                     grad_features):
    """Gather form of group_points_grad_wrapper (no float atomics; grad_features fully written)."""
    _lib.check_cuda(grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features)
    ws = _lib.workspace.get(_lib.query("glx_group_points_grad_workspace_bytes", M, N, nsample), grad_out.device)
    call("glx_group_points_grad_gather", B, M, C, N, nsample, grad_out, idx, idx_batch_cnt,
         features_batch_cnt, grad_features, ws, _lib.size_arg(ws.numel()))
    return 1


def group_rows_wrapper(features, idx, out):
    """Row-major grouping on the voxel query's raw output: features (N,C), idx (M,ns) GLOBAL rows with
    idx[m,0] < 0 for an empty ball, out (M,ns,C) (zeros for empty balls)."""
    _lib.check_cuda(features, idx, out)
    call("glx_group_rows", features, idx, idx.shape[0], idx.shape[1], features.shape[1], out)
    return 1


def group_rows_grad_wrapper(grad_out, idx, grad_features):
    """Gradient of group_rows_wrapper: grad_out (M,ns,C) -> grad_features (N,C), gather form."""
    _lib.check_cuda(grad_out, idx, grad_features)
    m, ns = idx.shape
    n, c = grad_features.shape
    ws = _lib.workspace.get(_lib.query("glx_group_points_grad_workspace_bytes", m, n, ns), grad_out.device)
    call("glx_group_rows_grad", grad_out, idx, m, ns, c, n, grad_features, ws, _lib.size_arg(ws.numel()))
    return 1


def relu_add_max_wrapper(a, b, out, arg):
    """out (M,C), arg (M,C) int32 = max / argmax over s of relu(a + b), a and b (M,ns,C) row-major."""
    _lib.check_cuda(a, b, out, arg)
    call("glx_relu_add_max", a, b, a.shape[0], a.shape[1], a.shape[2], out, arg)
    return 1


def relu_add_max_grad_wrapper(grad_out, out, arg, nsample, grad_in):
    _lib.check_cuda(grad_out, out, arg, grad_in)
    call("glx_relu_add_max_grad", grad_out, out, arg, out.shape[0], nsample, out.shape[1], grad_in)
    return 1


def stack_farthest_point_sampling_wrapper(xyz, temp, xyz_batch_cnt, idxs, num_sampled_points, max_points=None):
    """sampling.cpp:40-60: xyz (N,3), temp (N) filled with 1e10, idxs (sum m) int32 out.
    max_points: largest frame (points), if the caller knows it -- frames of up to 16 384 points run with
    points and running distances in registers; without it only the total N can vouch for that."""
    _lib.check_cuda(xyz, temp, xyz_batch_cnt, idxs, num_sampled_points)
    n = xyz.shape[0] if max_points is None else int(max_points)
    call("glx_stack_fps", xyz, xyz_batch_cnt, xyz_batch_cnt.shape[0], n if n <= 16384 else 0,
         num_sampled_points, temp, idxs)
    return 1


def farthest_point_sampling_wrapper(B, N, m, xyz, temp, idx):
    """sampling.cpp:24-37 (batched layout): xyz (B,N,3), temp (B,N) filled with 1e10, idx (B,m) with per-frame
    LOCAL indices -- the batch front end of the stacked kernel (same entry point as pointnet2_batch_cuda's)."""
    _lib.check_cuda(xyz, temp, idx)
    call("glx_batch_fps", B, N, m, xyz, temp, idx)
    return 1


def three_nn_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
    """interpolate.cpp:35-63: dist2 (N,3) squared distances, idx (N,3) global indices."""
    _lib.check_cuda(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx)
    call("glx_three_nn", unknown_batch_cnt.shape[0], unknown.shape[0], 0, unknown, unknown_batch_cnt,
         known, known_batch_cnt, dist2, idx)


def three_interpolate_wrapper(features, idx, weight, out):
    _lib.check_cuda(features, idx, weight, out)
    call("glx_three_interpolate", idx.shape[0], features.shape[1], features, idx, weight, out)


def three_interpolate_grad_wrapper(grad_out, idx, weight, grad_features):
    _lib.check_cuda(grad_out, idx, weight, grad_features)
    call("glx_three_interpolate_grad", idx.shape[0], grad_out.shape[1], grad_out, idx, weight, grad_features)


def _vp_ws(m, device):
    return _lib.workspace.get(_lib.query("glx_vector_pool_workspace_bytes", m), device)


def query_stacked_local_neighbor_idxs_wrapper_stack(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                                                    stack_neighbor_idxs, start_len, cumsum, avg_length_of_neighbor_idxs,
                                                    max_neighbour_distance, nsample, neighbor_type):
    """vector_pool.cpp:34-72: fills stack_neighbor_idxs / start_len (M,2) / cumsum (1,) in place."""
    _lib.check_cuda(support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, stack_neighbor_idxs, start_len, cumsum)
    m = new_xyz.shape[0]
    ws = _vp_ws(m, new_xyz.device)
    call("glx_query_stacked_local_neighbor_idxs", support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
         xyz_batch_cnt.shape[0], m, stack_neighbor_idxs, start_len, cumsum, int(avg_length_of_neighbor_idxs),
         float(max_neighbour_distance), int(nsample), int(neighbor_type), ws, _lib.size_arg(ws.numel()))
    return 0


def query_three_nn_by_stacked_local_idxs_wrapper_stack(support_xyz, new_xyz, new_xyz_grid_centers, new_xyz_grid_idxs,
                                                       new_xyz_grid_dist2, stack_neighbor_idxs, start_len, M,
                                                       num_total_grids):
    """vector_pool.cpp:75-112."""
    _lib.check_cuda(support_xyz, new_xyz_grid_centers, new_xyz_grid_idxs, new_xyz_grid_dist2, stack_neighbor_idxs,
                    start_len)
    call("glx_query_three_nn_by_stacked_local_idxs", support_xyz, new_xyz_grid_centers, new_xyz_grid_idxs,
         new_xyz_grid_dist2, stack_neighbor_idxs, start_len, int(M), int(num_total_grids))
    return 0


def vector_pool_wrapper(support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, new_features,
                        new_local_xyz, point_cnt_of_grid, grouped_idxs, num_grid_x, num_grid_y, num_grid_z,
                        max_neighbour_distance, use_xyz, num_max_sum_points, nsample, neighbor_type, pooling_type):
    """vector_pool.cpp:115-170 -> cum_sum (python int, as the reference's cudaMemcpy returns it)."""
    import torch
    _lib.check_cuda(support_xyz, xyz_batch_cnt, support_features, new_xyz, new_xyz_batch_cnt, new_features,
                    new_local_xyz, point_cnt_of_grid, grouped_idxs)
    m = new_xyz.shape[0]
    cum = torch.zeros(1, dtype=torch.int32, device=new_xyz.device)
    ws = _vp_ws(m, new_xyz.device)
    call("glx_vector_pool", support_xyz, support_features, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
         xyz_batch_cnt.shape[0], m, support_features.shape[1], new_features.shape[1], int(num_grid_x), int(num_grid_y),
         int(num_grid_z), float(max_neighbour_distance), 1 if use_xyz else 0, int(num_max_sum_points), int(nsample),
         int(neighbor_type), int(pooling_type), new_features, new_local_xyz, point_cnt_of_grid, grouped_idxs, cum, ws,
         _lib.size_arg(ws.numel()))
    return int(cum.item())


def vector_pool_grad_wrapper(grad_new_features, point_cnt_of_grid, grouped_idxs, grad_support_features):
    """vector_pool.cpp:173-200."""
    _lib.check_cuda(grad_new_features, point_cnt_of_grid, grouped_idxs, grad_support_features)
    call("glx_vector_pool_grad", grad_new_features, point_cnt_of_grid, grouped_idxs, grouped_idxs.shape[0],
         grad_support_features.shape[1], grad_new_features.shape[1], point_cnt_of_grid.shape[1], grad_support_features)
    return 0


def voxel_pool_agg_wrapper(M, nsample, Cm, Co, feats, xyz, new_xyz, idx, empty, w_pos, b_pos, w_out,
                           b_out, out):
    """Fused grouping + position MLP + ReLU + max-pool + output MLP of one RoI-grid pooling scale
    (inference; no reference counterpart -- it replaces the tensor ops of
    voxel_pool_modules.py:88-108)."""
    _lib.check_cuda(feats, xyz, new_xyz, idx, empty, w_pos, b_pos, w_out, b_out, out)
    call("glx_voxel_pool_agg", feats, xyz, new_xyz, idx, empty, M, nsample, Cm, Co, w_pos, b_pos,
         w_out, b_out, out)
