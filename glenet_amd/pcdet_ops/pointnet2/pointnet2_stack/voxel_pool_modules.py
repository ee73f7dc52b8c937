"""Mirror of pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py: the RoI-grid pooling set
abstraction of Voxel-RCNN (called from voxelrcnn_head.py:106-191).  Same constructor keywords,
submodule names (groupers / mlps_in / mlps_pos / mlps_out, so checkpoints load) and forward
signature; the query + grouping run on glenet_amd kernels."""
import os

import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import voxel_query_utils


def _conv_bn(cin, cout, dims, relu=False):
    conv = (nn.Conv1d if dims == 1 else nn.Conv2d)(cin, cout, kernel_size=1, bias=False)
    bn = (nn.BatchNorm1d if dims == 1 else nn.BatchNorm2d)(cout)
    return nn.Sequential(conv, bn, nn.ReLU()) if relu else nn.Sequential(conv, bn)


class GroupRows(torch.autograd.Function):
    """out[m, s, :] = features[idx[m, s], :] on the voxel query's raw (M, ns) global rows (zeros for an
    empty ball): the (M, ns, C) layout of the row-major training path (csrc/glx_points.hip,
    k_group_rows; gradient = the gather form on contiguous rows)."""

    @staticmethod
    def forward(ctx, features, idx):
        from . import pointnet2_stack_cuda as pointnet2
        features = features.contiguous()
        out = torch.empty((idx.shape[0], idx.shape[1], features.shape[1]), dtype=features.dtype,
                          device=features.device)
        pointnet2.group_rows_wrapper(features, idx, out)
        ctx.save_for_backward(idx)
        ctx.n = features.shape[0]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import pointnet2_stack_cuda as pointnet2
        idx, = ctx.saved_tensors
        grad = torch.empty((ctx.n, grad_out.shape[2]), dtype=grad_out.dtype, device=grad_out.device)
        pointnet2.group_rows_grad_wrapper(grad_out.contiguous(), idx, grad)
        return grad, None


class ReluAddMax(torch.autograd.Function):
    """max over the neighbours of relu(a + b) on (M, ns, C) rows in one pass; one gradient tensor for
    both inputs (csrc/glx_points.hip, k_relu_add_max)."""

    @staticmethod
    def forward(ctx, a, b):
        from . import pointnet2_stack_cuda as pointnet2
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty((a.shape[0], a.shape[2]), dtype=a.dtype, device=a.device)
        arg = torch.empty((a.shape[0], a.shape[2]), dtype=torch.int32, device=a.device)
        pointnet2.relu_add_max_wrapper(a, b, out, arg)
        ctx.save_for_backward(out, arg)
        ctx.ns = a.shape[1]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import pointnet2_stack_cuda as pointnet2
        out, arg = ctx.saved_tensors
        grad = torch.empty((out.shape[0], ctx.ns, out.shape[1]), dtype=out.dtype, device=out.device)
        pointnet2.relu_add_max_grad_wrapper(grad_out.contiguous(), out, arg, ctx.ns, grad)
        return grad, grad


class PosPool(torch.autograd.Function):
    """max_s relu(feats[idx[m,s]] + BatchNorm(Conv1x1(xyz[idx[m,s]] - new_xyz[m]))) in one pass forward and one
    backward (csrc/glx_roipool.hip): the position branch + add + ReLU + max-pool of forward() below without any
    (M, nsample, C) tensor; BatchNorm statistics from the moments of the offsets."""

    @staticmethod
    def forward(ctx, feats, w_pos, gamma, beta, bn, idx, xyz, new_xyz, w_out=None, bn_out=None):
        """w_out (C, C) / bn_out: the layer's output MLP (mlps_out's Conv1d weight and its training-mode BatchNorm1d) --
        y_out = pooled @ w_out^T and the BatchNorm's batch statistics are formed in the pooling launch
        (glx_pos_pool_forward_out); returns (pooled, arg, y_out, coef, mean, invstd) then."""
        import ctypes
        from .... import _lib
        feats, w, xyz, new_xyz, idx = (feats.contiguous().float(), w_pos.reshape(w_pos.shape[0], 3).contiguous().float(),
                                       xyz.contiguous().float(), new_xyz.contiguous().float(), idx.contiguous())
        _lib.check_cuda(feats, w, xyz, new_xyz, idx)
        (n, c), (m, ns) = feats.shape, idx.shape
        dev = feats.device
        training = bn.training or not bn.track_running_stats
        pooled = torch.empty((m, c), dtype=torch.float32, device=dev)
        arg = torch.empty((m, c), dtype=torch.uint8, device=dev)
        save = torch.empty(_lib.query("glx_pos_pool_save_floats", c), dtype=torch.float32, device=dev)
        moments = torch.empty(9, dtype=torch.float64, device=dev)
        ws = _lib.workspace.get(_lib.query("glx_pos_pool_workspace_bytes", c), dev)
        rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
        extra = ()
        if w_out is not None:
            from ....spconv import core
            wo = w_out.detach().reshape(c, c).contiguous().float()
            y_out = torch.empty((m, c), dtype=torch.float32, device=dev)
            stats = tuple(torch.empty(k, dtype=torch.float32, device=dev) for k in (2 * c, c, c))
            st = _lib.bn_stats(core._bn_state(dev), bn_out, *stats)
            _lib.call("glx_pos_pool_forward_out", feats, n, c, xyz, new_xyz, idx, m, ns, w, gamma, beta, rm, rv,
                      ctypes.c_float(bn.momentum if bn.momentum is not None else 0.1), ctypes.c_float(bn.eps),
                      1 if training else 0, pooled, arg, save, moments, wo, y_out, ctypes.byref(st), ws,
                      _lib.size_arg(ws.numel()))
            if bn_out.track_running_stats:
                _lib.bump_weights_epoch((bn_out.running_mean, bn_out.running_var))
            extra = (y_out,) + stats
            ctx.wo = wo
        else:
            _lib.call("glx_pos_pool_forward", feats, n, c, xyz, new_xyz, idx, m, ns, w, gamma, beta, rm, rv,
                      ctypes.c_float(bn.momentum if bn.momentum is not None else 0.1), ctypes.c_float(bn.eps),
                      1 if training else 0, pooled, arg, save, moments, ws, _lib.size_arg(ws.numel()))
            ctx.wo = None
        if training and rm is not None:
            _lib.bump_weights_epoch((rm, rv))             # running statistics updated through raw pointers
        ctx.save_for_backward(feats, w, gamma, pooled, arg, idx, xyz, new_xyz, save, moments)
        ctx.training, ctx.wshape = training, w_pos.shape
        ctx.wo_shape = w_out.shape if w_out is not None else None
        if extra:
            ctx.mark_non_differentiable(arg, *extra[1:])
            ctx.set_materialize_grads(False)
        else:
            ctx.mark_non_differentiable(arg)
        return (pooled, arg) + extra

    @staticmethod
    def backward(ctx, dpooled, _darg, dy_out=None, *_stats):
        from .... import _lib
        feats, w, gamma, pooled, arg, idx, xyz, new_xyz, save, moments = ctx.saved_tensors
        (n, c), (m, ns) = feats.shape, idx.shape
        dev = feats.device
        d_wo = None
        if ctx.wo is not None and dy_out is not None:
            # the output MLP's two gradients: into the pooled rows, and the (C, C) filter -- rows in 128 batched products
            # so that the contraction over M rows is a batched GEMM + a sum, not one 32 x 32 tile with K = M
            dy_out = dy_out.contiguous().float()
            g = dy_out @ ctx.wo
            dpooled = g if dpooled is None else dpooled.contiguous().float() + g
            s_ = 128 if (m % 128 == 0 and m >= 128 * 64) else 1
            d_wo = torch.bmm(dy_out.view(s_, m // s_, c).transpose(1, 2), pooled.view(s_, m // s_, c)).sum(0).view(ctx.wo_shape)
        elif dpooled is None:
            dpooled = torch.zeros_like(pooled)
        dfeats = torch.empty_like(feats)
        dw = torch.empty((c, 3), dtype=torch.float32, device=dev)
        dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        dbeta = torch.empty(c, dtype=torch.float32, device=dev)
        ws = _lib.workspace.get(_lib.query("glx_pos_pool_workspace_bytes", c), dev)
        _lib.call("glx_pos_pool_backward", dpooled.contiguous().float(), pooled, arg, idx, xyz, new_xyz, m, ns, c, n, w,
                  gamma, save, moments, 1 if ctx.training else 0, dfeats, dw, dgamma, dbeta, ws,
                  _lib.size_arg(ws.numel()))
        return (dfeats, dw.view(ctx.wshape), dgamma if gamma is not None else None,
                dbeta if gamma is not None else None, None, None, None, None, d_wo, None)


def _count(bn):
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        from ....spconv import core
        if core.DEFERRED_COUNTERS is not None:       # a training step adds 1 to all its counters in one launch
            core.DEFERRED_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1


def pos_pool(feats, mlp_pos, idx, xyz, new_xyz):
    """mlp_pos = Sequential(Conv2d(3, C, 1, bias=False), BatchNorm2d(C)) -> pooled (M, C)."""
    conv, bn = mlp_pos[0], mlp_pos[1]
    _count(bn)
    return PosPool.apply(feats, conv.weight, bn.weight, bn.bias, bn, idx, xyz, new_xyz)[0]


# Off by default: measured on the GLENet-VR step (A/B on one box) 6.475 ms with it against 6.419 without -- the exchange of
# the pooled row between a point's lanes and the statistics tail (atomics, ticket, last-block finalize) cost the pooling
# launch more than the GEMM and the statistics pass they replace (14 + 13 us per scale).  Kept for the record and its test.
POS_POOL_OUT = False


def pos_pool_out_supported(feats, mlp_pos, mlp_out):
    """The pooling launch can carry the layer's output MLP: Sequential(Conv1d(C, C, 1, bias=False), BatchNorm1d(C), ReLU)
    in training mode with C <= 32 (csrc/glx_roipool.hip k_rp_forward<C, true>)."""
    from ....spconv import core
    if not (POS_POOL_OUT and torch.is_grad_enabled() and pos_pool_supported(feats, mlp_pos) and len(mlp_out) == 3):
        return False
    conv, bn, act = mlp_out[0], mlp_out[1], mlp_out[2]
    c = feats.shape[1]
    return (isinstance(conv, nn.Conv1d) and conv.kernel_size == (1,) and conv.bias is None and conv.in_channels == c
            and conv.out_channels == c and c <= 32 and isinstance(bn, nn.BatchNorm1d) and isinstance(act, nn.ReLU)
            and bn.training and bn.affine and core.USE_BN_STATE and core.USE_FUSED_TRAIN_BN and bn.momentum is not None)


def pos_pool_out(feats, mlp_pos, mlp_out, idx, xyz, new_xyz):
    """relu(bn_out(conv_out(pos_pool(...)))) (M, C): pooling, the output convolution and the BatchNorm statistics in one
    launch, the transform in a second (spconv.core.FusedBNApply)."""
    from ....spconv import core
    conv, bn = mlp_pos[0], mlp_pos[1]
    _count(bn)
    _count(mlp_out[1])
    _, _, y, coef, mean, invstd = PosPool.apply(feats, conv.weight, bn.weight, bn.bias, bn, idx, xyz, new_xyz,
                                                mlp_out[0].weight, mlp_out[1])
    return core.FusedBNApply.apply(y, coef, mean, invstd, mlp_out[1].weight, mlp_out[1].bias, True)


def pos_pool_supported(feats, mlp_pos):
    conv, bn = mlp_pos[0], mlp_pos[1]
    return (feats.is_cuda and feats.dtype == torch.float32 and len(mlp_pos) == 2 and conv.bias is None
            and conv.in_channels == 3 and conv.out_channels in (16, 32, 64) and conv.out_channels == feats.shape[1]
            and isinstance(bn, (nn.BatchNorm1d, nn.BatchNorm2d)))


ROWS_LINEAR_FN = True


class _RowsLinearFn(torch.autograd.Function):
    """x (rows, C_in) @ w^T for tall-skinny products (tens of thousands of rows x <= 64 channels) with the backward written
    out: dX = dY @ w, dW = sum over 128 row groups of dY_g^T X_g (split-K as a batched product + one sum, in the weight's own
    layout).  The bmm-with-expanded-weight formulation this replaces left autograd a second bmm, the sum over the expanded
    dimension AND a transpose copy per call (tools/torch_ops_in_step.py: 12 bmm + 6 sum + 7 copy_ per training step)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        rows = x.shape[0]
        gx = gy @ w if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            gw = torch.bmm(gy.view(128, rows // 128, -1).transpose(1, 2), x.view(128, rows // 128, -1)).sum(0)
        return gx, gw


ROWS_CONV_BN = True
ROWS_64_128_F16X2 = True      # RowsConvBN's 64 -> 128 layer on the 16-bit matrix pipe, forward and backward (False: fp32 MFMA, two column halves)


class RowsConvBN(torch.autograd.Function):
    """Sequential(Conv(k = 1, bias = False), BatchNorm[, ReLU]) in training mode on a (rows, C_in) matrix, csrc/glx_rows.hip
    (voxel_pool_modules.py:70-130's mlps_in / mlps_out): forward = the product with the BatchNorm statistics in its epilogue +
    the transform (2 launches); backward = the BatchNorm's backward sums, then ONE launch that applies the BatchNorm / ReLU
    backward to dy on load and forms both dX = dZ W and the per-block partials of dW = dZ^T X, then their fixed-order sum
    (3 launches).  The library formulation this replaces: 3 launches forward, 5 backward, the products at 17 - 46 us each."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bn, relu, count, lazy=False):
        """lazy: the transform is NOT applied: returns (z, coef) -- the raw product and the BatchNorm's scale | shift -- for a
        consumer that reads relu?(z scale + shift) on load (dense_path.PointMaxBN); the gradient it hands back for z is taken as the
        gradient of the TRANSFORMED output, as in the eager form."""
        import ctypes
        from .... import _lib
        from ....spconv import core
        x = x.contiguous().float()
        _lib.check_cuda(x, weight)
        rows, cin = x.shape
        cout = weight.shape[0]
        dev = x.device
        w = weight.detach().reshape(cout, cin).contiguous().float()
        z = torch.empty((rows, cout), dtype=torch.float32, device=dev)
        y = None if lazy else torch.empty((rows, cout), dtype=torch.float32, device=dev)
        coef = torch.empty(2 * cout, dtype=torch.float32, device=dev)
        mean = torch.empty(cout, dtype=torch.float32, device=dev)
        invstd = torch.empty(cout, dtype=torch.float32, device=dev)
        rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
        if ROWS_64_128_F16X2 and (cin, cout) == (64, 128):
            # the CVAE's second point layer: f16 x 2 products (the fp32-MFMA form is matrix-bound there), csrc/glx_pointnet.hip
            from ....dense_path import PointFeat
            wh, ew = PointFeat._f16x2_image(w)
            _lib.call("glx_rows_linear_bn_forward_64_128_f16x2", x, rows, wh, ew, count, z, gamma, beta, ctypes.c_float(bn.eps),
                      ctypes.c_float(bn.momentum), rm, rv, coef, mean, invstd, core._bn_state(dev))
        else:
            _lib.call("glx_rows_linear_bn_forward", x, rows, cin, w, cout, count, z, gamma, beta, ctypes.c_float(bn.eps),
                      ctypes.c_float(bn.momentum), rm, rv, coef, mean, invstd, core._bn_state(dev))
        if not lazy:
            _lib.call("glx_bn_apply_forward", z, coef, 1 if relu else 0, rows, cout, count, y, 0)
        if rm is not None:
            _lib.bump_weights_epoch((rm, rv))             # running statistics updated through raw pointers
        ctx.save_for_backward(x, w, z, coef, mean, invstd, gamma, beta)
        ctx.relu, ctx.count, ctx.wshape = relu, count, weight.shape
        ctx.leaf = weight if weight.is_leaf else None
        ctx.lazy = bool(lazy)
        if lazy:
            ctx.mark_non_differentiable(coef)
            return z.view_as(z), coef
        return y

    @staticmethod
    def backward(ctx, dy, *_):
        from .... import _lib
        from ....spconv import core
        x, w, z, coef, mean, invstd, gamma, beta = ctx.saved_tensors
        dy = dy.contiguous().float()
        rows, cin = x.shape
        cout = w.shape[0]
        dev = x.device
        coef3 = torch.empty(3 * cout, dtype=torch.float32, device=dev)
        dgamma = torch.empty(cout, dtype=torch.float32, device=dev)
        dbeta = torch.empty(cout, dtype=torch.float32, device=dev)
        from .... import dense_path
        taken = dense_path.BWD_PARTIALS.pop(dy.data_ptr(), None)
        if taken is not None and taken[2] == z.data_ptr() and ctx.relu and ctx.count is None and cout == 128:
            # the launches that wrote dy took the sums on their way (dense_path.PointMaxBN.backward): no pass over dy and z
            pa, pb = taken[0], taken[1]
            _lib.call("glx_bn_backward_from_partials", pa, int(pa.shape[0]), pb, int(pb.shape[0]), cout, ctypes.c_longlong(rows), gamma,
                      mean, invstd, dgamma, dbeta, coef3)
        else:
            _lib.call("glx_bn_backward_sums", z, dy, rows, cout, gamma, beta, mean, invstd, 1 if ctx.relu else 0, dgamma, dbeta,
                      ctx.count, coef3, core._bn_state(dev))
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            # written where the optimizer reads it when the filter is a leaf it owns (no gather copy afterwards)
            gw = _lib.grad_buffer(ctx.leaf, (cout, cin)) if ctx.leaf is not None else torch.empty((cout, cin), dtype=torch.float32, device=dev)
        if ROWS_64_128_F16X2 and (cin, cout) == (64, 128):
            # one pass, both products on the 16-bit matrix pipe (csrc/glx_pointnet.hip, k_rows_bwd_64_128_f16)
            from ....dense_path import PointFeat
            wth, ewt = PointFeat._f16x2_image(w.t())
            ws = _lib.workspace.get(_lib.query("glx_rows_bwd_64_128_workspace_bytes"), dev)
            _lib.call("glx_rows_linear_bn_backward_64_128_f16x2", x, z, dy, rows, wth, ewt, ctx.count, coef, 1 if ctx.relu else 0,
                      coef3, mean, invstd, gx, gw, ws, _lib.size_arg(ws.numel()))
        else:
            ws = _lib.workspace.get(_lib.query("glx_rows_linear_workspace_bytes", cin, cout), dev)
            _lib.call("glx_rows_linear_bn_backward", x, z, dy, rows, cin, w, cout, ctx.count, coef, 1 if ctx.relu else 0, coef3,
                      mean, invstd, gx, gw, ws, _lib.size_arg(ws.numel()))
        return (gx, gw.view(ctx.wshape) if gw is not None else None, dgamma if gamma is not None else None,
                dbeta if gamma is not None else None, None, None, None, None)


def rows_conv_bn_supported(seq, x2d):
    """Sequential(Conv1d / Conv2d (k = 1, bias = False), BatchNorm[, ReLU]) in training mode that RowsConvBN covers."""
    from .... import _lib
    from ....spconv import core
    if not (ROWS_CONV_BN and x2d.is_cuda and x2d.dtype == torch.float32 and x2d.dim() == 2 and x2d.shape[0] >= 1024
            and len(seq) in (2, 3)):
        return False
    conv, bn = seq[0], seq[1]
    if len(seq) == 3 and not isinstance(seq[2], nn.ReLU):
        return False
    return (isinstance(conv, (nn.Conv1d, nn.Conv2d)) and all(k == 1 for k in conv.kernel_size) and conv.bias is None
            and conv.groups == 1 and conv.in_channels == x2d.shape[1] and conv.weight.dtype == torch.float32
            and isinstance(bn, (nn.BatchNorm1d, nn.BatchNorm2d)) and bn.training and bn.affine and bn.momentum is not None
            and core.USE_BN_STATE and core.USE_FUSED_TRAIN_BN and torch.is_grad_enabled()
            and bool(_lib.query("glx_rows_linear_supported", conv.in_channels, conv.out_channels)))


def rows_conv_bn(seq, x2d, count=None):
    """count: device int32 live-row count of a shape-static matrix whose rows past it are ZERO (statistics over the live rows,
    zero output rows and zero gradients past them)."""
    conv, bn = seq[0], seq[1]
    _count(bn)
    return RowsConvBN.apply(x2d, conv.weight, bn.weight, bn.bias, bn, len(seq) > 2, count)


class NeighborVoxelSAModuleMSG(nn.Module):
    def __init__(self, *, query_ranges, radii, nsamples, mlps, use_xyz=True, pool_method='max_pool'):
        super().__init__()
        assert len(query_ranges) == len(nsamples) == len(mlps)
        self.groupers, self.mlps_in = nn.ModuleList(), nn.ModuleList()
        self.mlps_pos, self.mlps_out = nn.ModuleList(), nn.ModuleList()
        for rng, radius, nsample, (c_in, c_mid, c_out) in zip(query_ranges, radii, nsamples, mlps):
            self.groupers.append(voxel_query_utils.VoxelQueryAndGrouping(rng, radius, nsample))
            self.mlps_in.append(_conv_bn(c_in, c_mid, 1))
            self.mlps_pos.append(_conv_bn(3, c_mid, 2))
            self.mlps_out.append(_conv_bn(c_mid, c_out, 1, relu=True))
        self.relu = nn.ReLU()
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, new_coords, features,
                voxel2point_indices):
        """xyz (N,3) voxel centres, new_xyz (M,3) grid points, new_coords (M,4) [b,x,y,z] voxel
        coords of the grid points, features (N,C), voxel2point_indices: dense (B,Z,Y,X) map or the
        SparseConvTensor itself -> (M, sum of mlps[k][-1])."""
        coords_bzyx = new_coords[:, [0, 3, 2, 1]].contiguous()      # voxel_pool_modules.py:84
        if self._fusable(features):
            return self._forward_fused(xyz, new_xyz, coords_bzyx, features, voxel2point_indices)
        if self.USE_ROW_MAJOR and features.is_cuda and self.pool_method == 'max_pool':
            return self._forward_rows(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, coords_bzyx, features,
                                      voxel2point_indices)
        outs = []
        for grouper, mlp_in, mlp_pos, mlp_out in zip(self.groupers, self.mlps_in, self.mlps_pos,
                                                     self.mlps_out):
            feats = mlp_in(features.t().unsqueeze(0)).squeeze(0).t().contiguous()     # (N, c_mid)
            g_feat, g_xyz, empty = grouper(coords_bzyx, xyz, xyz_batch_cnt, new_xyz,
                                           new_xyz_batch_cnt, feats, voxel2point_indices)
            g_feat[empty] = 0
            rel = g_xyz - new_xyz.unsqueeze(-1)
            rel[empty] = 0
            pos = mlp_pos(rel.permute(1, 0, 2).unsqueeze(0))                           # (1,c,M,ns)
            x = self.relu(g_feat.permute(1, 0, 2).unsqueeze(0) + pos)
            if self.pool_method == 'max_pool':
                x = F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)
            elif self.pool_method == 'avg_pool':
                x = F.avg_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)
            else:
                raise NotImplementedError
            outs.append(mlp_out(x).squeeze(0).t())
        return torch.cat(outs, dim=1)

    # ---- inference fast path: one kernel per scale after the voxel query (csrc/glx_points.hip,
    # k_voxel_pool_agg).  Same arithmetic with the eval-mode BatchNorms folded into the 1x1 convs;
    # the (M, C, ns) grouped tensors are never materialised.
    USE_FUSED = True

    def _fusable(self, features):
        return (self.USE_FUSED and not self.training and not torch.is_grad_enabled()
                and features.is_cuda and features.dtype == torch.float32
                and self.pool_method == 'max_pool'
                and all(m[0].out_channels <= 64 for m in self.mlps_out)
                and all(m[0].out_channels <= 64 for m in self.mlps_in))

    @staticmethod
    def _fold(conv, bn):
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = conv.weight.reshape(conv.out_channels, conv.in_channels) * s[:, None]
        b = bn.bias - bn.running_mean * s
        if conv.bias is not None:
            b = b + conv.bias * s
        return w.float().contiguous(), b.float().contiguous()

    def _folded(self):
        mods = [m for seq in (*self.mlps_in, *self.mlps_pos, *self.mlps_out) for m in (seq[0], seq[1])]
        tensors = [t for m in mods for t in (m.weight, m.bias) if t is not None]
        tensors += [t for m in mods if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d))
                    for t in (m.running_mean, m.running_var)]
        from .... import _lib
        tag = tuple((t._version, t.data_ptr()) for t in tensors) + (_lib.weights_epoch(*tensors),)
        cache = self.__dict__.get("_glx_folded")
        if cache is None or cache[0] != tag:
            with torch.no_grad():
                cache = (tag, [tuple(self._fold(seq[0], seq[1]) for seq in (a, b, c))
                               for a, b, c in zip(self.mlps_in, self.mlps_pos, self.mlps_out)])
            self.__dict__["_glx_folded"] = cache
        return cache[1]

    def _forward_fused(self, xyz, new_xyz, coords_bzyx, features, voxel2point_indices):
        from . import pointnet2_stack_cuda as pointnet2
        m = new_xyz.shape[0]
        xyz, new_xyz = xyz.contiguous(), new_xyz.contiguous()
        widths = [seq[0].out_channels for seq in self.mlps_out]
        out = torch.empty((m, sum(widths)), dtype=torch.float32, device=features.device)
        outs = []
        for grouper, ((w_in, b_in), (w_pos, b_pos), (w_out, b_out)) in zip(self.groupers, self._folded()):
            feats = torch.addmm(b_in, features, w_in.t())                               # (N, c_mid)
            idx = voxel_query_utils.voxel_query_raw(grouper.max_range, grouper.radius, grouper.nsample,
                                                    xyz, new_xyz, coords_bzyx, voxel2point_indices)
            o = out if len(widths) == 1 else torch.empty((m, w_out.shape[0]), dtype=torch.float32,
                                                         device=features.device)
            pointnet2.voxel_pool_agg_wrapper(m, grouper.nsample, w_out.shape[1], w_out.shape[0], feats,
                                             xyz, new_xyz, idx, None, w_pos, b_pos, w_out, b_out, o)
            outs.append(o)
        return out if len(widths) == 1 else torch.cat(outs, dim=1)

    # ---- training path, row-major: the same arithmetic with the 1x1 convolutions written as matrix
    # products on (rows, channels) tensors and the BatchNorms applied to rows.  The reference's
    # (1, C, M, ns) Conv2d / Conv1d formulation sends MIOpen into its worst case on this stack: the
    # weight gradient of a 3 -> 32 channel 1x1 conv over 1.4 M positions ran 40-50 ms per scale
    # (naive / batched-GEMM wrw solvers); as a GEMM it is microseconds.  Parameters, running
    # statistics and results are those of the module path (tested), which stays available
    # (USE_ROW_MAJOR = False) as the statement-by-statement mirror of voxel_pool_modules.py:88-108.
    USE_ROW_MAJOR = True
    SPLITK_MIN_ROWS = 1 << 13

    @staticmethod
    def _linear_rows(x2d, w, bias=None):
        """x2d (rows, C_in) @ w^T.  Tall-skinny products (tens of thousands of rows x <= 64 channels) run as 128
        batched products, so that the weight gradient autograd derives is a batched GEMM + a sum over the batch
        (split-K) instead of one (C_out x C_in) GEMM with K = rows, which hipBLASLt runs on a single 32x32 tile
        (300 us for 60 k rows; 2.6 ms for the 1.4 M rows of the position conv before it was fused away)."""
        rows = x2d.shape[0]
        if rows >= NeighborVoxelSAModuleMSG.SPLITK_MIN_ROWS and rows % 128 == 0 and bias is None:
            if ROWS_LINEAR_FN and x2d.is_cuda and x2d.is_contiguous() and x2d.dtype == torch.float32:
                return _RowsLinearFn.apply(x2d, w)
            return torch.bmm(x2d.view(128, rows // 128, -1), w.t().unsqueeze(0).expand(128, -1, -1)).view(rows, -1)
        return F.linear(x2d, w, bias)

    @staticmethod
    def _conv_bn_rows(seq, x2d):
        """Sequential(Conv(k=1, bias=False), BatchNorm[, ReLU]) on a (rows, C_in) tensor."""
        if rows_conv_bn_supported(seq, x2d):
            return rows_conv_bn(seq, x2d)
        conv = seq[0]
        w = conv.weight.reshape(conv.out_channels, conv.in_channels)
        y = NeighborVoxelSAModuleMSG._linear_rows(x2d, w, conv.bias)
        return NeighborVoxelSAModuleMSG._bn_rows(seq, y)

    @staticmethod
    def _bn_rows(seq, y):
        """The BatchNorm (+ ReLU) of Sequential(Conv, BatchNorm[, ReLU]) on (rows, C) tensors."""
        bn = seq[1]
        from ....spconv import core
        if y.is_cuda and core.can_fuse_train_bn(bn, y):
            # two launches forward, two backward on the (rows, C) matrix (csrc/glx_bn.hip) instead of torch's
            # statistics / transform / reduce / elementwise kernels + a separate ReLU
            return core.fused_train_bn(bn, y, len(seq) > 2, None)
        if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        use_batch = bn.training or not bn.track_running_stats
        y = F.batch_norm(y, bn.running_mean if bn.track_running_stats else None,
                         bn.running_var if bn.track_running_stats else None, bn.weight, bn.bias, use_batch,
                         bn.momentum if bn.momentum is not None else 0.1, bn.eps)
        return F.relu(y) if len(seq) > 2 else y

    def _forward_rows(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, coords_bzyx, features,
                      voxel2point_indices):
        outs = []
        m = new_xyz.shape[0]
        xyz, new_xyz = xyz.contiguous(), new_xyz.contiguous()
        for grouper, mlp_in, mlp_pos, mlp_out in zip(self.groupers, self.mlps_in, self.mlps_pos, self.mlps_out):
            feats = self._conv_bn_rows(mlp_in, features)                                 # (N, c_mid)
            ns = grouper.nsample
            idx = voxel_query_utils.voxel_query_raw(grouper.max_range, grouper.radius, ns, xyz, new_xyz,
                                                    coords_bzyx, voxel2point_indices)    # (M, ns) global rows
            keep = (idx[:, :1] >= 0).to(feats.dtype).view(m, 1, 1)
            g_feat = GroupRows.apply(feats, idx)                                         # (M, ns, c_mid)
            with torch.no_grad():
                rel = (GroupRows.apply(xyz, idx) - new_xyz.view(m, 1, 3)) * keep         # (M, ns, 3)
            pos = self._conv_bn_rows(mlp_pos, rel.view(m * ns, 3))                       # (M*ns, c_mid)
            if self.pool_method == 'max_pool' and m > 0:
                pooled = ReluAddMax.apply(g_feat, pos.view(m, ns, -1))                   # (M, c_mid)
            else:
                x = F.relu(g_feat + pos.view(m, ns, -1))                                 # (M, ns, c_mid)
                pooled = x.max(dim=1)[0] if self.pool_method == 'max_pool' else x.mean(dim=1)
            outs.append(self._conv_bn_rows(mlp_out, pooled))                             # (M, c_out)
        return torch.cat(outs, dim=1)
