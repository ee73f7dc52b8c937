"""The dense (MFMA) side of the hot path: SURVEY.md section 8 rows a21-a23.

`north_star` keeps these on PyTorch-ROCm (MIOpen / hipBLASLt drive the matrix cores); what this
module owns is the *shape* of the work and the parameter names, so that GLENet checkpoints load
and the harness (bench / tests) can run the whole detector data flow:

  BEVBackbone   layer list of BaseBEVBackbone   (pcdet/models/backbones_2d/base_bev_backbone.py:6-112)
  AnchorHead    the three 1x1 convs of AnchorHeadSingle (pcdet/models/dense_heads/anchor_head_single.py:41-58)
  RoIFCStack    shared / cls / reg FC towers of VoxelRCNNHead (pcdet/models/roi_heads/voxelrcnn_head.py:22-66)
  CVAE          PointNet encoders + box decoder of the label-uncertainty generator
                (cvae_uncertainty/point_net.py:10-49, cvae_uncertainty/model.py:33-142,150-265)

Every module is table-driven (one builder, several shapes).  Parity with the reference's own
modules is pinned by tests/golden/dense_path_ref.npz (state dicts + outputs produced by the
reference code on CPU, see tests/golden/make_golden.py).
"""
import contextlib
import ctypes
import math
import os
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from . import conv2d as own_conv


class _ConvSplitBackward(torch.autograd.Function):
    """F.conv2d / F.conv_transpose2d whose backward issues the two gradients as SEPARATE library calls: the input
    gradient on the current stream (it feeds the chain conv <- BatchNorm <- conv <- ...), the weight (+ bias)
    gradient on spconv.core.WGRAD_STREAM when a training step has set it -- a leaf of the backward pass, exactly like
    the sparse convs' weight gradients.  torch's own node computes both on one stream, back to back; here MIOpen's
    weight-gradient kernels (with their zero fills) run beside the next layer's BatchNorm backward, whose statistics
    kernels are short dependent chains that leave the chip mostly idle.  Same library kernels, same values."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, dilation, transposed, output_padding, groups):
        ctx.cfg = (tuple(stride), tuple(padding), tuple(dilation), bool(transposed), tuple(output_padding), int(groups))
        ctx.save_for_backward(x, w)
        ctx.bias_sizes = [int(bias.shape[0])] if bias is not None else None
        return torch.ops.aten.convolution(x, w, bias, *ctx.cfg)

    @staticmethod
    def backward(ctx, gy):
        from .spconv import core
        x, w = ctx.saved_tensors
        stride, padding, dilation, transposed, output_padding, groups = ctx.cfg
        gx = gw = gb = None
        want_w = ctx.needs_input_grad[1] or (ctx.bias_sizes is not None and ctx.needs_input_grad[2])
        if want_w:
            side = core.WGRAD_STREAM
            if side is not None:
                cur = torch.cuda.current_stream(x.device)
                side.wait_stream(cur)
                for t in (x, gy, w):
                    t.record_stream(side)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                _, gw, gb = torch.ops.aten.convolution_backward(
                    gy, x, w, ctx.bias_sizes, stride, padding, dilation, transposed, output_padding, groups,
                    [False, bool(ctx.needs_input_grad[1]), ctx.bias_sizes is not None and bool(ctx.needs_input_grad[2])])
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(gy, x, w, None, stride, padding, dilation, transposed,
                                                     output_padding, groups, [True, False, False])[0]
        return gx, gw, gb, None, None, None, None, None, None


class _OwnStridedForward(torch.autograd.Function):
    """ZeroPad2d(1) + Conv2d(c, 2c, 3, stride 2) of the second BEV block: forward on csrc/glx_deconv2d.hip
    (glx_conv3x3s2_forward, the split-bf16 arithmetic of the stride-1 layers), backward the library's two calls as in
    _ConvSplitBackward.  The point is reproducibility more than speed: MIOpen's forward kernel for this layer sums
    split-K slices with float atomics, so that every activation behind it -- and with it the ReLU masks of the whole
    second block -- changes in the last bit from run to run (tools/forward_determinism.py); an element of a
    pre-activation that sits at zero then takes or drops its whole gradient at random, and two passes over one batch
    differ by up to 1 % in every weight gradient upstream (tools/step_repeat.py)."""

    @staticmethod
    def forward(ctx, x, w, bn=None):
        """bn: the training-mode BatchNorm2d behind the layer -- statistics in the kernel's epilogue
        (glx_conv3x3s2_forward_bn), returns (y, coef, save_mean, save_invstd)."""
        import ctypes
        from ._lib import call
        b, c, h, wd = x.shape
        cout = int(w.shape[0])
        fwd, _ = own_conv.packs(w, strided=True)
        xd = x.detach()
        y = torch.empty((b, cout, h // 2, wd // 2), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        ctx.cfg = ((2, 2), (1, 1), (1, 1), False, (0, 0), 1)
        ctx.bias_sizes = None
        ctx.save_for_backward(x, w)
        if bn is None:
            call("glx_conv3x3s2_forward", xd, b, h, wd, c, fwd, cout, y)
            return y
        from .spconv import core
        stats = tuple(torch.empty(n, dtype=torch.float32, device=x.device) for n in (2 * cout, cout, cout))
        st = _lib.bn_stats(core._bn_state(x.device), bn, *stats)
        call("glx_conv3x3s2_forward_bn", xd, b, h, wd, c, fwd, cout, y, ctypes.byref(st))
        if bn.track_running_stats:
            _lib.bump_weights_epoch((bn.running_mean, bn.running_var))
        ctx.mark_non_differentiable(*stats)
        ctx.set_materialize_grads(False)
        return (y,) + stats

    @staticmethod
    def backward(ctx, gy, *_):
        if gy is None:
            return None, None, None
        x, w = ctx.saved_tensors
        gy = gy.contiguous(memory_format=torch.channels_last)
        if OWN_STRIDED_GRADS and gy.dtype == torch.float32 and w.shape[0] % 64 == 0 and w.shape[1] % 32 == 0:
            # the layer is the stride-1 convolution sampled at the even pixels: its gradients are the stride-1 kernels' on the
            # output gradient spread over the stride-1 map (csrc/glx_deconv2d.hip: glx_spread_stride2) -- no library call
            from ._lib import call
            from .spconv import core
            b, cout, ho, wo = gy.shape
            up = torch.empty((b, cout, 2 * ho, 2 * wo), dtype=torch.float32, device=gy.device, memory_format=torch.channels_last)
            call("glx_spread_stride2", gy, b, ho, wo, cout, up)
            gx = gw = None
            if ctx.needs_input_grad[1]:
                side = core.WGRAD_STREAM
                if side is not None:
                    side.wait_stream(torch.cuda.current_stream(x.device))
                    for t in (x, up, w):
                        t.record_stream(side)
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                    gw = own_conv.wgrad(x.detach(), up, w)
            if ctx.needs_input_grad[0]:
                _, bwd = own_conv.packs(w)
                gx = own_conv._run(up, bwd, int(w.shape[1]))
            return gx, gw, None
        # (x, w, bias, ...) of _ConvSplitBackward: needs_input_grad has two entries here, the helper reads [0], [1] and [2].
        # A plain namespace, not a class: a class object sits in a reference cycle and would keep the saved activation --
        # and the whole graph above it -- alive until the collector runs.
        shim = types.SimpleNamespace(saved_tensors=ctx.saved_tensors, cfg=ctx.cfg, bias_sizes=None,
                                     needs_input_grad=(ctx.needs_input_grad[0], ctx.needs_input_grad[1], False))
        return _ConvSplitBackward.backward(shim, gy.contiguous(memory_format=torch.channels_last))[:2] + (None,)


def _own_strided_ok(x, w, stride, padding, dilation, groups, bias):
    return (OWN_STRIDED_FORWARD and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and bias is None and groups == 1
            and tuple(w.shape[2:]) == (3, 3) and tuple(stride) == (2, 2) and tuple(padding) == (1, 1)
            and tuple(dilation) == (1, 1) and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0 and x.shape[1] == w.shape[1]
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.is_contiguous(memory_format=torch.channels_last))


OWN_STRIDED_FORWARD = True
# ... and its two gradients on the stride-1 kernels over the spread output gradient (no library call; exact like them).  OFF by
# default: three of four pixels of the spread map are zeros that the stride-1 kernels multiply all the same -- 5.23 -> 5.28 ms per
# step against the library's two launches (three alternating runs per arm, profiles/r06_summary.md); set True for a step without
# a vendor convolution.
OWN_STRIDED_GRADS = False
SPLIT_CONV_BACKWARD = True
OWN_CONV3X3 = True     # 3x3 / stride 1 layers on csrc/glx_conv2d.hip
OWN_DECONV = True       # the deblocks' transposed convolutions on csrc/glx_deconv2d.hip
BEV_FIRST_KEY = "bev_first"      # indice_key of the first BEV layer's rule table (spconv.core.PlannedConv)
SPARSE_FIRST_BEV_LAYER = True   # see BEVBackbone._first_layer_sparse
FUSE_BN_IN_CONV3X3 = True    # ... with the next BatchNorm's statistics in the epilogue
# ... and a layer's BatchNorm + ReLU applied ON LOAD by the next 3x3 layer of the block (forward and weight gradient read the
# raw convolution output through scale / shift; the next layer's backward carries this BatchNorm's backward): the normalised
# map of the inner layers of a block is never written (base_bev_backbone.py:36-49)
BN_ON_LOAD = True
STRIDED_BN_STATS = True          # a block's strided layer: statistics in its epilogue
FIRST_LAYER_BN_STATS = True   # first BEV layer (sparse): statistics in its epilogue
HEAD_DGRAD_BN = True        # ... and their backward sums in the head's input gradient
# kernel form of that launch (glx_head1x1_input_grad_bn_form): 1 = a wave owns 64 channels -- 74 us against 109 alone, no
# difference on the recorded step (DESIGN 9.22 viii)
HEAD_DGRAD_FORM = 1
HEAD_BN_ON_LOAD = True    # deblocks' BatchNorm + ReLU applied by the anchor head's kernels
DECONV_BN_STATS = True   # deblocks: BatchNorm statistics in the deconv's epilogue


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else tuple(int(a) for a in v)


def _leaf(t):
    return t is None or t.is_leaf


def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """F.conv2d; in a training step on the device its backward splits over two streams (_ConvSplitBackward).
    Only for LEAF weights: their gradient goes straight to AccumulateGrad (no kernel) and is first read after the
    step has joined the side stream; a derived weight (the anchor head's concatenated filters) hands its gradient to
    another autograd node, which would run on the main stream without waiting for the side stream.
    The 3x3 / stride-1 / pad-1 layers without bias on channels-last maps (every block layer of the BEV backbone but
    the strided one) run on our own kernels (glenet_amd.conv2d), same rule for the weight gradient's stream."""
    if OWN_CONV3X3 and (_leaf(w) or not torch.is_grad_enabled()) and own_conv.supported(
            x, w, _pair(stride), _pair(padding), _pair(dilation), groups, bias):
        return own_conv.conv3x3(x, w)
    if OWN_CONV3X3 and _leaf(w) and _own_strided_ok(x, w, _pair(stride), _pair(padding), _pair(dilation), groups, bias):
        return _OwnStridedForward.apply(x, w)
    if (SPLIT_CONV_BACKWARD and x.is_cuda and torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)
            and _leaf(w) and _leaf(bias)):
        return _ConvSplitBackward.apply(x, w, bias, _pair(stride), _pair(padding), _pair(dilation), False, (0, 0), groups)
    return F.conv2d(x, w, bias, stride, padding, dilation, groups)


def conv_module(m, x):
    """nn.Conv2d / nn.ConvTranspose2d forward through our own kernels or the split-backward node when they apply."""
    if (isinstance(m, nn.Conv2d) and not isinstance(m.padding, str) and getattr(m, "padding_mode", "zeros") == "zeros"
            and OWN_CONV3X3 and x.is_cuda):
        return conv2d(x, m.weight, m.bias, m.stride, m.padding, m.dilation, m.groups)
    if (isinstance(m, nn.ConvTranspose2d) and OWN_DECONV and x.is_cuda and _leaf(m.weight) and own_conv.deconv_supported(
            x, m.weight, _pair(m.stride), _pair(m.padding), _pair(m.output_padding), _pair(m.dilation), m.groups, m.bias)):
        return own_conv.deconv(x, m.weight)
    if (SPLIT_CONV_BACKWARD and x.is_cuda and torch.is_grad_enabled() and (x.requires_grad or m.weight.requires_grad)
            and getattr(m, "padding_mode", "zeros") == "zeros" and not isinstance(m.padding, str)
            and _leaf(m.weight) and _leaf(m.bias)):
        if isinstance(m, nn.ConvTranspose2d):
            return _ConvSplitBackward.apply(x, m.weight, m.bias, _pair(m.stride), _pair(m.padding), _pair(m.dilation),
                                            True, _pair(m.output_padding), m.groups)
        if isinstance(m, nn.Conv2d):
            return _ConvSplitBackward.apply(x, m.weight, m.bias, _pair(m.stride), _pair(m.padding), _pair(m.dilation),
                                            False, (0, 0), m.groups)
    return m(x)


def _bn2d(c):
    return nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)      # base_bev_backbone.py:37


class BEVBackbone(nn.Module):
    """blocks[i] = ZeroPad2d(1) + conv3x3(stride s_i, pad 0) + BN + ReLU + n_i x (conv3x3 + BN + ReLU);
    deblocks[i] = ConvTranspose2d(k = stride = u_i) + BN + ReLU; outputs concatenated on channels."""

    def __init__(self, input_channels, layer_nums=(5, 5), layer_strides=(1, 2), num_filters=(64, 128),
                 upsample_strides=(1, 2), num_upsample_filters=(128, 128)):
        super().__init__()
        assert len(layer_nums) == len(layer_strides) == len(num_filters)
        assert len(upsample_strides) == len(num_upsample_filters)
        self.blocks, self.deblocks = nn.ModuleList(), nn.ModuleList()
        cin = input_channels
        for i, (n, s, c) in enumerate(zip(layer_nums, layer_strides, num_filters)):
            seq = [nn.ZeroPad2d(1), nn.Conv2d(cin, c, 3, stride=s, padding=0, bias=False), _bn2d(c), nn.ReLU()]
            for _ in range(n):
                seq += [nn.Conv2d(c, c, 3, padding=1, bias=False), _bn2d(c), nn.ReLU()]
            self.blocks.append(nn.Sequential(*seq))
            if i < len(upsample_strides):
                u, cu = upsample_strides[i], num_upsample_filters[i]
                if u >= 1:
                    up = nn.ConvTranspose2d(c, cu, int(u), stride=int(u), bias=False)
                else:                                   # fractional stride = strided conv (:57-66)
                    k = int(round(1.0 / u))
                    up = nn.Conv2d(c, cu, k, stride=k, bias=False)
                self.deblocks.append(nn.Sequential(up, _bn2d(cu), nn.ReLU()))
            cin = c
        self.num_bev_features = sum(num_upsample_filters)
        if len(upsample_strides) > len(layer_nums):     # one more deblock on the concatenation (:70-75)
            u, c = int(upsample_strides[-1]), self.num_bev_features
            self.deblocks.append(nn.Sequential(nn.ConvTranspose2d(c, c, u, stride=u, bias=False), _bn2d(c),
                                               nn.ReLU()))

    head_on_load = False      # GLENetVR: leave the concatenation to the anchor head (_Head1x1Parts) in training steps

    @staticmethod
    def concat_parts(parts):
        """The concatenated map of `spatial_features_2d_parts` (what _ups_fused would have returned)."""
        from .spconv import core
        args = []
        for rows, coef, mean, invstd, bn in parts["parts"]:
            args += [rows, coef, mean, invstd, bn.weight, bn.bias]
        b, h, w = parts["shape"]
        y = core.FusedBNApplyCat.apply(True, *args)
        return y.view(b, h, w, y.shape[1]).permute(0, 3, 1, 2)

    def _ups_fused(self, raw):
        """torch.cat of the upsampled maps (:100-104) with each deblock's BatchNorm2d + ReLU writing its channel
        block of the concatenated channels-last map directly (no 144 MB copy forward, no slice copies backward);
        raw: the deblocks' convolution outputs.  None when the fused kernels do not cover the case.  With head_on_load
        (two parts, statistics taken by the transposed convolutions) the parts themselves are returned as a dict."""
        from .spconv import core
        if len(raw) < 2 or len(self.deblocks) != len(raw):
            return None
        with_stats = [isinstance(u, tuple) for u in raw]
        maps = [u[0] if t else u for u, t in zip(raw, with_stats)]
        bns = []
        for blk, u in zip(self.deblocks, maps):
            mods = list(blk)
            if not (len(mods) == 3 and isinstance(mods[2], nn.ReLU) and self._can_fuse_bn(mods[1], u)
                    and u.shape[0] == maps[0].shape[0] and u.shape[2:] == maps[0].shape[2:]):
                return None
            bns.append(mods[1])
        b, _, h, w = maps[0].shape
        rows = [u.permute(0, 2, 3, 1).reshape(b * h * w, u.shape[1]) for u in maps]
        if all(with_stats) and self.head_on_load and len(raw) == 2 and torch.is_grad_enabled() and HEAD_BN_ON_LOAD \
                and len(self.deblocks) == len(self.blocks):
            return dict(shape=(b, h, w), parts=[(r_.contiguous(), u[1], u[2], u[3], bn) for r_, u, bn in zip(rows, raw, bns)])
        if all(with_stats):       # the transposed convolutions took the statistics: one transform launch per part
            args = []
            for r_, u, bn in zip(rows, raw, bns):
                args += [r_, u[1], u[2], u[3], bn.weight, bn.bias]
            y = core.FusedBNApplyCat.apply(True, *args)
        elif any(with_stats):
            return None
        else:
            y = core.fused_train_bn_cat(bns, rows, True)
        return y.view(b, h, w, y.shape[1]).permute(0, 3, 1, 2)

    FUSE_UPS_CAT = True

    def _first_layer_sparse(self, st):
        """blocks[0]'s ZeroPad2d(1) + Conv2d(C * D -> c, 3) applied to the SPARSE tensor the BEV map is the dense
        image of (HeightCompression left it to us): with the depth folded into channels (channel c * D + z,
        height_compression.py:21-25) the 3x3 convolution over (y, x) IS a sparse convolution with kernel (D, 3, 3),
        stride (D, 1, 1), padding (0, 1, 1) -- the same sums without the zeros.  ~16 % of the BEV cells of a KITTI
        frame are occupied: a ninth to a sixth of the dense layer's products in forward, input gradient (needed at
        the occupied cells only) and weight gradient, and the 144 MB dense map is never built.  Returns the conv's
        output as a channels-last (B, c, H, W) map, or None when the layer is not of that form."""
        from .spconv import core
        mods = list(self.blocks[0])
        if not (SPARSE_FIRST_BEV_LAYER and len(mods) >= 2 and isinstance(mods[0], nn.ZeroPad2d)
                and tuple(mods[0].padding) == (1, 1, 1, 1) and isinstance(mods[1], nn.Conv2d)):
            return None
        conv = mods[1]
        meta = st.features_meta()
        c, d = int(meta.shape[1]), int(st.spatial_shape[0])
        if not (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.dilation == (1, 1)
                and conv.groups == 1 and conv.bias is None and conv.in_channels == c * d and meta.is_cuda
                and 9 * d <= 27 and c in (16, 32, 64, 128) and conv.out_channels in (16, 32, 64, 128)
                and meta.shape[0] > 0 and st._index is not None):
            return None
        cout = conv.out_channels
        # (cout, c * D + z, ky, kx) -> (z, ky, kx, c, cout): views up to the final reshape
        w = conv.weight.permute(2, 3, 1, 0).unflatten(2, (c, d)).permute(3, 0, 1, 2, 4).reshape(9 * d, c, cout)
        geom = ((d, 3, 3), (d, 1, 1), (0, 1, 1))
        rs = st.indice_dict.get(BEV_FIRST_KEY) if st.indice_dict is not None else None
        if rs is not None and (tuple(tuple(g) for g in rs.geom[1:]) != geom or rs.in_indices is not st.indices):
            rs = None
        if rs is None:                                  # not planned with the backbone's tables: build it here
            rs = core.build_strided_rules(st, *geom)
        elif rs.ready is not None:                      # built on the plan stream
            torch.cuda.current_stream(st.features.device).wait_event(rs.ready)
        bn = mods[2] if len(mods) > 3 and isinstance(mods[3], nn.ReLU) else None
        stats = None
        # the producer (conv_out) may have left relu(bn(raw)) pending: take the raw rows, transform on load
        pend = st._pending if (st._features is None and core.BN_ON_LOAD and core.BN_BWD_IN_DGRAD and core.USE_BN_STATE
                               and core.USE_PAIR_LISTS and torch.is_grad_enabled() and not (c >= 128 and cout >= 128)
                               and _lib.query("glx_sconv_packed_bytes", 9 * d, c, cout)) else None
        x_in = pend.raw if pend is not None else st.features
        pre_args = (None,) * 6 if pend is None else (pend.coef, pend.mean, pend.invstd, pend.bn.weight, pend.bn.bias, pend.count)
        if (FIRST_LAYER_BN_STATS and bn is not None and self._bn_fusable(bn) and bn.num_features == cout and core.USE_BN_STATE
                and core.FUSE_BN_STATS_IN_CONV and torch.is_grad_enabled() and not (c >= 128 and cout >= 128)
                and rs.out_spatial_shape[0] == 1):
            # the layer's BatchNorm2d counts every pixel of the dense map; the cells the sparse convolution does not store
            # are zeros: the statistics are the rows' sums over B * H * W elements -- taken in the convolution's epilogue
            pixels = int(st.batch_size) * int(rs.out_spatial_shape[1]) * int(rs.out_spatial_shape[2])
            feats, *stats = core.SparseConvFunction.apply(x_in, w, None, rs, False, None, False, (bn, pixels), None, None, *pre_args)
            own_conv._count_batch(bn)
        else:
            feats = core.SparseConvFunction.apply(x_in, w, None, rs, False, None, False, None, None, None, *pre_args)
        out = core.SparseConvTensor(feats, rs.out_indices, rs.out_spatial_shape, st.batch_size, st.grid, st.voxel_num,
                                    st.indice_dict, st.benchmark, rs.count_out)
        out._index = rs.out_index
        dense = out.dense_bev()
        return dense if stats is None else (dense,) + tuple(stats) + (bn,)

    # ---- inference: every eval-mode BatchNorm2d (+ ReLU) folded into the epilogue of the convolution in front of it,
    # the deblocks write their slices of the concatenated map, the first layer runs on the sparse tensor
    FUSE_EVAL = True

    def _eval_plan(self, data_dict):
        """[(kind, conv, bn)] per block + the deblocks, or None when the module is not of the reference's form."""
        if not (self.FUSE_EVAL and OWN_CONV3X3 and OWN_DECONV and not self.training and not torch.is_grad_enabled()
                and len(self.deblocks) == len(self.blocks)):
            return None
        plan = []
        for blk, up in zip(self.blocks, self.deblocks):
            mods, layers = list(blk), []
            if not (len(mods) >= 4 and (len(mods) - 1) % 3 == 0 and isinstance(mods[0], nn.ZeroPad2d)
                    and tuple(mods[0].padding) == (1, 1, 1, 1)):
                return None
            for k in range(1, len(mods), 3):
                conv, bn, act = mods[k], mods[k + 1], mods[k + 2]
                first = k == 1
                if not (isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d) and isinstance(act, nn.ReLU)
                        and conv.kernel_size == (3, 3) and conv.bias is None and conv.groups == 1 and conv.dilation == (1, 1)
                        and conv.padding == ((0, 0) if first else (1, 1)) and bn.track_running_stats
                        and conv.stride in (((1, 1), (2, 2)) if first else ((1, 1),))
                        and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0 and conv.weight.is_cuda):
                    return None
                layers.append((conv, bn))
            um = list(up)
            if not (len(um) == 3 and isinstance(um[0], nn.ConvTranspose2d) and isinstance(um[1], nn.BatchNorm2d)
                    and isinstance(um[2], nn.ReLU) and um[0].bias is None and um[0].kernel_size == um[0].stride
                    and um[0].kernel_size in ((1, 1), (2, 2)) and um[0].padding == (0, 0) and um[0].output_padding == (0, 0)
                    and um[0].groups == 1 and um[0].in_channels % 64 == 0 and um[0].out_channels % 64 == 0
                    and um[1].track_running_stats):
                return None
            plan.append((layers, um[0], um[1]))
        return plan

    def _forward_eval(self, data_dict, plan):
        from .spconv import core
        x = data_dict.get("spatial_features")
        st = data_dict.get("encoded_spconv_tensor")
        ups = []
        for i, (layers, upconv, upbn) in enumerate(plan):
            for k, (conv, bn) in enumerate(layers):
                scale, shift = core._bn_affine(bn)
                if i == 0 and k == 0 and x is None:
                    x = self._first_layer_sparse_eval(st, conv, scale, shift)
                    if x is None:
                        x = data_dict["spatial_features"] = st.dense_bev()
                    else:
                        continue
                if conv.stride == (2, 2):
                    if x.shape[2] % 2 or x.shape[3] % 2:
                        return None
                    x = own_conv.conv3x3s2_affine(x, conv.weight, scale, shift, True)
                else:
                    x = own_conv.conv3x3_affine(x, conv.weight, scale, shift, True)
            ups.append((x, upconv, upbn))
        h0 = int(st.spatial_shape[1]) if data_dict.get("spatial_features") is None else int(data_dict["spatial_features"].shape[2])
        for x, _, _ in ups:
            data_dict["spatial_features_%dx" % int(h0 / x.shape[2])] = x
        sizes = {(x.shape[2] * int(u.stride[0]), x.shape[3] * int(u.stride[1])) for x, u, _ in ups}
        if len(sizes) != 1:
            return None
        (ho, wo), = sizes
        b = ups[0][0].shape[0]
        total = sum(u.out_channels for _, u, _ in ups)
        cat = torch.empty((b, total, ho, wo), dtype=torch.float32, device=ups[0][0].device, memory_format=torch.channels_last)
        off = 0
        for x, upconv, upbn in ups:
            scale, shift = core._bn_affine(upbn)
            own_conv.deconv_affine(x, upconv.weight, scale, shift, True, out=cat, channel_offset=off)
            off += upconv.out_channels
        data_dict["spatial_features_2d"] = cat
        return data_dict

    def _first_layer_sparse_eval(self, st, conv, scale, shift):
        """Inference twin of _first_layer_sparse: the sparse conv, the dense image of its result, then the folded
        BatchNorm + ReLU on the dense map (cells without an occupied neighbour hold relu(shift), not zero)."""
        from .spconv import core
        if st is None or not SPARSE_FIRST_BEV_LAYER:
            return None
        c, d = int(st.features.shape[1]), int(st.spatial_shape[0])
        if not (conv.stride == (1, 1) and conv.in_channels == c * d and 9 * d <= 27 and c in (16, 32, 64, 128)
                and conv.out_channels in (16, 32, 64, 128) and st.features.shape[0] > 0 and st._index is not None):
            return None
        cout = conv.out_channels
        w = conv.weight.detach().permute(2, 3, 1, 0).unflatten(2, (c, d)).permute(3, 0, 1, 2, 4).reshape(9 * d, c, cout).contiguous()
        geom = ((d, 3, 3), (d, 1, 1), (0, 1, 1))
        rs = st.indice_dict.get(BEV_FIRST_KEY) if st.indice_dict is not None else None
        if rs is not None and (tuple(tuple(g) for g in rs.geom[1:]) != geom or rs.in_indices is not st.indices):
            rs = None
        if rs is None:
            rs = core.build_strided_rules(st, *geom)
        elif rs.ready is not None:
            torch.cuda.current_stream(st.features.device).wait_event(rs.ready)
        feats = core._sconv(st.features.contiguous().float(), w, None, rs.nbr, rs.tile_order_out, rs.N_out, rules=rs,
                            n_live=rs.count_out)
        out = core.SparseConvTensor(feats, rs.out_indices, rs.out_spatial_shape, st.batch_size, st.grid, st.voxel_num,
                                    st.indice_dict, st.benchmark, rs.count_out)
        out._index = rs.out_index
        raw = out.dense_bev()
        b, ch, h, wd = raw.shape
        rows = raw.permute(0, 2, 3, 1).reshape(b * h * wd, ch)
        y = torch.empty_like(rows)
        _lib.call("glx_bn_apply_forward", rows, torch.cat([scale, shift]), 1, b * h * wd, ch, None, y, 0)
        return y.view(b, h, wd, ch).permute(0, 3, 1, 2)

    convert_input = False      # dropin.accelerate(): an NCHW `spatial_features` map is converted to channels-last on entry

    def forward(self, data_dict):
        x_in = data_dict.get("spatial_features")
        if (self.convert_input and x_in is not None and x_in.is_cuda and x_in.dim() == 4
                and not x_in.is_contiguous(memory_format=torch.channels_last)):
            data_dict["spatial_features"] = x_in.contiguous(memory_format=torch.channels_last)
        plan = self._eval_plan(data_dict)
        if plan is not None:
            x_in = data_dict.get("spatial_features")
            if x_in is None or (x_in.is_cuda and x_in.is_contiguous(memory_format=torch.channels_last)):
                keys = set(data_dict.keys())
                out = self._forward_eval(data_dict, plan)
                if out is not None:
                    return out
                for k in set(data_dict.keys()) - keys:          # a shape the fused path does not cover: module by module
                    del data_dict[k]
        x0 = data_dict.get("spatial_features")
        first = None
        if x0 is None:                                  # HeightCompression(defer=True): the map is ours to make
            st = data_dict["encoded_spconv_tensor"]
            first = self._first_layer_sparse(st) if torch.is_grad_enabled() else None
            if first is None:
                x0 = data_dict["spatial_features"] = st.dense_bev()
            h0 = int(st.spatial_shape[1])
        else:
            h0 = int(x0.shape[2])
        x, ups, raw = x0, [], []
        fuse = self.FUSE_UPS_CAT and len(self.deblocks) == len(self.blocks) and len(self.blocks) > 1
        for i, blk in enumerate(self.blocks):
            if i == 0 and isinstance(first, tuple):       # the first layer's statistics are taken: its BatchNorm + ReLU is pending
                x = self._run_block(blk, first[0], 4, pending=first)
            else:
                x = self._run_block(blk, first, 2) if (i == 0 and first is not None) else self._run_block(blk, x)
            data_dict["spatial_features_%dx" % int(h0 / x.shape[2])] = x
            if fuse:
                up, ubn = self.deblocks[i][0], (self.deblocks[i][1] if len(self.deblocks[i]) > 1 else None)
                if (DECONV_BN_STATS and OWN_DECONV and isinstance(up, nn.ConvTranspose2d) and self._bn_fusable(ubn)
                        and own_conv.bn_state_available() and ubn.num_features == up.out_channels and _leaf(up.weight)
                        and own_conv.deconv_supported(x, up.weight, _pair(up.stride), _pair(up.padding),
                                                      _pair(up.output_padding), _pair(up.dilation), up.groups, up.bias)):
                    raw.append(own_conv.deconv_bn_raw(x, up.weight, ubn))      # (y, coef, mean, invstd): statistics in the epilogue
                    continue
                raw.append(conv_module(self.deblocks[i][0], x))   # the deblock's (transposed) convolution only
            else:
                ups.append(self._run_block(self.deblocks[i], x) if len(self.deblocks) > 0 else x)
        if fuse:
            x = self._ups_fused(raw)
            if isinstance(x, dict):                         # the anchor head reads the parts (head_on_load)
                data_dict["spatial_features_2d"] = None
                data_dict["spatial_features_2d_parts"] = x
                return data_dict
            if x is None:                                   # not covered: BatchNorm + ReLU per map, then concatenate
                ups = []
                for i, u in enumerate(raw):
                    mods = list(self.deblocks[i])
                    if isinstance(u, tuple):                # statistics already taken (and counted) by the deconv
                        y = own_conv.bn_apply(u + (mods[1],), len(mods) > 2 and isinstance(mods[2], nn.ReLU))
                        ups.append(self._run_block(nn.Sequential(*mods[3:]), y) if len(mods) > 3 else y)
                    else:
                        ups.append(self._run_block(nn.Sequential(*mods[1:]), u))
        if not fuse or x is None:
            x = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        if len(self.deblocks) > len(self.blocks):
            x = self._run_block(self.deblocks[-1], x)
        data_dict["spatial_features_2d"] = x
        return data_dict

    @staticmethod
    def _fused_bn_relu(bn, x, relu):
        """Training-mode BatchNorm2d (+ ReLU) on a channels-last map through the fused row kernels of the sparse
        backbone (csrc/glx_bn.hip): a channels-last (B, C, H, W) tensor IS a row-major (B*H*W, C) matrix.  Four
        launches forward + backward instead of MIOpen's six NHWC kernels + two ReLU passes; same statistics
        semantics as nn.BatchNorm2d (tests/test_sparse_gpu.py::test_fused_train_batchnorm_matches_torch)."""
        from .spconv import core
        b, c, h, w = x.shape
        rows = x.permute(0, 2, 3, 1).reshape(b * h * w, c)            # a view of channels-last memory
        y = core.fused_train_bn(bn, rows, relu, None)
        return y.view(b, h, w, c).permute(0, 3, 1, 2)

    @staticmethod
    def _bn_fusable(bn):
        from .spconv import core
        return (isinstance(bn, nn.BatchNorm2d) and bn.training and torch.is_grad_enabled()
                and core.USE_FUSED_TRAIN_BN and bn.affine and bn.momentum is not None
                and bn.num_features % 4 == 0 and bn.num_features <= 512 and 1024 % bn.num_features == 0)

    @staticmethod
    def _can_fuse_bn(bn, x):
        return (BEVBackbone._bn_fusable(bn) and x.is_cuda and x.dim() == 4
                and x.is_contiguous(memory_format=torch.channels_last))

    @staticmethod
    def _own_bn_conv(conv, bn, x):
        """conv: a 3x3 / stride-1 / pad-1 Conv2d on the own kernels whose training-mode BatchNorm `bn` rides in its epilogue
        -- the layers that can read their input through the previous layer's BatchNorm + ReLU."""
        return (FUSE_BN_IN_CONV3X3 and OWN_CONV3X3 and not isinstance(conv.padding, str) and conv.padding_mode == "zeros"
                and _pair(conv.padding) == (1, 1) and _pair(conv.stride) == (1, 1) and conv.kernel_size == (3, 3)
                and _leaf(conv.weight) and BEVBackbone._bn_fusable(bn) and own_conv.bn_state_available() and x.is_cuda
                and bn.num_features == conv.out_channels and conv.in_channels == x.shape[1]
                and own_conv.supported(x, conv.weight, (1, 1), (1, 1), _pair(conv.dilation), conv.groups, conv.bias))

    @staticmethod
    def _run_block(blk, x, start=0, pending=None):
        """nn.Sequential semantics with ZeroPad2d(1) + Conv2d(k=3, padding=0) run as ONE convolution with
        padding=1: the same sums over the same zeros, without materialising the padded copy of the input (77 us
        forward + 55 us backward for the 144 MB BEV map) -- module list and parameter names stay the reference's."""
        mods = list(blk)
        i = start
        # pending: (y, coef, mean, invstd, bn) of a conv whose BatchNorm + ReLU the NEXT conv applies on load
        while i < len(mods):
            m = mods[i]
            conv, step = None, 1
            if pending is not None and not (isinstance(m, nn.Conv2d) and i + 1 < len(mods)
                                            and BEVBackbone._own_bn_conv(m, mods[i + 1], x)):
                x, pending = own_conv.bn_apply(pending, True), None         # not the layer that was expected: materialise
            if (isinstance(m, nn.ZeroPad2d) and tuple(m.padding) == (1, 1, 1, 1) and i + 1 < len(mods)
                    and isinstance(mods[i + 1], nn.Conv2d) and mods[i + 1].padding == (0, 0)
                    and mods[i + 1].kernel_size == (3, 3) and mods[i + 1].dilation == (1, 1)):
                conv, pad, step = mods[i + 1], (1, 1), 2
            elif isinstance(m, nn.Conv2d) and not isinstance(m.padding, str) and m.padding_mode == "zeros":
                conv, pad = m, _pair(m.padding)
            if conv is not None:
                # Conv2d -> BatchNorm2d (-> ReLU) on the own 3x3 kernels: the statistics ride in the conv's epilogue
                j = i + step
                bn = mods[j] if j < len(mods) else None
                if (FUSE_BN_IN_CONV3X3 and OWN_CONV3X3 and _leaf(conv.weight) and BEVBackbone._bn_fusable(bn)
                        and own_conv.bn_state_available()
                        and x.is_cuda and bn.num_features == conv.out_channels and own_conv.supported(
                            x, conv.weight, _pair(conv.stride), pad, _pair(conv.dilation), conv.groups, conv.bias)):
                    relu = j + 1 < len(mods) and isinstance(mods[j + 1], nn.ReLU)
                    raw = own_conv.conv3x3_bn_raw(x, conv.weight, bn, pending)
                    pending = None
                    nxt = j + (2 if relu else 1)
                    # BatchNorm + ReLU of this layer on load in the next own convolution of the block (no normalised map)
                    if (BN_ON_LOAD and relu and nxt + 1 < len(mods) and isinstance(mods[nxt], nn.Conv2d)
                            and BEVBackbone._own_bn_conv(mods[nxt], mods[nxt + 1], raw[0])):
                        pending, x = raw, raw[0]
                    else:
                        x = own_conv.bn_apply(raw, relu)
                    i = nxt
                    continue
                if (STRIDED_BN_STATS and BN_ON_LOAD and torch.is_grad_enabled() and _leaf(conv.weight) and BEVBackbone._bn_fusable(bn)
                        and own_conv.bn_state_available() and bn.num_features == conv.out_channels
                        and _own_strided_ok(x, conv.weight, _pair(conv.stride), pad, _pair(conv.dilation), conv.groups, conv.bias)
                        and j + 2 < len(mods) and isinstance(mods[j + 1], nn.ReLU) and isinstance(mods[j + 2], nn.Conv2d)
                        and j + 3 < len(mods)):
                    # the strided layer of a block: statistics in its epilogue, BatchNorm + ReLU on load in the next layer
                    raw = _OwnStridedForward.apply(x, conv.weight, bn)
                    if BEVBackbone._own_bn_conv(mods[j + 2], mods[j + 3], raw[0]):
                        own_conv._count_batch(bn)
                        pending, x = raw + (bn,), raw[0]
                        i = j + 2
                        continue
                    own_conv._count_batch(bn)
                    x = own_conv.bn_apply(raw + (bn,), True)
                    i = j + 2
                    continue
                x = conv2d(x, conv.weight, conv.bias, conv.stride, pad, conv.dilation, conv.groups)
                i += step
                continue
            if BEVBackbone._can_fuse_bn(m, x):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = BEVBackbone._fused_bn_relu(m, x, relu)
                i += 2 if relu else 1
                continue
            x = conv_module(m, x) if isinstance(m, nn.ConvTranspose2d) else m(x)
            i += 1
        if pending is not None:
            x = own_conv.bn_apply(pending, True)
        return x

    @staticmethod
    def flops_per_frame(h, w, input_channels=256, layer_nums=(5, 5), layer_strides=(1, 2),
                        num_filters=(64, 128), upsample_strides=(1, 2), num_upsample_filters=(128, 128)):
        """Multiply-add x2 of one forward pass."""
        total, cin = 0, input_channels
        for n, s, c, u, cu in zip(layer_nums, layer_strides, num_filters, upsample_strides, num_upsample_filters):
            h, w = h // s, w // s
            total += 2 * h * w * 9 * cin * c + n * 2 * h * w * 9 * c * c
            total += 2 * h * w * c * cu * int(u) * int(u)
            cin = c
        return total


class _Head1x1(torch.autograd.Function):
    """The anchor head's 1x1 convolutions + permute(0, 2, 3, 1).contiguous() on a channels-last map as one launch per
    direction (csrc/glx_head.hip): x (B, C, H, W) channels-last, (weight (n, C, 1, 1), bias (n)) per head ->
    (B, H, W, n) predictions per head."""

    @staticmethod
    def forward(ctx, x, *wb):
        import ctypes
        from ._lib import call
        ws, bs = wb[0::2], wb[1::2]
        b, c, h, w = x.shape
        M = b * h * w
        outs = [torch.empty((b, h, w, wt.shape[0]), dtype=torch.float32, device=x.device) for wt in ws]
        P3, I3 = ctypes.c_void_p * 3, ctypes.c_int32 * 3
        pad = lambda seq: list(seq) + [None] * (3 - len(seq))                                   # noqa: E731
        ptr = lambda seq: P3(*[t.data_ptr() if t is not None else None for t in pad(seq)])      # noqa: E731
        ctx.n = I3(*[int(wt.shape[0]) for wt in ws] + [0] * (3 - len(ws)))
        ws2 = [wt.detach().reshape(wt.shape[0], c).contiguous() for wt in ws]
        call("glx_head1x1_forward", x, ctypes.c_int64(M), c, ptr(ws2), ptr([t.detach() for t in bs]), ctx.n, ptr(outs))
        ctx.save_for_backward(x, *ws2)
        ctx.shape = (b, c, h, w, M)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        from ._lib import call, query, size_arg
        from .spconv import core
        x, ws2 = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        b, c, h, w, M = ctx.shape
        P3 = ctypes.c_void_p * 3
        pad = lambda seq: list(seq) + [None] * (3 - len(seq))                                   # noqa: E731
        ptr = lambda seq: P3(*[t.data_ptr() if t is not None else None for t in pad(seq)])      # noqa: E731
        gs = [g.contiguous() for g in grads]
        gx = None
        gws = [torch.empty((wt.shape[0], c, 1, 1), dtype=torch.float32, device=x.device) for wt in ws2]
        gbs = [torch.empty(wt.shape[0], dtype=torch.float32, device=x.device) for wt in ws2]
        side = core.WGRAD_STREAM
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(x.device))
            for t in [x] + gs + gws + gbs:
                t.record_stream(side)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            nb = query("glx_head1x1_wgrad_workspace_bytes", c)
            wsp = _lib.workspace.get(nb, x.device)
            call("glx_head1x1_weight_grad", ptr(gs), x, ctypes.c_int64(M), c, ctx.n, ptr(gws), ptr(gbs), wsp, size_arg(nb))
        if ctx.needs_input_grad[0]:
            gx = torch.empty((b, c, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
            call("glx_head1x1_input_grad", ptr(gs), ctypes.c_int64(M), c, ptr(list(ws2)), ctx.n, gx)
        out = [gx]
        for gw_, gb_ in zip(gws, gbs):
            out += [gw_, gb_]
        return tuple(out)


class _Head1x1Parts(torch.autograd.Function):
    """_Head1x1 reading the map as the two deblocks' RAW outputs r_p (M, c_p) and applying their training-mode BatchNorm +
    ReLU on load (coef_p = scale | shift from the transposed convolution's epilogue, glx_head1x1_forward_parts): the
    concatenated 144 MB map is neither written nor read back.  backward: the weight gradient transforms on load too, the
    input gradient (M, C) goes through both BatchNorms' backward here (what FusedBNApplyCat.backward did)."""

    @staticmethod
    def forward(ctx, shape, r0, coef0, mean0, invstd0, g0, b0, r1, coef1, mean1, invstd1, g1, b1, *wb):
        import ctypes
        from ._lib import call
        ws, bs = wb[0::2], wb[1::2]
        b, h, w = shape
        M, c0, c1 = b * h * w, r0.shape[1], r1.shape[1]
        C = c0 + c1
        outs = [torch.empty((b, h, w, wt.shape[0]), dtype=torch.float32, device=r0.device) for wt in ws]
        P3, I3 = ctypes.c_void_p * 3, ctypes.c_int32 * 3
        pad = lambda seq: list(seq) + [None] * (3 - len(seq))                                   # noqa: E731
        ptr = lambda seq: P3(*[t.data_ptr() if t is not None else None for t in pad(seq)])      # noqa: E731
        ctx.n = I3(*[int(wt.shape[0]) for wt in ws] + [0] * (3 - len(ws)))
        ws2 = [wt.detach().reshape(wt.shape[0], C).contiguous() for wt in ws]
        call("glx_head1x1_forward_parts", r0, r1, c0, coef0, coef1, ctypes.c_int64(M), C, ptr(ws2), ptr([t.detach() for t in bs]),
             ctx.n, ptr(outs))
        ctx.save_for_backward(r0, coef0, mean0, invstd0, g0, b0, r1, coef1, mean1, invstd1, g1, b1, *ws2)
        ctx.dims = (M, c0, c1)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        from ._lib import call, query, size_arg
        from .spconv import core
        r0, coef0, mean0, invstd0, g0, b0, r1, coef1, mean1, invstd1, g1, b1 = ctx.saved_tensors[:12]
        ws2 = ctx.saved_tensors[12:]
        M, c0, c1 = ctx.dims
        C, dev = c0 + c1, r0.device
        P3 = ctypes.c_void_p * 3
        pad = lambda seq: list(seq) + [None] * (3 - len(seq))                                   # noqa: E731
        ptr = lambda seq: P3(*[t.data_ptr() if t is not None else None for t in pad(seq)])      # noqa: E731
        gs = [g.contiguous() for g in grads]
        gws = [torch.empty((wt.shape[0], C, 1, 1), dtype=torch.float32, device=dev) for wt in ws2]
        gbs = [torch.empty(wt.shape[0], dtype=torch.float32, device=dev) for wt in ws2]
        side = core.WGRAD_STREAM
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(dev))
            for t in [r0, r1, coef0, coef1] + gs + gws + gbs:
                t.record_stream(side)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            nb = query("glx_head1x1_wgrad_workspace_bytes", C)
            wsp = _lib.workspace.get(nb, dev)
            call("glx_head1x1_weight_grad_parts", ptr(gs), r0, r1, c0, coef0, coef1, ctypes.c_int64(M), C, ctx.n, ptr(gws),
                 ptr(gbs), wsp, size_arg(nb))
        out = [None]
        parts = ((r0, coef0, mean0, invstd0, g0, b0, c0), (r1, coef1, mean1, invstd1, g1, b1, c1))
        if HEAD_DGRAD_BN and C == 256 and core.USE_BN_STATE:
            # the input-gradient launch masks with the two ReLUs and takes both BatchNorm backwards' sums (no statistics pass
            # over the 144 MB map); what is left per part is the transform
            state = core._bn_state(dev)
            dzs, bns, res = [], [], []
            for r_, coef, mean, invstd, gam, bet, c in parts:
                dz = torch.empty_like(r_)
                coef3 = torch.empty(3 * c, dtype=torch.float32, device=dev)
                dgamma = torch.empty(c, dtype=torch.float32, device=dev)
                dbeta = torch.empty(c, dtype=torch.float32, device=dev)
                bns.append(_lib.BnBwdStats(*[_lib._p(t) for t in (state, None, coef, mean, invstd, gam, coef3, dgamma, dbeta)]))
                dzs.append(dz)
                res.append((coef3, dgamma, dbeta))
            call("glx_head1x1_input_grad_bn_form", ptr(gs), ctypes.c_int64(M), C, ptr(list(ws2)), ctx.n, r0, r1, c0,
                 ctypes.byref(bns[0]), ctypes.byref(bns[1]), dzs[0], dzs[1], int(HEAD_DGRAD_FORM))
            for (r_, coef, mean, invstd, gam, bet, c), dz, (coef3, dgamma, dbeta) in zip(parts, dzs, res):
                dx = torch.empty_like(r_)
                call("glx_bn_backward_apply", r_, dz, coef3, mean, invstd, M, c, None, dx)
                out += [dx, None, None, None, dgamma, dbeta]
        else:
            gx = torch.empty((M, C), dtype=torch.float32, device=dev)
            call("glx_head1x1_input_grad", ptr(gs), ctypes.c_int64(M), C, ptr(list(ws2)), ctx.n, gx)
            col = 0
            for r_, coef, mean, invstd, gam, bet, c in parts:
                dx = torch.empty_like(r_)
                dgamma = torch.empty(c, dtype=torch.float32, device=dev)
                dbeta = torch.empty(c, dtype=torch.float32, device=dev)
                wsb = core.workspace.get(query("glx_bn_workspace_bytes", c), dev)
                call("glx_bn_relu_backward", r_, gx[:, col:], None, M, c, gam, bet, mean, invstd, 1, dx, dgamma, dbeta, None, wsb,
                     size_arg(wsb.numel()), core._bn_state(dev), C)
                out += [dx, None, None, None, dgamma, dbeta]
                col += c
        for gw_, gb_ in zip(gws, gbs):
            out += [gw_, gb_]
        return tuple(out)


class AnchorHead(nn.Module):
    """conv_cls / conv_box / conv_dir_cls, 1x1, with AnchorHeadSingle's bias init (:60-62)."""

    def __init__(self, input_channels, num_class=3, num_anchors_per_location=6, code_size=7, num_dir_bins=2):
        super().__init__()
        self.num_class, self.code_size = num_class, code_size
        self.conv_cls = nn.Conv2d(input_channels, num_anchors_per_location * num_class, 1)
        self.conv_box = nn.Conv2d(input_channels, num_anchors_per_location * code_size, 1)
        self.conv_dir_cls = (nn.Conv2d(input_channels, num_anchors_per_location * num_dir_bins, 1)
                             if num_dir_bins else None)
        nn.init.constant_(self.conv_cls.bias, -math.log((1 - 0.01) / 0.01))
        nn.init.normal_(self.conv_box.weight, mean=0, std=0.001)

    # Training: the three 1x1 heads read the same (B, 256, H, W) map -- 144 MB at the KITTI size.  Run separately,
    # backward computes three input gradients of that size and adds them (two 60 us passes over 3 x 144 MB) next
    # to three weight-gradient convolutions with their zero fills; as ONE convolution over the concatenated
    # filters (2 + 14 + 4 output channels) the map is read once per direction and its gradient is written once.
    # Parameters (and state-dict keys) stay conv_cls / conv_box / conv_dir_cls; the concatenation is differentiable.
    FUSE_HEADS = True

    def _forward_fused(self, data_dict, x):
        convs = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
        if torch.is_grad_enabled():
            w = torch.cat([c.weight for c in convs], dim=0)
            b = torch.cat([c.bias for c in convs], dim=0)
        else:                                   # inference: the concatenated filters are cached until the weights move
            tag = ((_lib.weights_epoch(*[c.weight for c in convs], *[c.bias for c in convs]),)
                   + tuple(c.weight._version for c in convs) + tuple(c.bias._version for c in convs))
            hit = self.__dict__.get("_glx_fused_heads")
            if hit is None or hit[0] != tag or hit[1].device != x.device:
                w = torch.cat([c.weight.detach() for c in convs], dim=0)
                if x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
                    w = w.contiguous(memory_format=torch.channels_last)
                hit = (tag, w, torch.cat([c.bias.detach() for c in convs], dim=0))
                self.__dict__["_glx_fused_heads"] = hit
            w, b = hit[1], hit[2]
        y = conv2d(x, w, b).permute(0, 2, 3, 1)                                           # (B,H,W,sum Cout)
        parts = y.split([c.out_channels for c in convs], dim=-1)
        data_dict["cls_preds"] = parts[0].contiguous()
        data_dict["box_preds"] = parts[1].contiguous()
        if self.conv_dir_cls is not None:
            data_dict["dir_cls_preds"] = parts[2].contiguous()
        return data_dict

    OWN_HEAD = True

    def _own(self, x):
        convs = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
        return (self.OWN_HEAD and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % 64 == 0 and 64 <= x.shape[1] <= 512
                and all(c.kernel_size == (1, 1) and c.stride == (1, 1) and c.padding == (0, 0) and c.groups == 1
                        and c.bias is not None for c in convs)
                and sum(c.out_channels for c in convs) <= 32)

    def _own_parts(self, parts):
        convs = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
        rows = [p[0] for p in parts["parts"]]
        c = sum(r.shape[1] for r in rows)
        return (self.OWN_HEAD and len(rows) == 2 and rows[0].is_cuda and c % 64 == 0 and 64 <= c <= 512
                and rows[0].shape[1] % 16 == 0 and all(r.is_contiguous() and r.dtype == torch.float32 for r in rows)
                and all(cv.kernel_size == (1, 1) and cv.stride == (1, 1) and cv.padding == (0, 0) and cv.groups == 1
                        and cv.bias is not None and cv.in_channels == c for cv in convs)
                and sum(cv.out_channels for cv in convs) <= 32)

    def forward(self, data_dict):
        x = data_dict.get("spatial_features_2d")
        parts = data_dict.get("spatial_features_2d_parts") if x is None else None
        if parts is not None and self._own_parts(parts):
            convs = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
            args = []
            for rows, coef, mean, invstd, bn in parts["parts"]:
                args += [rows, coef, mean, invstd, bn.weight, bn.bias]
            outs = _Head1x1Parts.apply(parts["shape"], *args, *[t for c in convs for t in (c.weight, c.bias)])
            data_dict["cls_preds"], data_dict["box_preds"] = outs[0], outs[1]
            if self.conv_dir_cls is not None:
                data_dict["dir_cls_preds"] = outs[2]
            return data_dict
        if x is None:                       # the parts were left for us but this head cannot take them: concatenate after all
            x = data_dict["spatial_features_2d"] = BEVBackbone.concat_parts(parts)
        if self._own(x):
            convs = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
            outs = _Head1x1.apply(x, *[t for c in convs for t in (c.weight, c.bias)])
            data_dict["cls_preds"], data_dict["box_preds"] = outs[0], outs[1]
            if self.conv_dir_cls is not None:
                data_dict["dir_cls_preds"] = outs[2]
            return data_dict
        if (self.FUSE_HEADS and self.conv_cls.bias is not None and self.conv_box.bias is not None
                and (self.conv_dir_cls is None or self.conv_dir_cls.bias is not None)
                and all(c.kernel_size == (1, 1) for c in (self.conv_cls, self.conv_box))):
            return self._forward_fused(data_dict, x)
        data_dict["cls_preds"] = self.conv_cls(x).permute(0, 2, 3, 1).contiguous()      # (B,H,W,A*cls)
        data_dict["box_preds"] = self.conv_box(x).permute(0, 2, 3, 1).contiguous()
        if self.conv_dir_cls is not None:
            data_dict["dir_cls_preds"] = self.conv_dir_cls(x).permute(0, 2, 3, 1).contiguous()
        return data_dict


OWN_WIDE_LINEAR = True      # the 20 736-wide first RoI Linear on csrc/glx_rows.hip (exact fp32 MFMA products) instead of library GEMMs


def _wide_linear_ok(x, w):
    """The shapes glx_linear_wide_* / glx_linear_wgrad_multi take: dense fp32 device matrices, a long K, few rows."""
    return (OWN_WIDE_LINEAR and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 2 and w.dim() == 2
            and x.is_contiguous() and w.is_contiguous() and x.shape[1] == w.shape[1] and x.shape[1] >= 4096 and x.shape[1] % 64 == 0
            and w.shape[0] % 64 == 0 and 1 <= x.shape[0] <= 4096)


def wide_linear_forward(x, w):
    rows, k = x.shape
    n = w.shape[0]
    y = torch.empty((rows, n), dtype=torch.float32, device=x.device)
    nb = _lib.query("glx_linear_wide_workspace_bytes", rows, n, k)
    ws = _lib.workspace.get(nb, x.device)
    _lib.call("glx_linear_wide_forward", x, w, y, rows, n, k, ws, _lib.size_arg(nb))
    return y


def wide_linear_input_grad(gy, w):
    rows, n = gy.shape
    k = w.shape[1]
    gx = torch.empty((rows, k), dtype=torch.float32, device=gy.device)
    _lib.call("glx_linear_wide_input_grad", gy, w, gx, rows, n, k)
    return gx


class _SplitKLinearFn(torch.autograd.Function):
    """y = x @ W^T for the RoI FC towers (a few hundred rows) with the products hipBLASLt's heuristics starve split
    along K into batched GEMMs:
      * forward with a very long K (20 736 -> 256: one 256x16-tile kernel, 210 us) as `split` partial products;
      * the weight gradient dW = dY^T @ X, whose output is one 256 x 256 tile with K = rows (ONE workgroup, 117 us
        per layer, five layers) as row-chunk partial products summed afterwards (~10 us)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        k = x.shape[1]
        if _wide_linear_ok(x, w):
            return wide_linear_forward(x, w)
        split = next((s_ for s_ in (32, 27, 24, 16, 12, 8) if k % s_ == 0), 0) if (k >= 4096 and x.shape[0] <= 4096) else 0
        if split:
            xs = x.view(x.shape[0], split, k // split).transpose(0, 1)               # (S, R, k/S) view
            ws = w.view(w.shape[0], split, k // split).permute(1, 2, 0)              # (S, k/S, out) view
            return torch.bmm(xs, ws).sum(0)
        return x @ w.t()

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = wide_linear_input_grad(gy, w) if _wide_linear_ok(x, w) else gy @ w
        if ctx.needs_input_grad[1]:
            if DEFERRED_FC_WGRADS is not None and w.is_leaf:
                # a leaf of the backward pass: a staged training step computes it later, on the main stream's idle time,
                # instead of on the RoI branch (its critical path) -- see run_deferred_fc_wgrads
                DEFERRED_FC_WGRADS.append((x, gy, w, _deferred_event(x.device), _SplitKLinearFn.weight_grad))
            else:
                gw = _SplitKLinearFn.weight_grad(x, gy, w)
        return gx, gw

    @staticmethod
    def weight_grad(x, gy, w, out=None):
        """out: where to write the gradient (the optimizer's flat-buffer view of w: no gather copy afterwards)."""
        rows = x.shape[0]
        chunks = next((c for c in (16, 8, 4, 2) if rows % c == 0 and rows // c >= 16), 0) \
            if w.shape[0] * w.shape[1] <= 256 * 1024 else 0
        if out is not None and not (out.shape == w.shape and out.is_contiguous() and out.dtype == torch.float32):
            out = None
        if _wide_linear_ok(x, w) and gy.is_contiguous() and gy.dtype == torch.float32:
            # dW (N, K) = gy^T x on the towers' weight-gradient kernel (one job): 4 x 324 blocks of 64 x 64
            dst = out if out is not None else torch.empty_like(w)
            ptr = lambda t: (ctypes.c_void_p * 1)(t.data_ptr())
            _lib.call("glx_linear_wgrad_multi", 1, ptr(x), ptr(gy), ptr(dst), rows, int(w.shape[1]), int(w.shape[0]))
            return dst
        if chunks:
            parts = torch.bmm(gy.view(chunks, rows // chunks, -1).transpose(1, 2), x.view(chunks, rows // chunks, -1))
            return parts.sum(0) if out is None else torch.sum(parts, 0, out=out)
        return gy.t() @ x if out is None else torch.mm(gy.t(), x, out=out)


DEFERRED_FC_WGRADS = None     # a list while a staged backward collects the FC towers' weight-gradient jobs
GRADS_IN_PLACE = True      # FC weight gradients straight into the optimizer's buffer
DEFERRED_FC_SAME_STREAM = False   # the collector will run the jobs on the stream that creates them: no events


def _deferred_event(device):
    if DEFERRED_FC_SAME_STREAM:
        return None
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    return ev


FC_WGRADS_GROUPED = True   # the towers' 256 x 256 weight gradients in one launch


def _grad_target(w):
    """Where a deferred weight gradient of `w` is written: the optimizer's flat-buffer view when the parameter has no gradient
    yet this step (lent once per step: the stamp _lib.grad_buffer keeps), else None (a fresh tensor, added)."""
    gen = _lib.grad_generation_of(w)
    view = getattr(w, "_glx_grad_view", None) if (w.grad is None and GRADS_IN_PLACE
                                                  and getattr(w, "_glx_grad_lent", -1) != gen) else None
    if view is not None:
        w._glx_grad_lent = gen
    return view


def run_deferred_fc_wgrads(jobs):
    """The weight gradients _SplitKLinearFn.backward left out, on the current stream (which waits for the event each
    job recorded where its gy became available); written into the parameters' .grad like AccumulateGrad would.  Jobs of one
    small shape (the towers' five 256 x 256 filters over the same RoI rows) run as ONE launch (csrc/glx_rows.hip,
    glx_linear_wgrad_multi) instead of a batched library GEMM + a sum each."""
    import ctypes
    cur = torch.cuda.current_stream()
    with torch.no_grad():
        for x, gy, w, ev, weight_grad in jobs:
            if ev is not None:
                cur.wait_event(ev)
                x.record_stream(cur)
                gy.record_stream(cur)
        groups, rest = {}, []
        for job in jobs:
            x, gy, w = job[0], job[1], job[2]
            small = (FC_WGRADS_GROUPED and w.dim() == 2 and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0
                     and w.shape[0] * w.shape[1] <= 512 * 512 and x.is_cuda and x.dtype == gy.dtype == w.dtype == torch.float32
                     and x.is_contiguous() and gy.is_contiguous() and x.shape == (gy.shape[0], w.shape[1])
                     and gy.shape[1] == w.shape[0] and x.shape[0] >= 1)
            if small:
                groups.setdefault((tuple(w.shape), x.shape[0]), []).append(job)
            else:
                rest.append(job)
        for (shape, rows), group in groups.items():
            for lo in range(0, len(group), 8):
                part = group[lo:lo + 8]
                outs = []
                for x, gy, w, ev, weight_grad in part:
                    view = _grad_target(w)
                    outs.append(view if view is not None and view.is_contiguous() and view.shape == w.shape
                                else torch.empty_like(w, memory_format=torch.contiguous_format))
                ptrs = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
                _lib.call("glx_linear_wgrad_multi", len(part), ptrs([j[0] for j in part]), ptrs([j[1] for j in part]), ptrs(outs),
                          rows, shape[1], shape[0])
                for (x, gy, w, ev, weight_grad), gw in zip(part, outs):
                    w.grad = gw if w.grad is None else w.grad + gw
        for x, gy, w, ev, weight_grad in rest:
            view = _grad_target(w)
            gw = weight_grad(x, gy, w, view) if view is not None else weight_grad(x, gy, w)
            w.grad = gw if w.grad is None else w.grad + gw


class SplitKLinear(nn.Linear):
    """nn.Linear (same parameters, same state-dict keys) whose CUDA training path is _SplitKLinearFn."""

    def forward(self, x):
        if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and torch.is_grad_enabled() and x.is_contiguous():
            y = _SplitKLinearFn.apply(x, self.weight)
            return y if self.bias is None else y + self.bias
        return super().forward(x)


def bn_relu(bn, x):
    """relu(bn(x)) for (N, C) activations; a training-mode BatchNorm1d the fused kernels cover runs as ONE launch
    forward and one backward (csrc/glx_bn.hip -- the short-matrix kernels for the few hundred RoI rows) instead of
    torch's five + three."""
    from .spconv import core
    if x.dim() == 2 and torch.is_grad_enabled() and core.can_fuse_train_bn(bn, x):
        return core.fused_train_bn(bn, x, True)
    return torch.relu(bn(x))


class FCTower(nn.Sequential):
    """Linear -> BatchNorm1d -> ReLU (-> Dropout) stack of the RoI head (voxelrcnn_head.py:40-66); module indices,
    hence parameter names, are nn.Sequential's.  Training: each BatchNorm1d + ReLU pair is one fused launch."""

    def forward(self, x):
        from .spconv import core
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if (isinstance(m, nn.BatchNorm1d) and x.dim() == 2 and torch.is_grad_enabled()
                    and core.can_fuse_train_bn(m, x)):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = core.fused_train_bn(m, x, relu)
                i += 2 if relu else 1
                continue
            x = m(x)
            i += 1
        return x


# ---- the towers behind the first Linear as one launch per direction (csrc/glx_fctower.hip)
FC_TOWER_FUSED = True
FC_TOWER_COOPERATIVE = True     # 0: one launch per phase (5 forward, 4 backward)
_FCT_SUPPORT = {}


def _fct_barrier(head, device):
    """The grid barrier's counters of ONE head module on ONE stream: zero once, every launch leaves them zero.  Owned by the
    module (not a process-global cache): the graph of a pipeline and an eager step of another model never share them.
    Allocated OUTSIDE stream capture only -- a buffer born inside a capture would live in the graph's private pool while
    this cache kept handing it out; the pipelines' warm-up pass creates it before they record."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    owned = head.__dict__.setdefault("_glx_fct_barriers", {})
    b = owned.get(key)
    if b is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("fc tower: the grid-barrier counters of this head do not exist yet on the capturing stream; "
                               "run one eager (warm-up) pass on that stream before recording it")
        b = owned[key] = torch.zeros(32, dtype=torch.int32, device=device)
    return b


def fc_tower_support(device, rows):
    """(supported, cooperative) of glx_fc_tower_supported on `device`, cached per (device, rows)."""
    key = (device.index, int(rows))
    hit = _FCT_SUPPORT.get(key)
    if hit is None:
        coop = ctypes.c_int(0)
        with torch.cuda.device(device):
            ok = _lib.load().glx_fc_tower_supported(int(rows), ctypes.byref(coop))
        hit = _FCT_SUPPORT[key] = (bool(ok), bool(coop.value))
    return hit


def fc_tower_barrier_gave_up(head):
    """True when a one-launch tower of `head` gave up at a grid barrier since the last call (host sync; diagnostics)."""
    bad = False
    for b in head.__dict__.get("_glx_fct_barriers", {}).values():
        flag = ctypes.c_int(0)
        _lib.call_nostream("glx_fc_tower_barrier_status", b, ctypes.byref(flag))
        bad = bad or bool(flag.value)
    return bad


def _fc_bn(bn, mean, invstd):
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    return _lib.FcBn(_lib._p(bn.weight), _lib._p(bn.bias), _lib._p(rm), _lib._p(rv), _lib._p(mean), _lib._p(invstd), float(bn.eps),
                     float(bn.momentum))


def fc_tower_layers(head):
    """(Linear, BatchNorm1d) of layers 0..5 of a VoxelRCNNKLHead-shaped module, or None when the towers are not the
    Linear(bias=False) + BatchNorm1d + ReLU (+ Dropout behind the first layer) x 2 stacks of width 256 the kernel covers."""
    pairs, ps = [], set()
    for seq in (head.shared_fc_layer, head.cls_fc_layers, head.reg_fc_layers):
        mods = list(seq)
        kinds = [type(m) for m in mods]
        with_drop = len(mods) == 7
        want = [nn.Linear, nn.BatchNorm1d, nn.ReLU] + ([nn.Dropout] if with_drop else []) + [nn.Linear, nn.BatchNorm1d, nn.ReLU]
        if len(kinds) != len(want) or not all(issubclass(k, w) for k, w in zip(kinds, want)):
            return None
        ps.add(float(mods[3].p) if with_drop else 0.0)
        b = 4 if with_drop else 3
        for lin, bn in ((mods[0], mods[1]), (mods[b], mods[b + 1])):
            if lin.bias is not None:
                return None
            pairs.append((lin, bn))
    if len(ps) != 1 or any(l.out_features != 256 for l, _ in pairs) or any(l.in_features != 256 for l, _ in pairs[1:]):
        return None
    return pairs


def fc_tower_usable(head, x):
    """Training-mode heads() of a VoxelRCNNKLHead on the fused kernels: every BatchNorm in training mode with affine
    parameters and a momentum, 7 box codes, one class, the rows a multiple of 16 up to 1024."""
    if not (FC_TOWER_FUSED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and torch.is_grad_enabled() and head.training):
        return False
    if x.shape[0] % 16 or not 16 <= x.shape[0] <= 1024:
        return False
    if not fc_tower_support(x.device, x.shape[0])[0]:       # the phases' LDS does not fit this device: module-by-module path
        return False
    pairs = head.__dict__.get("_glx_fct_pairs", 0)
    if pairs == 0:
        pairs = head.__dict__["_glx_fct_pairs"] = fc_tower_layers(head)
    if pairs is None:
        return False
    bns = [bn for _, bn in pairs] + [head.reg_std_bn, head.reg_std_bn1]
    if not all(bn.training and bn.affine and bn.momentum is not None for bn in bns):
        return False
    return (head.cls_pred_layer.out_features == 1 and head.reg_pred_layer.out_features == 7
            and head.reg_std_layer.out_features == 7 and head.reg_std_fc1.out_features == 64 and head.reg_std_fc2.out_features == 1)


class FCTowersFn(torch.autograd.Function):
    """(ori_cls (R,1), std_logit (R,1), rcnn_reg (R,7), rcnn_reg_std (R,7)) of VoxelRCNNKLHead.heads in training mode:
    the first Linear as the split-K library product, everything behind it in glx_fc_tower_forward; backward =
    glx_fc_tower_backward + the Linear weight gradients (deferred like _SplitKLinearFn's when a staged step collects
    them) + the pooled features' gradient."""

    @staticmethod
    def forward(ctx, head, drop_p, x, w0, *params):
        pairs = head._glx_fct_pairs
        R, dev = x.shape[0], x.device
        with torch.no_grad():
            z0 = _SplitKLinearFn.forward(_NoCtx(), x, w0)
        f32 = dict(dtype=torch.float32, device=dev)
        u = None
        if drop_p > 0:
            # fixed_dropout_draws: (3, R, 256) uniforms for layers 0 / 2 / 4 instead of fresh ones (tests replaying a reference run)
            u = getattr(head, "fixed_dropout_draws", None)
            u = torch.rand((3, R, 256), **f32) if u is None else u.to(**f32).contiguous()
        zh = torch.empty((11, R, 256), **f32)                 # z[1..5], h[0..5]
        stats = torch.empty((12, 256), **f32)
        small = torch.empty((2, 7 + 64), **f32)
        dense = torch.empty((2, R), **f32)                    # ori_cls, std_logit
        reg = torch.empty((2, R, 7), **f32)                   # rcnn_reg, rcnn_reg_std
        scratch = _lib.workspace.get(_lib.query("glx_fc_tower_scratch_bytes", R), dev)
        t = _lib.FcTower()
        t.R, t.drop_p, t.drop_u, t.z0 = R, float(drop_p), _lib._p(u), z0.data_ptr()
        for l, (lin, bn) in enumerate(pairs):
            t.w[l] = lin.weight.data_ptr() if l else None
            t.bn[l] = _fc_bn(bn, stats[2 * l], stats[2 * l + 1])
            t.z[l] = zh[l - 1].data_ptr() if l else None
            t.h[l] = zh[5 + l].data_ptr()
        t.w_cls, t.b_cls = head.cls_pred_layer.weight.data_ptr(), head.cls_pred_layer.bias.data_ptr()
        t.w_reg, t.b_reg = head.reg_pred_layer.weight.data_ptr(), head.reg_pred_layer.bias.data_ptr()
        t.w_std, t.b_std = head.reg_std_layer.weight.data_ptr(), head.reg_std_layer.bias.data_ptr()
        t.bn_s7 = _fc_bn(head.reg_std_bn, small[0, :7], small[1, :7])
        t.w_fc1, t.b_fc1 = head.reg_std_fc1.weight.data_ptr(), head.reg_std_fc1.bias.data_ptr()
        t.bn_s64 = _fc_bn(head.reg_std_bn1, small[0, 7:], small[1, 7:])
        t.w_fc2, t.b_fc2 = head.reg_std_fc2.weight.data_ptr(), head.reg_std_fc2.bias.data_ptr()
        t.ori_cls, t.std_logit = dense[0].data_ptr(), dense[1].data_ptr()
        t.rcnn_reg, t.rcnn_reg_std = reg[0].data_ptr(), reg[1].data_ptr()
        t.scratch, t.barrier = scratch.data_ptr(), _fct_barrier(head, dev).data_ptr()
        t.cooperative = 1 if FC_TOWER_COOPERATIVE else 0
        _lib.call("glx_fc_tower_forward", ctypes.byref(t))
        from .spconv import core
        written = []
        for bn in [b for _, b in pairs] + [head.reg_std_bn, head.reg_std_bn1]:
            if bn.track_running_stats:
                written += [bn.running_mean, bn.running_var]
                if bn.num_batches_tracked is not None:
                    if core.DEFERRED_COUNTERS is not None:
                        core.DEFERRED_COUNTERS.append(bn.num_batches_tracked)
                    else:
                        bn.num_batches_tracked += 1
        if written:
            _lib.bump_weights_epoch(written)
        ctx.head, ctx.drop_p = head, drop_p
        ctx.keep = (x, w0, z0, u, zh, stats, small, reg, dense)
        ctx.n_params = len(params)
        return dense[0].view(R, 1), dense[1].view(R, 1), reg[0], reg[1]

    @staticmethod
    def backward(ctx, g_cls, g_logit, g_reg, g_std):
        head, drop_p = ctx.head, ctx.drop_p
        x, w0, z0, u, zh, stats, small, reg, dense = ctx.keep
        pairs = head._glx_fct_pairs
        R, dev = x.shape[0], x.device
        f32 = dict(dtype=torch.float32, device=dev)
        cont = lambda g_: None if g_ is None else g_.contiguous().float()
        g_cls, g_logit, g_reg, g_std = cont(g_cls), cont(g_logit), cont(g_reg), cont(g_std)
        scratch = _lib.workspace.get(_lib.query("glx_fc_tower_scratch_bytes", R), dev)
        t = _lib.FcTower()
        t.R, t.drop_p, t.drop_u, t.z0 = R, float(drop_p), _lib._p(u), z0.data_ptr()
        for l, (lin, bn) in enumerate(pairs):
            t.w[l] = lin.weight.data_ptr() if l else None
            t.bn[l] = _fc_bn(bn, stats[2 * l], stats[2 * l + 1])
            t.z[l] = zh[l - 1].data_ptr() if l else None
            t.h[l] = zh[5 + l].data_ptr()
        t.w_cls, t.b_cls = head.cls_pred_layer.weight.data_ptr(), head.cls_pred_layer.bias.data_ptr()
        t.w_reg, t.b_reg = head.reg_pred_layer.weight.data_ptr(), head.reg_pred_layer.bias.data_ptr()
        t.w_std, t.b_std = head.reg_std_layer.weight.data_ptr(), head.reg_std_layer.bias.data_ptr()
        t.bn_s7 = _fc_bn(head.reg_std_bn, small[0, :7], small[1, :7])
        t.w_fc1, t.b_fc1 = head.reg_std_fc1.weight.data_ptr(), head.reg_std_fc1.bias.data_ptr()
        t.bn_s64 = _fc_bn(head.reg_std_bn1, small[0, 7:], small[1, 7:])
        t.w_fc2, t.b_fc2 = head.reg_std_fc2.weight.data_ptr(), head.reg_std_fc2.bias.data_ptr()
        t.ori_cls, t.std_logit = dense[0].data_ptr(), dense[1].data_ptr()
        t.rcnn_reg, t.rcnn_reg_std = reg[0].data_ptr(), reg[1].data_ptr()
        t.scratch, t.barrier = scratch.data_ptr(), _fct_barrier(head, dev).data_ptr()
        t.cooperative = 1 if FC_TOWER_COOPERATIVE else 0
        dz = torch.empty((6, R, 256), **f32)
        dgb = torch.empty((12, 256), **f32)
        dw_cls, db_cls = torch.empty((1, 256), **f32), torch.empty(1, **f32)
        dw_rs, db_rs = torch.empty((2, 7, 256), **f32), torch.empty((2, 7), **f32)
        d7, d64 = torch.empty((2, 7), **f32), torch.empty((2, 64), **f32)
        dw_fc1, db_fc1 = torch.empty((64, 7), **f32), torch.empty(64, **f32)
        dw_fc2, db_fc2 = torch.empty((1, 64), **f32), torch.empty(1, **f32)
        g = _lib.FcTowerGrads()
        g.g_cls, g.g_logit, g.g_reg, g.g_std = _lib._p(g_cls), _lib._p(g_logit), _lib._p(g_reg), _lib._p(g_std)
        for l in range(6):
            g.dz[l], g.dgamma[l], g.dbeta[l] = dz[l].data_ptr(), dgb[2 * l].data_ptr(), dgb[2 * l + 1].data_ptr()
        g.dw_cls, g.db_cls = dw_cls.data_ptr(), db_cls.data_ptr()
        g.dw_reg, g.db_reg, g.dw_std, g.db_std = dw_rs[0].data_ptr(), db_rs[0].data_ptr(), dw_rs[1].data_ptr(), db_rs[1].data_ptr()
        g.dgamma7, g.dbeta7, g.dgamma64, g.dbeta64 = d7[0].data_ptr(), d7[1].data_ptr(), d64[0].data_ptr(), d64[1].data_ptr()
        g.dw_fc1, g.db_fc1, g.dw_fc2, g.db_fc2 = dw_fc1.data_ptr(), db_fc1.data_ptr(), dw_fc2.data_ptr(), db_fc2.data_ptr()
        g.scratch = scratch.data_ptr()
        _lib.call("glx_fc_tower_backward", ctypes.byref(t), ctypes.byref(g))
        gx = None
        if ctx.needs_input_grad[2]:
            gx = wide_linear_input_grad(dz[0], w0) if _wide_linear_ok(x, w0) else dz[0] @ w0
        # the Linear weight gradients: leaves, deferred when a staged step collects them
        ins = [x] + [zh[5 + l] for l in range(5)]              # layer l's input: pooled, h0, h1, h2, h1, h4
        ins[4] = zh[5 + 1]
        wgrads = []
        for l in range(6):
            w = w0 if l == 0 else pairs[l][0].weight
            if DEFERRED_FC_WGRADS is not None and w.is_leaf:
                DEFERRED_FC_WGRADS.append((ins[l], dz[l], w, _deferred_event(dev), _SplitKLinearFn.weight_grad))
                wgrads.append(None)
            else:
                wgrads.append(_SplitKLinearFn.weight_grad(ins[l], dz[l], w))
        grads = [None, None, gx, wgrads[0]]
        for l in range(6):
            if l:
                grads.append(wgrads[l])
            grads += [dgb[2 * l], dgb[2 * l + 1]]
        grads += [dw_cls, db_cls, dw_rs[0], db_rs[0], dw_rs[1], db_rs[1], d7[0], d7[1], dw_fc1, db_fc1, d64[0], d64[1], dw_fc2, db_fc2]
        assert len(grads) == 4 + ctx.n_params, (len(grads), ctx.n_params)
        return tuple(grads)


class _NoCtx:
    def save_for_backward(self, *a):
        pass


def fc_towers(head, x):
    """heads() of a VoxelRCNNKLHead-shaped module on the fused kernels (fc_tower_usable says when)."""
    pairs = head._glx_fct_pairs
    drops = [m for m in head.shared_fc_layer if isinstance(m, nn.Dropout)]
    drop_p = float(drops[0].p) if drops else 0.0
    params = [pairs[0][1].weight, pairs[0][1].bias]
    for lin, bn in pairs[1:]:
        params += [lin.weight, bn.weight, bn.bias]
    params += [head.cls_pred_layer.weight, head.cls_pred_layer.bias, head.reg_pred_layer.weight, head.reg_pred_layer.bias,
               head.reg_std_layer.weight, head.reg_std_layer.bias, head.reg_std_bn.weight, head.reg_std_bn.bias,
               head.reg_std_fc1.weight, head.reg_std_fc1.bias, head.reg_std_bn1.weight, head.reg_std_bn1.bias,
               head.reg_std_fc2.weight, head.reg_std_fc2.bias]
    return FCTowersFn.apply(head, drop_p, x.contiguous(), pairs[0][0].weight, *params)


def _fc_tower(cin, widths, dp_ratio):
    layers = []
    for k, w in enumerate(widths):
        layers += [SplitKLinear(cin, w, bias=False), nn.BatchNorm1d(w), nn.ReLU(inplace=True)]
        cin = w
        if k != len(widths) - 1 and dp_ratio > 0:
            layers.append(nn.Dropout(dp_ratio))
    return FCTower(*layers), cin


class RoIFCStack(nn.Module):
    """shared_fc_layer -> (cls_fc_layers -> cls_pred_layer, reg_fc_layers -> reg_pred_layer)."""

    def __init__(self, pooled_channels, grid_size=6, shared_fc=(256, 256), cls_fc=(256, 256),
                 reg_fc=(256, 256), dp_ratio=0.3, num_class=1, code_size=7):
        super().__init__()
        pre = grid_size ** 3 * pooled_channels
        self.shared_fc_layer, pre = _fc_tower(pre, shared_fc, dp_ratio)
        self.cls_fc_layers, c = _fc_tower(pre, cls_fc, dp_ratio)
        self.cls_pred_layer = nn.Linear(c, num_class, bias=True)
        self.reg_fc_layers, c = _fc_tower(pre, reg_fc, dp_ratio)
        self.reg_pred_layer = nn.Linear(c, code_size * num_class, bias=True)

    def forward(self, pooled):
        """pooled (R, G^3, C) or (R, G^3*C) -> rcnn_cls (R, num_class), rcnn_reg (R, code*num_class)."""
        x = pooled.reshape(pooled.shape[0], -1)
        if self.USE_FUSED and x.is_cuda and not self.training and not torch.is_grad_enabled() \
                and x.dtype == torch.float32:
            return self._forward_folded(x)
        shared = self.shared_fc_layer(x)
        return self.cls_pred_layer(self.cls_fc_layers(shared)), self.reg_pred_layer(self.reg_fc_layers(shared))

    # ---- inference: eval-mode BatchNorm folded into each Linear (one library GEMM + ReLU per layer
    # instead of GEMM + 4 normalisation kernels), and the first layer -- (R, 20736) x (20736, 256)
    # with R of a few hundred rows, which hipBLASLt runs on ~32 workgroups -- split along K into a
    # batched GEMM whose partial sums are reduced afterwards (~1 000 workgroups).
    USE_FUSED = True
    SPLIT_K_MIN = 4096

    def _folded(self):
        towers = (self.shared_fc_layer, self.cls_fc_layers, self.reg_fc_layers)
        pairs = [(seq[i], seq[i + 1]) for seq in towers for i in range(len(seq))
                 if isinstance(seq[i], nn.Linear)]
        tensors = [t for lin, bn in pairs for t in (lin.weight, bn.weight, bn.bias, bn.running_mean,
                                                    bn.running_var)]
        tag = tuple((t._version, t.data_ptr()) for t in tensors) + (_lib.weights_epoch(*tensors),)
        cache = self.__dict__.get("_glx_folded")
        if cache is None or cache[0] != tag:
            with torch.no_grad():
                folded = []
                for lin, bn in pairs:
                    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    folded.append(((lin.weight * s[:, None]).float().contiguous(),
                                   (bn.bias - bn.running_mean * s).float().contiguous()))
            n = [sum(isinstance(m, nn.Linear) for m in seq) for seq in towers]
            cache = (tag, (folded[:n[0]], folded[n[0]:n[0] + n[1]], folded[n[0] + n[1]:]))
            self.__dict__["_glx_folded"] = cache
        return cache[1]

    @classmethod
    def _affine_relu(cls, x, w, b):
        k = x.shape[1]
        split = next((s for s in (32, 27, 24, 16, 12, 8) if k % s == 0), 0) \
            if (k >= cls.SPLIT_K_MIN and x.shape[0] <= 4096) else 0
        if split:
            xs = x.view(x.shape[0], split, k // split).transpose(0, 1)              # (S, R, k/S) view
            ws = w.view(w.shape[0], split, k // split).permute(1, 2, 0)             # (S, k/S, out) view
            y = torch.bmm(xs, ws).sum(0).add_(b)
        else:
            y = torch.addmm(b, x, w.t())
        return torch.relu_(y)

    def _forward_folded(self, x):
        shared_w, cls_w, reg_w = self._folded()
        x = x.contiguous()
        for w, b in shared_w:
            x = self._affine_relu(x, w, b)
        c = r = x
        for w, b in cls_w:
            c = self._affine_relu(c, w, b)
        for w, b in reg_w:
            r = self._affine_relu(r, w, b)
        return self.cls_pred_layer(c), self.reg_pred_layer(r)


# ------------------------------------------------------------------------------ CVAE
class PointFeat(nn.Module):
    """Shared point MLP (Conv1d k=1 = GEMM) + max over points; no ReLU after the last BN
    (point_net.py:22-28).  widths (64,128,512) = PointNetfeat(x=1), (8,8,8) = SimPointNetfeat(x=0.5)."""

    def __init__(self, pts_dim, widths=(64, 128, 512)):
        super().__init__()
        self.output_channel = widths[2]
        self.conv1 = nn.Conv1d(pts_dim, widths[0], 1)
        self.conv2 = nn.Conv1d(widths[0], widths[1], 1)
        self.conv3 = nn.Conv1d(widths[1], widths[2], 1)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm1d(widths[0]), nn.BatchNorm1d(widths[1]), nn.BatchNorm1d(widths[2])

    def forward(self, x):                       # x (B, pts_dim, P)
        if self._fusable(x):
            return self._forward_fused(x)
        if self._narrow_trainable(x):
            return NarrowFeatTrain.apply(x, self.conv1.weight, self.conv1.bias, self.bn1.weight, self.bn1.bias, self.conv2.weight,
                                         self.conv2.bias, self.bn2.weight, self.bn2.bias, self.conv3.weight, self.conv3.bias,
                                         self.bn3.weight, self.bn3.bias, (self.bn1, self.bn2, self.bn3))
        if self._rows_trainable(x):
            return self._forward_train_rows(x)
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.bn3(self.conv3(x))
        return x.max(dim=2)[0]

    # ---- training on the device: points as rows.  Conv1d(k=1) on (B, C, P) is a GEMM on the (B*P, C) row matrix and
    # BatchNorm1d over (B, C, P) is BatchNorm over those rows: the (B, C, P) layout makes MIOpen run batched
    # strided GEMMs and four-pass NCL BatchNorms; as rows every layer is ONE hipBLASLt GEMM (MFMA) + the fused
    # statistics / transform kernels of csrc/glx_bn.hip (two launches forward, two backward, ReLU included), and the
    # final max over the points of an object is a reduction over P consecutive rows.
    # the 8-wide extractor's training pass without intermediate tensors (csrc/glx_narrowfeat.hip): ten launches forward, nine backward,
    # every pass reads the points only; False: layer by layer on the row kernels (_forward_train_rows)
    NARROW_FUSED_TRAIN = True

    def _narrow_trainable(self, x):
        bns = (self.bn1, self.bn2, self.bn3)
        return (self.NARROW_FUSED_TRAIN and x.is_cuda and self.training and torch.is_grad_enabled() and x.dtype == torch.float32
                and not x.requires_grad and x.dim() == 3 and x.shape[1] <= 8
                and self.conv1.out_channels == self.conv2.out_channels == self.conv3.out_channels == 8
                and all(b.affine and b.momentum is not None and b.track_running_stats and b.eps == bns[0].eps
                        and b.momentum == bns[0].momentum for b in bns))

    def _rows_trainable(self, x):
        from .spconv import core
        ok = lambda c: c % 4 == 0 and c <= 512 and 1024 % c == 0                                # noqa: E731
        # ROWS_MAX: a switch back to the reference's (B, C, P) modules for very tall inputs.  Rounds 2-3 kept the row form
        # below 1 M rows because the RECORDED step returned NaN gradients at configs[3]'s 2.1 M rows; that was ROCm 7.2
        # replaying recorded memset nodes (torch's reductions zero their semaphores with one) with a stale pattern, which
        # _lib.finish_graph now removes from every recorded step (csrc/glx_graph.hip).
        if x.shape[0] * x.shape[2] > self.ROWS_MAX:
            return False
        return (x.is_cuda and self.training and torch.is_grad_enabled() and x.dtype == torch.float32
                and core.USE_FUSED_TRAIN_BN and all(ok(m.num_features) and m.affine and m.momentum is not None
                                                    for m in (self.bn1, self.bn2, self.bn3)))

    USE_POINTMAX = True
    LAZY_H2 = True              # see _forward_train_rows
    ROW_CHUNKS = 128
    ROWS_MAX = 1 << 22

    @classmethod
    def _rows_linear(cls, x2d, conv, bias=True):
        """x2d (rows, C_in) @ W^T + b as ROW_CHUNKS batched products when the matrix is tall: the weight gradient
        autograd derives is then a batched GEMM + a sum over the batch (split-K) instead of ONE (C_out x C_in) GEMM
        with K = rows, for which the library's own choice at 2.1 M rows is slow."""
        w, rows = conv.weight[:, :, 0], x2d.shape[0]
        b_ = conv.bias if bias else None
        s_ = cls.ROW_CHUNKS
        if rows % s_ == 0 and rows // s_ >= 64:
            y = torch.bmm(x2d.view(s_, rows // s_, -1), w.t().unsqueeze(0).expand(s_, -1, -1)).view(rows, -1)
            return y if b_ is None else y + b_
        return F.linear(x2d, w, b_)

    # A bias in front of a training-mode BatchNorm moves the batch mean and nothing else: the normalised output is the same
    # function of x W^T, the bias's gradient is the sum of the BatchNorm's input gradient, which is zero.  True: the product
    # runs without the bias (one elementwise pass over the (rows, C) matrix less, and one column reduction less in the
    # backward), the running mean gets momentum x bias added, the bias's gradient is an exact zero (the reference's is
    # rounding noise around it).
    BIAS_INTO_RUNNING_MEAN = True

    # Layers up to 64 -> 128 channels on csrc/glx_rows.hip (RowsConvBN: the product with the BatchNorm statistics in its
    # epilogue + the transform forward; the statistics launch + ONE launch that applies the BatchNorm / ReLU backward on load and
    # forms both gradients backward) instead of library products + separate BatchNorm passes.  The 4 point features of the
    # first layer are padded to the kernels' 16.
    OWN_ROW_LAYERS = True

    @classmethod
    def _rows_layer(cls, x2d, conv, bn, relu, lazy=False):
        """relu?(bn(conv(x2d))) on rows; x2d may carry zero columns behind the conv's input channels and the result may carry
        zero columns behind its output channels (the narrow extractor's 8 channels run as 16: the kernels' tiles).
        lazy: -> (z, coef) with the transform left to the consumer (RowsConvBN's lazy form), or (h, None) where this layer
        does not run on the row kernels."""
        from .spconv import core
        pre = None
        fold = cls.BIAS_INTO_RUNNING_MEAN and conv.bias is not None and bn.track_running_stats
        cout, cin = conv.weight.shape[:2]
        xin = x2d.shape[1]
        if not fold and conv.bias is not None:
            h = core.fused_train_bn(bn, cls._rows_linear(x2d[:, :cin] if xin != cin else x2d, conv), relu, None)
            return (h, None) if lazy else h
        kin = xin if xin in (16, 32, 64) else (16 if xin < 16 else 0)
        kout = max(cout, 16)
        if cls.OWN_ROW_LAYERS and kin and core.USE_BN_STATE and bn.track_running_stats \
                and _lib.query("glx_rows_linear_supported", kin, kout):
            from .pcdet_ops.pointnet2.pointnet2_stack import voxel_pool_modules as vpm
            w = conv.weight[:, :, 0]
            if kin != cin or kout != cout:
                w = F.pad(w, (0, kin - cin, 0, kout - cout))
            if kin != xin:
                x2d = F.pad(x2d, (0, kin - xin))
            vpm._count(bn)
            if kout != cout:
                # a BatchNorm of kout channels around the module's: weight 1 / bias 0 / mean 0 / variance 1 behind its own, so
                # the extra columns stay exactly zero; the running statistics go back into the module's buffers afterwards
                pad = kout - cout
                proxy = types.SimpleNamespace(
                    weight=F.pad(bn.weight, (0, pad), value=1.0), bias=F.pad(bn.bias, (0, pad)), eps=bn.eps, momentum=bn.momentum,
                    track_running_stats=True, running_mean=F.pad(bn.running_mean, (0, pad)),
                    running_var=F.pad(bn.running_var, (0, pad), value=1.0))
                h = vpm.RowsConvBN.apply(x2d, w, proxy.weight, proxy.bias, proxy, relu, None)
                with torch.no_grad():
                    bn.running_mean.copy_(proxy.running_mean[:cout])
                    bn.running_var.copy_(proxy.running_var[:cout])
            else:
                h = vpm.RowsConvBN.apply(x2d, w, bn.weight, bn.bias, bn, relu, None, lazy)
                if lazy:
                    h, pre = h
            if conv.bias is not None:
                h = _ZeroGradOperand.apply(h, conv.bias)
        else:
            z = cls._rows_linear(x2d[:, :cin] if xin != cin else x2d, conv, bias=False)
            if conv.bias is not None:
                z = _ZeroGradOperand.apply(z, conv.bias)
            h = core.fused_train_bn(bn, z, relu, None)
        if conv.bias is not None:
            with torch.no_grad():
                bn.running_mean.add_(conv.bias, alpha=bn.momentum)
        return (h, pre) if lazy else h

    # one (B P, 16) row matrix of the points for every extractor that runs inside `with PointFeat.shared_rows():` on the same
    # points tensor (the CVAE's three extractors: the transposition and the zero padding to the row kernels' 16 columns are 0.1 ms
    # and 170 MB of traffic each)
    _ROWS_MEMO = None

    @classmethod
    @contextlib.contextmanager
    def shared_rows(cls):
        outer = cls._ROWS_MEMO
        cls._ROWS_MEMO = {}
        try:
            yield
        finally:
            cls._ROWS_MEMO = outer

    # the first layer (C -> 64) from the points: statistics from the moments of x, one pass that writes h1, one pass over its gradient
    # (csrc/glx_narrowfeat.hip, second half); False: the row kernels (product, statistics epilogue, transform pass; two backward passes)
    LAYER1_FROM_POINTS = True

    def _layer1_from_points(self, x):
        bn = self.bn1
        return (self.LAYER1_FROM_POINTS and self.conv1.out_channels == 64 and x.shape[1] <= 8 and not x.requires_grad
                and bn.affine and bn.momentum is not None and bn.track_running_stats)

    def _forward_train_rows(self, x):
        b, cin, p = x.shape
        if self._layer1_from_points(x):
            h = PointLayer1Train.apply(x, self.conv1.weight, self.conv1.bias, self.bn1.weight, self.bn1.bias, self.bn1)
            return self._forward_train_rows_tail(h, b, p)
        memo, key = type(self)._ROWS_MEMO, None
        if memo is not None and not x.requires_grad and cin < 16:
            key = (x.data_ptr(), x._version, tuple(x.shape), tuple(x.stride()))
            rows = memo.get(key)
            if rows is None:
                rows = memo[key] = F.pad(x.transpose(1, 2).reshape(b * p, cin), (0, 16 - cin))
        else:
            rows = x.transpose(1, 2).reshape(b * p, cin)
        h = self._rows_layer(rows, self.conv1, self.bn1, True)
        return self._forward_train_rows_tail(h, b, p)

    def _forward_train_rows_tail(self, h, b, p):
        pointmax = self.USE_POINTMAX and self.conv2.out_channels == 128 and self.conv3.out_channels == 512 and self.bn3.affine
        # the second layer's BatchNorm + ReLU applied by its consumers on load: h2 (1 GB at configs[3]) is never written
        lazy = pointmax and self.LAZY_H2 and PointMaxBN.F16X2 and PointMaxBN.FUSED_BN and PointMaxBN.OWN_MOMENTS \
            and self.bn3.momentum is not None
        h = self._rows_layer(h, self.conv2, self.bn2, True, lazy=lazy)
        pre = None
        if lazy:
            h, pre = h
        if pointmax and h.shape[1] == 128:
            # the 512-wide layer + BatchNorm + max over the points without the (B, P, 512) tensor
            return PointMaxBN.apply(h, self.conv3.weight[:, :, 0], self.conv3.bias, self.bn3.weight, self.bn3.bias, self.bn3, b, p, pre)
        h = self._rows_layer(h, self.conv3, self.bn3, False)
        return h.view(b, p, -1).amax(dim=1)[:, :self.conv3.out_channels]

    # ---- eval-mode fast path: one hand-written MFMA kernel for the whole extractor
    def _fusable(self, x):
        widths = (self.conv1.out_channels, self.conv2.out_channels, self.conv3.out_channels)
        return (x.is_cuda and not self.training and not torch.is_grad_enabled() and x.dtype == torch.float32
                and (widths == (64, 128, 512) or max(widths) <= 16) and self.conv1.in_channels <= 8)

    @staticmethod
    def _fold(conv, bn):
        """Conv1d(k=1) followed by eval-mode BatchNorm1d as one affine map (W, b)."""
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = conv.weight[:, :, 0] * s[:, None]
        b = (conv.bias - bn.running_mean) * s + bn.bias
        return w.float().contiguous(), b.float().contiguous()

    def _packed(self):
        """Folded weights in the kernel's MFMA fragment order (csrc/glx_pointnet.hip), cached until
        a parameter or running statistic changes."""
        tensors = [t for m in (self.conv1, self.bn1, self.conv2, self.bn2, self.conv3, self.bn3)
                   for t in (m.weight, m.bias)] + [self.bn1.running_mean, self.bn1.running_var,
                                                   self.bn2.running_mean, self.bn2.running_var,
                                                   self.bn3.running_mean, self.bn3.running_var]
        tag = tuple((t._version, t.data_ptr()) for t in tensors) + (_lib.weights_epoch(*tensors),)
        cache = self.__dict__.get("_glx_packed")
        if cache is None or cache[0] != tag:
            with torch.no_grad():
                w1, b1 = self._fold(self.conv1, self.bn1)
                w2, b2 = self._fold(self.conv2, self.bn2)
                w3, b3 = self._fold(self.conv3, self.bn3)
                if w3.shape[0] == 512:
                    # [tile_out, i, tile_in, q, e] -> [tile_out, tile_in, q, i, e]: lane = 16 q + i
                    w2p = w2.view(8, 16, 4, 4, 4).permute(0, 2, 3, 1, 4).contiguous()
                    w3p = w3.view(32, 16, 8, 4, 4).permute(0, 2, 3, 1, 4).contiguous()
                else:
                    w2p, w3p = w2, w3                      # narrow extractor: plain row-major
            cache = (tag, (w1, b1, w2p, b2, w3p, b3))
            self.__dict__["_glx_packed"] = cache
        return cache[1]

    # layers 2 and 3 of the wide extractor as f16 x 2 products (csrc/glx_pointnet.hip "f16 x 2 form"): three fp16 MFMAs per product
    # tile instead of eight fp32 ones, >= 20.4 bits per product; False: exact fp32 MFMA products (glx_pointnet_feat)
    F16X2 = True

    @staticmethod
    def _f16x2_image(w, row_scale=None, scale=1.0, sign_only=False):
        """(Cout, Cin) fp32 (any strides), times row_scale[row] times scale -> (two fp16 planes of w 2^ew[row] in the kernels'
        operand order, ew (Cout,) int32): [tile][k-step s][plane][lane 16 q + m][slot 4 h + e] = W[16 tile + m][32 s + 16 h + 4 q + e].
        One launch on the device (glx_f16x2_pack); the tensor statements below are the same image for host tensors."""
        cout, cin = w.shape
        if w.is_cuda and w.dtype == torch.float32:
            img = torch.empty((cout // 16, cin // 32, 2, 4, 16, 2, 4), dtype=torch.float16, device=w.device)
            ew = torch.empty(cout, dtype=torch.int32, device=w.device)
            _lib.call("glx_f16x2_pack", w, cout, cin, ctypes.c_longlong(w.stride(0)), ctypes.c_longlong(w.stride(1)),
                      row_scale, 1 if sign_only else 0, ctypes.c_float(scale), img, ew)
            return img, ew
        if row_scale is not None:
            w = w * (torch.where(row_scale >= 0, 1.0, -1.0) if sign_only else row_scale)[:, None]
        w = (w * scale).contiguous()
        m = w.abs().amax(dim=1)
        e = torch.where(m > 0, 14 - torch.floor(torch.log2(m.clamp_min(1e-38))), torch.zeros_like(m))
        # floor(log2) in floating point can be one off at exact powers of two: settle with the integer test the kernels use
        scaled = m * torch.exp2(e)
        e = torch.where(scaled >= 2.0 ** 15, e - 1, torch.where((scaled < 2.0 ** 14) & (m > 0), e + 1, e)).clamp_(-110, 110)
        ws = w * torch.exp2(e)[:, None]
        a = ws.half()
        b = (ws - a.float()).half()

        def order(p):           # (Cout, Cin) -> [tile][s][q][m][h][e] = [tile][s][lane][slot]
            return p.view(cout // 16, 16, cin // 32, 2, 4, 4).permute(0, 2, 4, 1, 3, 5).contiguous()
        img = torch.stack([order(a), order(b)], dim=2).contiguous()          # [tile][s][plane][q][m][h][e]
        return img, e.to(torch.int32).contiguous()

    def _packed_f16(self):
        tag, packed = self.__dict__["_glx_packed"]
        hit = self.__dict__.get("_glx_packed_f16")
        if hit is None or hit[0] != tag:
            w1, b1, w2p, b2, w3p, b3 = packed
            with torch.no_grad():
                w2, _ = self._fold(self.conv2, self.bn2)
                w3, _ = self._fold(self.conv3, self.bn3)
                hit = (tag, self._f16x2_image(w2) + self._f16x2_image(w3))
            self.__dict__["_glx_packed_f16"] = hit
        return hit[1]

    def _forward_fused(self, x):
        from ._lib import call
        x = x.contiguous()
        B, cin, P = x.shape
        w1, b1, w2p, b2, w3p, b3 = self._packed()
        c3 = self.conv3.out_channels
        out = torch.empty((B, c3), dtype=torch.float32, device=x.device)
        if c3 == 512 and self.F16X2:
            w2h, e2, w3h, e3 = self._packed_f16()
            call("glx_pointnet_feat_f16x2", x, B, cin, P, w1, b1, w2h, e2, b2, w3h, e3, b3, out)
        elif c3 == 512:
            call("glx_pointnet_feat", x, B, cin, P, w1, b1, w2p, b2, w3p, b3, out)
        else:
            call("glx_pointnet_feat_small", x, B, cin, P, self.conv1.out_channels, self.conv2.out_channels,
                 c3, w1, b1, w2p, b2, w3p, b3, out)
        return out


class _ZeroGradOperand(torch.autograd.Function):
    """z, with `other` tied into the graph at an exactly zero gradient (PointFeat.BIAS_INTO_RUNNING_MEAN)."""

    @staticmethod
    def forward(ctx, z, other):
        ctx.save_for_backward(other)
        return z.view_as(z)

    @staticmethod
    def backward(ctx, g):
        return g, torch.zeros_like(ctx.saved_tensors[0])


class PointLayer1Train(torch.autograd.Function):
    """conv1 + bn1 + relu of the wide extractor in training mode, from the points (csrc/glx_narrowfeat.hip): x (B, C <= 8, P) ->
    h1 (B P, 64) rows.  The running statistics are updated by the launch; the convolution's bias receives exact zeros."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, bn):
        from .pcdet_ops.pointnet2.pointnet2_stack import voxel_pool_modules as vpm
        x = x.contiguous()
        B, C, P = x.shape
        dev = x.device
        w2 = w.detach().reshape(64, C).contiguous().float()
        h1 = torch.empty((B * P, 64), dtype=torch.float32, device=dev)
        coef = torch.empty((4, 64), dtype=torch.float32, device=dev)
        moments = torch.empty(44, dtype=torch.float64, device=dev)
        wsp = _lib.workspace.get(_lib.query("glx_point_layer1_workspace_bytes"), dev)
        vpm._count(bn)
        _lib.call("glx_point_layer1_train_forward", x, B, C, P, w2, bias, gamma, beta, bn.running_mean, bn.running_var,
                  ctypes.c_float(bn.eps), ctypes.c_float(bn.momentum), h1, coef, moments, wsp, _lib.size_arg(wsp.numel()))
        _lib.bump_weights_epoch((bn.running_mean, bn.running_var))
        ctx.save_for_backward(x, w2, coef, moments)
        ctx.wshape, ctx.has_bias = w.shape, bias is not None
        return h1

    @staticmethod
    def backward(ctx, dh1):
        x, w2, coef, moments = ctx.saved_tensors
        B, C, P = x.shape
        dev = x.device
        grads = torch.empty(64 * C + 128, dtype=torch.float32, device=dev)
        wsp = _lib.workspace.get(_lib.query("glx_point_layer1_workspace_bytes"), dev)
        _lib.call("glx_point_layer1_train_backward", x, B, C, P, w2, coef, moments, dh1.contiguous().float(), grads, wsp,
                  _lib.size_arg(wsp.numel()))
        dw = grads[:64 * C].view(ctx.wshape)
        db = torch.zeros(64, dtype=torch.float32, device=dev) if ctx.has_bias else None
        return None, dw, db, grads[64 * C:64 * C + 64], grads[64 * C + 64:], None


class NarrowFeatTrain(torch.autograd.Function):
    """PointFeat(C, (8, 8, 8)) in training mode on the device without intermediate tensors (csrc/glx_narrowfeat.hip: batch statistics of
    every BatchNorm from the moments of its input, every pass recomputes the layers in front from the points): x (B, C <= 8, P) -> (B, 8).
    Running statistics are updated by the launch; the convolutions' biases receive exact zeros (a bias in front of a training BatchNorm
    only moves the batch mean)."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3, g3, be3, bns):
        from .pcdet_ops.pointnet2.pointnet2_stack import voxel_pool_modules as vpm
        x = x.contiguous()
        B, C, P = x.shape
        dev = x.device
        ws = [w.detach().reshape(8, -1).contiguous().float() for w in (w1, w2, w3)]
        out = torch.empty((B, 8), dtype=torch.float32, device=dev)
        arg = torch.empty((B, 8), dtype=torch.int32, device=dev)
        xh = torch.empty((B, 8), dtype=torch.float32, device=dev)
        coef = torch.empty((3, 4, 8), dtype=torch.float32, device=dev)
        wsp = _lib.workspace.get(_lib.query("glx_narrowfeat_workspace_bytes"), dev)
        for bn in bns:
            vpm._count(bn)
        _lib.call("glx_narrowfeat_train_forward", x, B, C, P, ws[0], b1, g1, be1, bns[0].running_mean, bns[0].running_var, ws[1], b2, g2, be2,
                  bns[1].running_mean, bns[1].running_var, ws[2], b3, g3, be3, bns[2].running_mean, bns[2].running_var,
                  ctypes.c_float(bns[0].eps), ctypes.c_float(bns[0].momentum), out, arg, xh, coef, wsp, _lib.size_arg(wsp.numel()))
        _lib.bump_weights_epoch([t for bn in bns for t in (bn.running_mean, bn.running_var)])
        ctx.save_for_backward(x, ws[0], ws[1], ws[2], coef, arg, xh)
        ctx.shapes = (w1.shape, w2.shape, w3.shape)
        ctx.has_bias = (b1 is not None, b2 is not None, b3 is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, w1, w2, w3, coef, arg, xh = ctx.saved_tensors
        B, C, P = x.shape
        dev = x.device
        grads = torch.empty((4, 64), dtype=torch.float32, device=dev)
        wsp = _lib.workspace.get(_lib.query("glx_narrowfeat_workspace_bytes"), dev)
        _lib.call("glx_narrowfeat_train_backward", x, B, C, P, w1, w2, w3, coef, gout.contiguous().float(), arg, xh, grads, wsp,
                  _lib.size_arg(wsp.numel()))
        dw1 = grads[0].view(8, 8)[:, :C].reshape(ctx.shapes[0])
        dw2, dw3 = grads[1].view(ctx.shapes[1]), grads[2].view(ctx.shapes[2])
        v = grads[3]
        zero = lambda has: torch.zeros(8, dtype=torch.float32, device=dev) if has else None          # noqa: E731
        return (None, dw1, zero(ctx.has_bias[0]), v[0:8], v[8:16], dw2, zero(ctx.has_bias[1]), v[16:24], v[24:32],
                dw3, zero(ctx.has_bias[2]), v[32:40], v[40:48], None)


# gradient tensor (data pointer) -> (partial sums of the affine launch, of the scatter launch, data pointer of the raw output they belong
# to): BatchNorm-backward sums that the producers of a gradient took, for the backward of the lazy row layer in front
# (pcdet_ops ... voxel_pool_modules.RowsConvBN.backward); one entry, replaced by the next backward pass
BWD_PARTIALS = {}


class PointMaxBN(torch.autograd.Function):
    """out[b, :] = max_p BatchNorm1d(W3 h2[b, p, :] + b3) in training mode (point_net.py:22-28: conv3 + bn3 + max over the
    points) without the (B, P, 512) tensor -- csrc/glx_pointnet.hip, "training twin, layer 3".  h2 (B * P, 128) rows;
    weight (512, 128); returns (B, 512).  Running statistics of `bn` are updated as nn.BatchNorm1d does."""

    # the backward sums of the BatchNorm in front (lazy form: `pre`) taken by the launches that write the gradient (no extra pass)
    BWD_SUMS_IN_PRODUCERS = True

    @staticmethod
    def forward(ctx, h2, weight, bias, gamma, beta, bn, B, P, pre=None):
        """pre (scale | shift, 256 floats) != None: `h2` is the RAW output z of the layer in front and the rows this layer works on
        are relu(z scale + shift), formed on load by every kernel that reads them (f16 x 2 path with the fused BatchNorm only); the
        gradient returned for `h2` is the gradient of those rows."""
        BWD_PARTIALS.clear()            # (sums of an earlier backward pass that nobody took)
        from ._lib import call
        dev = h2.device
        h2 = h2.contiguous()
        ctx.pre = None
        if pre is not None and not (PointMaxBN.F16X2 and PointMaxBN.FUSED_BN and PointMaxBN.OWN_MOMENTS and bn.momentum is not None):
            y = torch.empty_like(h2)                       # a path without the on-load form: materialise the rows after all
            call("glx_bn_apply_forward", h2, pre, 1, h2.shape[0], h2.shape[1], None, y, 0)
            h2, pre = y, None
        W3 = weight.contiguous()
        vmax, vmin = (torch.empty((B, 512), dtype=torch.float32, device=dev) for _ in range(2))
        amax, amin = (torch.empty((B, 512), dtype=torch.int32, device=dev) for _ in range(2))
        R = B * P
        G2 = H1 = None
        if PointMaxBN.F16X2:
            # f16 x 2 products; the batch statistics of y = h2 W3^T from the two moments of h2 the backward needs anyway:
            # sum_r y = W3 (sum_r h2), sum_r y^2 = diag(W3 (h2^T h2) W3^T)
            # one extreme per channel: the BatchNorm's weight decides which (scale = gamma invstd, invstd > 0), so the rows of W3
            # go in with its sign and the pass returns max_p (sign y)
            w3h, e3 = PointFeat._f16x2_image(W3.detach(), row_scale=gamma.detach(), sign_only=True)
            call("glx_pointmax_forward_f16x2", h2, B, P, w3h, e3, vmax, amax, pre)
            G2d, H1 = PointMaxBN._moments(h2, R, pre)
            ctx.pre = pre
            if PointMaxBN.FUSED_BN and bn.momentum is not None:
                # statistics, running statistics, the choice of the extreme and the transform: two launches (~40 tensor statements)
                mean_nb, invstd, scale = (torch.empty(512, dtype=torch.float32, device=dev) for _ in range(3))
                out = torch.empty((B, 512), dtype=torch.float32, device=dev)
                rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
                call("glx_pointmax_bn_forward", W3, G2d, H1, ctypes.c_longlong(R), vmax, B, gamma, beta, bias, ctypes.c_float(bn.eps),
                     ctypes.c_float(bn.momentum), rm, rv, mean_nb, invstd, scale, out)
                if bn.track_running_stats:
                    bn.num_batches_tracked.add_(1)
                    _lib.bump_weights_epoch((rm, rv))
                ctx.save_for_backward(h2, W3, vmax, amax, mean_nb, invstd, scale)
                ctx.moments, ctx.fused = (G2d, H1), True
                ctx.dims = (B, P, bias is not None)
                return out
            sgn = torch.where(gamma.detach() >= 0, 1.0, -1.0)
            vmax *= sgn
            vmin, amin = vmax, amax
            W3d = W3.detach().double()
            mean_nb = (W3d @ H1.double()) / R
            var = (((W3d @ G2d) * W3d).sum(1) / R - mean_nb * mean_nb).clamp_min_(0.0)
            G2 = G2d.float()
        else:
            w3p = W3.view(32, 16, 8, 4, 4).permute(0, 2, 3, 1, 4).contiguous()
            s1, s2 = (torch.empty((B, 512), dtype=torch.float32, device=dev) for _ in range(2))
            call("glx_pointmax_forward", h2, B, P, w3p, vmax, vmin, amax, amin, s1, s2)
            mean_nb = s1.double().sum(0) / R                        # batch mean of y without the bias
            var = (s2.double().sum(0) / R - mean_nb * mean_nb).clamp_min_(0.0)
        invstd = torch.rsqrt(var + bn.eps).float()
        mean_nb = mean_nb.float()
        scale = gamma * invstd
        sel = scale >= 0
        ext = torch.where(sel, vmax, vmin)
        arg = torch.where(sel, amax, amin)
        out = (ext - mean_nb) * scale + beta
        if bn.track_running_stats:
            with torch.no_grad():
                m = bn.momentum
                bn.running_mean.mul_(1 - m).add_(mean_nb + (bias if bias is not None else 0), alpha=m)
                bn.running_var.mul_(1 - m).add_(var.float() * (R / max(R - 1, 1)), alpha=m)
                bn.num_batches_tracked.add_(1)
        ctx.save_for_backward(h2, W3, ext, arg, mean_nb, invstd, scale)
        ctx.moments, ctx.fused = (G2, H1), False
        ctx.dims = (B, P, bias is not None)
        return out

    # the BatchNorm's own arithmetic around the f16 x 2 pass (statistics from the moments, running statistics, transform; the backward's
    # sums, vectors and 128 x 128 matrices) as five launches of csrc/glx_pointnet.hip; False: tensor statements
    FUSED_BN = True

    # the 128 -> 512 layer as f16 x 2 products (csrc/glx_pointnet.hip, k_pointmax_fwd_f16); False: fp32 MFMA products
    F16X2 = True

    # the two moments in one pass of csrc/glx_pointnet.hip's k_rows128_moments (bf16 x 3 products, fixed summation order); False: a
    # library split-K product + a column reduction
    OWN_MOMENTS = True

    @staticmethod
    def _moments(h2, R, pre=None):
        """(h2^T h2 in fp64 from fp32 partial products over row chunks, sum_r h2); pre: of relu(h2 scale + shift)."""
        if PointMaxBN.OWN_MOMENTS and h2.is_cuda and h2.dtype == torch.float32 and h2.shape[1] == 128 and h2.is_contiguous():
            from ._lib import call, query, size_arg, workspace
            G = torch.empty((128, 128), dtype=torch.float64, device=h2.device)
            H = torch.empty(128, dtype=torch.float32, device=h2.device)
            n = query("glx_rows128_moments_workspace_bytes")
            ws = workspace.get(n, h2.device)
            call("glx_rows128_moments", h2, ctypes.c_longlong(R), G, H, pre, ws, size_arg(n))
            return G, H
        assert pre is None
        S = 128 if R % 128 == 0 and R >= 128 * 256 else 1
        hc = h2.view(S, R // S, 128)
        return torch.bmm(hc.transpose(1, 2), hc).double().sum(0), h2.sum(0)

    @staticmethod
    def backward(ctx, g):
        from ._lib import call, query, size_arg
        h2, W3, ext, arg, mean_nb, invstd, scale = ctx.saved_tensors
        B, P, has_bias = ctx.dims
        R = B * P
        g = g.contiguous()
        if ctx.fused:
            dev = h2.device
            dgamma, dbeta, bvec, cvec = (torch.empty(512, dtype=torch.float32, device=dev) for _ in range(4))
            M = torch.empty((128, 128), dtype=torch.float32, device=dev)
            nv = torch.empty(128, dtype=torch.float32, device=dev)
            call("glx_pointmax_bn_backward_sums", g, ext, B, ctypes.c_longlong(R), W3, mean_nb, invstd, scale, dgamma, dbeta, bvec, cvec,
                 M, nv)
            d_h2 = d_w = None
            if ctx.needs_input_grad[0]:
                mh, em = PointFeat._f16x2_image(M.t(), scale=-1.0)
                d_h2 = torch.empty_like(h2)
                if ctx.pre is not None and PointMaxBN.BWD_SUMS_IN_PRODUCERS:
                    # h2 is the raw output of the layer in front and d_h2 the gradient behind its BatchNorm + ReLU: the two launches
                    # that write d_h2 also take that BatchNorm's backward sums; RowsConvBN.backward picks them up (BWD_PARTIALS)
                    na = int(_lib.load().glx_rows128_affine_blocks(ctypes.c_longlong(R)))
                    pa = torch.empty((na, 2, 128), dtype=torch.float32, device=dev)
                    pb = torch.empty((B, 2, 128), dtype=torch.float32, device=dev)
                    call("glx_rows128_affine_f16x2_sums", h2, ctypes.c_longlong(R), mh, em, nv, d_h2, ctx.pre, pa)
                    call("glx_pointmax_scatter_add_scaled_sums", arg, g, scale, W3, B, P, d_h2, h2, ctx.pre, pb)
                    BWD_PARTIALS.clear()
                    BWD_PARTIALS[d_h2.data_ptr()] = (pa, pb, h2.data_ptr())
                else:
                    call("glx_rows128_affine_f16x2", h2, ctypes.c_longlong(R), mh, em, nv, d_h2, ctx.pre)
                    call("glx_pointmax_scatter_add_scaled", arg, g, scale, W3, B, P, d_h2)
            if ctx.needs_input_grad[1]:
                T = torch.empty_like(W3)
                ws = torch.empty(query("glx_pointmax_wsum_workspace_bytes"), dtype=torch.uint8, device=dev)
                call("glx_pointmax_wsum_pre", g, arg, h2, ctx.pre, B, P, T, ws, size_arg(ws.numel()))
                G2d, H1 = ctx.moments
                d_w = torch.empty_like(W3)
                call("glx_pointmax_bn_backward_weight", W3, G2d, H1, T, scale, bvec, cvec, mean_nb, d_w)
            d_b = torch.zeros_like(dbeta) if has_bias else None
            return d_h2, d_w, d_b, dgamma, dbeta, None, None, None, None
        xhat = (ext - mean_nb) * invstd
        dbeta = g.sum(0)
        dgamma = (g * xhat).sum(0)
        # dy[r, c] = a_c g^[r, c] - bvec_c - cvec_c (y[r, c] - mean_c):  BatchNorm backward with the max's sparse gradient g^
        bvec = scale * dbeta / R
        cvec = scale * invstd * dgamma / R
        d_h2 = d_w = None
        if ctx.needs_input_grad[0]:
            v = (bvec - cvec * mean_nb) @ W3                                    # (128)
            M = W3.t() @ (cvec[:, None] * W3)                                   # (128, 128)
            d_h2 = torch.empty_like(h2)
            if PointMaxBN.F16X2:
                # dense part first (f16 x 2 products, one pass: read h2, write d_h2), then the extreme points' rows on top
                mh, em = PointFeat._f16x2_image(M.t(), scale=-1.0)
                call("glx_rows128_affine_f16x2", h2, ctypes.c_longlong(R), mh, em, (-v).contiguous(), d_h2, None)
                call("glx_pointmax_scatter_add", arg, (g * scale).contiguous(), W3, B, P, d_h2)
            else:
                call("glx_pointmax_scatter", arg, (g * scale).contiguous(), W3, (-v).contiguous(), B, P, d_h2)
                d_h2.addmm_(h2, M, alpha=-1.0)
        if ctx.needs_input_grad[1]:
            T = torch.empty_like(W3)
            ws = torch.empty(query("glx_pointmax_wsum_workspace_bytes"), dtype=torch.uint8, device=h2.device)
            call("glx_pointmax_wsum", g, arg, h2, B, P, T, ws, size_arg(ws.numel()))
            G2, H1 = ctx.moments
            if G2 is None:
                G2d, H1 = PointMaxBN._moments(h2, R)                            # h2^T h2, sum_r h2
                G2 = G2d.float()
            d_w = scale[:, None] * T - bvec[:, None] * H1[None] - cvec[:, None] * (W3 @ G2 - mean_nb[:, None] * H1[None])
        d_b = torch.zeros_like(dbeta) if has_bias else None                     # sum_r dy = 0 exactly (the BatchNorm removes it)
        return d_h2, d_w, d_b, dgamma, dbeta, None, None, None, None


class LatentEncoder(nn.Module):
    """Encoder_x (cond_dim = 0) / Encoder_xy (cond_dim = 8): point feature (+ encoded box) ->
    mu, logvar of the latent Gaussian; scale of the distribution is exp(logvar) + 3e-22 (model.py:49,77)."""

    def __init__(self, input_channels, latent_size, cond_dim=0, widths=(64, 128, 512)):
        super().__init__()
        self.fe = PointFeat(input_channels, widths)
        self.fc1 = nn.Linear(widths[2] + cond_dim, latent_size)
        self.fc2 = nn.Linear(widths[2] + cond_dim, latent_size)

    def forward(self, points, cond=None):
        h = self.fe(points)
        if cond is not None:
            h = torch.cat([h, cond], dim=1)
        mu, logvar = self.fc1(h), self.fc2(h)
        # validate_args=False: the argument checks read a device boolean back (not capturable, and a sync per call)
        dist = torch.distributions.Independent(
            torch.distributions.Normal(mu, torch.exp(logvar) + 3e-22, validate_args=False), 1, validate_args=False)
        return dist, mu, logvar


class BoxDecoder(nn.Module):
    """Object_feat_encoder (model.py:82-142): small point feature + latent z -> 3 centre, 3 size,
    1 heading residual, num_bins direction logits."""

    def __init__(self, input_channels, latent_dim, num_bins=2, width=64):
        super().__init__()
        self.fe = PointFeat(input_channels, (8, 8, 8))
        self.fc1, self.fc2 = nn.Linear(8 + latent_dim, width), nn.Linear(width, width)
        self.bn1, self.bn2 = nn.BatchNorm1d(width), nn.BatchNorm1d(width)
        self.fc_s1, self.fc_s2 = nn.Linear(width, width), nn.Linear(width, 3, bias=False)
        self.fc_ce1, self.fc_ce2 = nn.Linear(width, width), nn.Linear(width, 3, bias=False)
        self.fc_hr1, self.fc_hr2 = nn.Linear(width, width), nn.Linear(width, 1, bias=False)
        self.fc_dir1, self.fc_dir2 = nn.Linear(width, width), nn.Linear(width, num_bins, bias=False)

    def forward(self, points, z):
        h = torch.cat([self.fe(points), z], dim=1)
        h = F.relu(self.bn1(self.fc1(h)))
        h = F.relu(self.bn2(self.fc2(h)))
        head = lambda a, b: b(F.relu(a(h)))                                   # noqa: E731
        return torch.cat([head(self.fc_ce1, self.fc_ce2), head(self.fc_s1, self.fc_s2),
                          head(self.fc_hr1, self.fc_hr2), head(self.fc_dir1, self.fc_dir2)], dim=1)


def limit_period(val, offset=0.5, period=np.pi):
    return val - torch.floor(val / period + offset) * period                  # common_utils.py:35-38


class CVAE(nn.Module):
    """Generator of cvae_uncertainty/model.py: x_encoder (prior), xy_encoder (posterior), obj_encoder
    (decoder).  `sample()` is its eval-mode forward (:245-265) with an explicit noise tensor, so the
    30 latent samples per object of predict.sh:8-11 are 30 calls / one batched call."""

    def __init__(self, input_channels=4, latent_dim=8, num_dir_bins=2, dir_offset=0.78539, dir_limit_offset=0.0):
        super().__init__()
        self.latent_dim, self.num_dir_bins = latent_dim, num_dir_bins
        self.dir_offset, self.dir_limit_offset = dir_offset, dir_limit_offset
        self.obj_encoder = BoxDecoder(input_channels, latent_dim, num_dir_bins)
        self.xy_encoder = LatentEncoder(input_channels, latent_dim, cond_dim=8)
        self.x_encoder = LatentEncoder(input_channels, latent_dim)

    @staticmethod
    def reparametrize(mu, logvar, eps):
        return eps * torch.exp(0.5 * logvar) + mu                              # model.py:194-198

    # eval-mode sampler as two launches: both extractors over the points (glx_pointnet_feat_f16x2_pair), everything behind them
    # (glx_cvae_sample_tail); False: module by module (the extractors still on their fused kernels)
    FUSED_SAMPLER = True

    def _sample_fusable(self, points):
        fe, fn, dec = self.x_encoder.fe, self.obj_encoder.fe, self.obj_encoder
        wide = (fe.conv1.out_channels, fe.conv2.out_channels, fe.conv3.out_channels) == (64, 128, 512)
        narrow = (fn.conv1.out_channels, fn.conv2.out_channels, fn.conv3.out_channels) == (8, 8, 8)
        return (self.FUSED_SAMPLER and PointFeat.F16X2 and points.is_cuda and points.dtype == torch.float32
                and not self.training and not torch.is_grad_enabled() and wide and narrow and points.shape[1] <= 8
                and fe.conv1.in_channels == fn.conv1.in_channels == points.shape[1] and self.latent_dim == 8
                and dec.fc1.out_features == dec.fc2.out_features == 64 and dec.fc1.in_features == 16
                and 1 <= self.num_dir_bins <= 9 and all(m.track_running_stats for m in (dec.bn1, dec.bn2)))

    def _sample_pack(self):
        """(the narrow extractor's 216 folded weights, the tail's weight buffer) as glx_pointnet_feat_f16x2_pair /
        glx_cvae_sample_tail take them; cached until a parameter or running statistic changes."""
        tensors = list(self.parameters()) + list(self.buffers())
        tag = tuple((t._version, t.data_ptr()) for t in tensors) + (_lib.weights_epoch(*tensors),)
        hit = self.__dict__.get("_glx_sample_pack")
        if hit is None or hit[0] != tag:
            with torch.no_grad():
                fn, dec, enc = self.obj_encoder.fe, self.obj_encoder, self.x_encoder
                w1, b1 = PointFeat._fold(fn.conv1, fn.bn1)
                w2, b2 = PointFeat._fold(fn.conv2, fn.bn2)
                w3, b3 = PointFeat._fold(fn.conv3, fn.bn3)
                narrow = torch.cat([F.pad(w1, (0, 8 - w1.shape[1])).reshape(-1), b1, w2.reshape(-1), b2, w3.reshape(-1), b3])

                def fold_fc(fc, bn):
                    sc = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    return fc.weight * sc[:, None], (fc.bias - bn.running_mean) * sc + bn.bias
                f1w, f1b = fold_fc(dec.fc1, dec.bn1)
                f2w, f2b = fold_fc(dec.fc2, dec.bn2)
                heads1 = (dec.fc_ce1, dec.fc_s1, dec.fc_hr1, dec.fc_dir1)
                heads2 = (dec.fc_ce2, dec.fc_s2, dec.fc_hr2, dec.fc_dir2)
                tail = torch.cat([torch.cat([enc.fc1.weight, enc.fc2.weight]).reshape(-1), enc.fc1.bias, enc.fc2.bias,
                                  f1w.t().reshape(-1), f1b, f2w.t().reshape(-1), f2b,
                                  torch.cat([h.weight.t().reshape(-1) for h in heads1]), torch.cat([h.bias for h in heads1]),
                                  torch.cat([h.weight for h in heads2]).reshape(-1)]).float().contiguous()
            hit = (tag, (narrow.float().contiguous(), tail))
            self.__dict__["_glx_sample_pack"] = hit
        return hit[1]

    def _sample_fused(self, points, eps):
        from ._lib import call
        points = points.contiguous()
        B, cin, P = points.shape
        fe = self.x_encoder.fe
        w1, b1, _, b2, _, b3 = fe._packed()
        w2h, e2, w3h, e3 = fe._packed_f16()
        narrow, tail = self._sample_pack()
        dev = points.device
        f512 = torch.empty((B, 512), dtype=torch.float32, device=dev)
        f8 = torch.empty((B, 8), dtype=torch.float32, device=dev)
        out = torch.empty((B, 7 + self.num_dir_bins), dtype=torch.float32, device=dev)
        if eps is None:
            eps = torch.randn((B, 8), dtype=torch.float32, device=dev)
        call("glx_pointnet_feat_f16x2_pair", points, B, cin, P, w1, b1, w2h, e2, b2, w3h, e3, b3, f512, narrow, f8)
        call("glx_cvae_sample_tail", f512, f8, eps.contiguous().float(), tail, B, self.num_dir_bins,
             ctypes.c_float(self.dir_offset), ctypes.c_float(self.dir_limit_offset), out)
        return out

    def sample(self, points, eps=None):
        """points (B, C, P) -> boxes (B, 7 + num_dir_bins) with the heading decoded from its bin."""
        if self._sample_fusable(points):
            return self._sample_fused(points, eps)
        _, mu, logvar = self.x_encoder(points)
        if eps is None:
            eps = torch.randn_like(mu)
        pred = self.obj_encoder(points, self.reparametrize(mu, logvar, eps))
        period = 2 * np.pi / self.num_dir_bins
        labels = pred[:, -self.num_dir_bins:].max(dim=-1)[1]
        rot = limit_period(pred[:, 6] - self.dir_offset, self.dir_limit_offset, period)
        pred = pred.clone()
        pred[:, 6] = rot + self.dir_offset + period * labels.to(pred.dtype)
        return pred

    def posterior_prior(self, points, gt_boxes_input):
        """Training-side encoders: (posterior, prior) distributions and their KL (model.py:205-212)."""
        post, mu_xy, logvar_xy = self.xy_encoder(points, gt_boxes_input)
        prior, mu_x, logvar_x = self.x_encoder(points)
        kl = torch.distributions.kl.kl_divergence(post, prior)
        return post, prior, kl, (mu_xy, logvar_xy, mu_x, logvar_x)

    FUSED_LOSSES = True          # training_losses: the regression / direction / KL terms with their gradients in one launch

    # ---- training step (cvae_uncertainty/model.py:205-240, 267-370; train_utils/train_utils.py:50-72)
    LOSS_WEIGHTS = dict(latent_weight=10.0, loc_weight=10.0, dir_weight=0.002, code_weights=(1.0,) * 7)   # cfgs/exp20.yaml

    def training_losses(self, points, gt_boxes_input, gt_boxes, eps_post=None, loss_weights=None, flat_optimizer=None):
        """The training branch of Generator.forward + get_training_loss.
        points (B, C, P), gt_boxes_input (B, 8) (the posterior's condition), gt_boxes (B, 7) (regression labels),
        eps_post (B, latent) = the noise of the posterior's reparametrisation (drawn here when None; the reference also
        draws one for the prior, model.py:222, and never uses it).
        -> (reg_loss_post, lattent_loss, regular_loss), parts -- the tuple train_one_epoch sums (after scaling the
        latent term by its annealing factor, train_utils.py:57-59); parts = device scalars named like the tb_dict.
        flat_optimizer: an optim.FlatAdamW that owns every parameter -- regular_loss then comes from its flat buffer WITHOUT a
        graph behind it (two launches), and the caller adds its gradient with `flat_optimizer.add_l2_norm_grad()` after the
        gradients are packed (cvae_train.CVAETrainStep): the same numbers as autograd's ~5 launches per parameter tensor."""
        w = dict(self.LOSS_WEIGHTS, **(loss_weights or {}))
        with PointFeat.shared_rows():
            post, mu_xy, logvar_xy = self.xy_encoder(points, gt_boxes_input)
            prior, mu_x, logvar_x = self.x_encoder(points)
            if eps_post is None:
                eps_post = torch.randn_like(mu_xy)
            pred = self.obj_encoder(points, self.reparametrize(mu_xy, logvar_xy, eps_post))
        if self.FUSED_LOSSES and pred.is_cuda and pred.dtype == torch.float32 and 1 <= self.num_dir_bins <= 9:
            # both data terms and their gradients in one launch (csrc/glx_pointnet.hip, k_cvae_losses)
            loc, dir_loss, latent = CvaeLosses.apply(pred, gt_boxes, mu_xy, logvar_xy, mu_x, logvar_x,
                                                     _code_weights(tuple(w["code_weights"]), pred.device), float(w["loc_weight"]),
                                                     float(w["dir_weight"]), float(w["latent_weight"]), float(self.dir_offset),
                                                     int(self.num_dir_bins))
            reg = loc + dir_loss
            parts = {"loss_loc": loc, "loss_dir": dir_loss, "loss_reg": reg}
        else:
            latent = torch.distributions.kl.kl_divergence(post, prior).mean() * w["latent_weight"]
            reg, parts = cvae_reg_loss(pred, gt_boxes, w, self.dir_offset, self.num_dir_bins)
        if flat_optimizer is not None:
            regular = flat_optimizer.l2_norm_sum(list(self.xy_encoder.parameters()) + list(self.x_encoder.parameters())
                                                 + list(self.obj_encoder.parameters()), 1e-4)
        else:
            regular = 1e-4 * (l2_regularisation(self.xy_encoder) + l2_regularisation(self.x_encoder)
                              + l2_regularisation(self.obj_encoder))
        parts = dict(parts, box_pred_post=pred)
        return (reg, latent, regular), parts


class CvaeLosses(torch.autograd.Function):
    """(loss_loc, loss_dir, latent) of cvae_reg_loss + the KL term (weights applied) with the gradients of all five inputs from ONE
    launch (glx_cvae_losses); backward scales them by the incoming gradients."""

    @staticmethod
    def forward(ctx, pred, labels, mu1, lv1, mu2, lv2, cw, loc_weight, dir_weight, latent_weight, dir_offset, bins):
        pred, labels = pred.contiguous(), labels.contiguous().float()
        B, L = pred.shape[0], mu1.shape[1]
        dev = pred.device
        out = torch.empty(3, dtype=torch.float32, device=dev)
        d_pred = torch.empty_like(pred)
        d_lat = torch.empty((4, B, L), dtype=torch.float32, device=dev)          # d mu1, d logvar1, d mu2, d logvar2
        _lib.call("glx_cvae_losses", pred, labels[:, :7].contiguous(), cw, B, bins, ctypes.c_float(1.0 / 9.0), ctypes.c_float(loc_weight),
                  ctypes.c_float(dir_weight), ctypes.c_float(dir_offset), mu1.contiguous(), lv1.contiguous(), mu2.contiguous(),
                  lv2.contiguous(), L, ctypes.c_float(latent_weight), out, d_pred, d_lat[0], d_lat[1], d_lat[2], d_lat[3])
        ctx.save_for_backward(d_pred, d_lat)
        ctx.bins = bins
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_loc, g_dir, g_lat):
        d_pred, d_lat = ctx.saved_tensors
        gp = torch.cat([g_loc.reshape(1).expand(7), g_dir.reshape(1).expand(ctx.bins)])
        dl = d_lat * g_lat
        return d_pred * gp, None, dl[0], dl[1], dl[2], dl[3], None, None, None, None, None, None


def l2_regularisation(module):
    """Sum of the 2-norms of a module's parameter tensors (model.py:20-28) -- one fused norm launch."""
    return torch.stack(torch._foreach_norm([p for p in module.parameters()], 2)).sum()


def cvae_direction_target(labels, dir_offset, num_bins):
    """Bin of the label heading (Generator.get_direction_target, model.py:278-294), as class indices."""
    offset_rot = limit_period(labels[..., 6] - dir_offset, 0, 2 * np.pi)
    return torch.clamp(torch.floor(offset_rot / (2 * np.pi / num_bins)).long(), min=0, max=num_bins - 1)


_CW_CACHE = {}


def _code_weights(values, device):
    """Device copy of the code weights, made once per (values, device): a host-to-device copy inside a recorded step
    is not capturable (the first, eager call of a step fills the cache)."""
    key = (values, str(device))
    if key not in _CW_CACHE:
        _CW_CACHE[key] = torch.tensor(values, dtype=torch.float32, device=device)
    return _CW_CACHE[key]


def cvae_reg_loss(box_preds, labels, weights, dir_offset, num_bins, beta=1.0 / 9.0):
    """Generator.reg_loss (model.py:296-345): code-weighted smooth-L1 (beta 1/9, loss_utils.py:74-141) on the seven box
    terms with the sin-difference heading encoding, summed over the batch and divided by it, plus the direction
    cross-entropy.  The reference hands WeightedCrossEntropyLoss a (B, 1, 2) weight tensor of ones where it expects
    (B, 1) (model.py:334-335): the (B, 1) losses broadcast to (B, B, 2), so `.sum() / batch_size` is TWICE the summed
    cross-entropy -- restated as such, it is part of the trained objective."""
    b = box_preds.shape[0]
    pred, tgt = box_preds[:, :7], labels[:, :7]
    sin_p = torch.sin(pred[:, 6:7]) * torch.cos(tgt[:, 6:7])
    sin_t = torch.cos(pred[:, 6:7]) * torch.sin(tgt[:, 6:7])
    pred = torch.cat([pred[:, :6], sin_p], dim=1)
    tgt = torch.cat([tgt[:, :6], sin_t], dim=1)
    tgt = torch.where(torch.isnan(tgt), pred, tgt)
    cw = _code_weights(tuple(weights["code_weights"]), pred.device)
    n = torch.abs((pred - tgt) * cw)
    loc = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta).sum() / b * weights["loc_weight"]
    dir_t = cvae_direction_target(labels, dir_offset, num_bins)
    ce = F.cross_entropy(box_preds[:, -num_bins:], dir_t, reduction="none")
    dir_loss = (2.0 * ce.sum()) * weights["dir_weight"]
    return loc + dir_loss, {"loss_loc": loc, "loss_dir": dir_loss, "loss_reg": loc + dir_loss}
