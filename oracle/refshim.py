"""TEST INFRASTRUCTURE -- never imported by the product (glenet_amd/), by smoke() or by the timed part of bench.py.

CPU stand-ins, BACKED BY THE ORACLE, for the compiled modules the reference's Python imports, so that the
reference's OWN network classes (VoxelRCNN, VoxelBackBone8x, VoxelRCNNKLLabelIoUHead, NeighborVoxelSAModuleMSG,
ProposalTargetLayer, AxisAlignedTargetAssigner, Detector3DTemplate.post_processing ...) can be executed in the
authoring container, which has no GPU, no spconv and no CUDA toolchain, and their every intermediate stored as a
golden fixture (tests/golden/make_golden.py refstep -> tests/golden/ref_step.npz).

What is provided (module name the reference imports -> what answers here):
  spconv / spconv.pytorch / spconv.conv / spconv.utils    SparseConvTensor, SubMConv3d, SparseConv3d,
        SparseInverseConv3d, SparseSequential, SparseModule, SparseConvolution over oracle.build_rules +
        orc_sconv_forward / orc_sconv_backward (oracle/glenet_oracle.c), autograd through a torch Function.
        Weight layout (kd, kh, kw, Cin, Cout) = spconv 1.x = glenet_amd.spconv, so state dicts move across unchanged.
  pcdet.ops.iou3d_nms.iou3d_nms_cuda                      iou3d_nms_api.cpp:11-17 signatures over oracle.boxes_* / nms_sorted
  pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda pointnet2_api.cpp:12-31: voxel_query_wrapper,
        group_points_wrapper, group_points_grad_wrapper, ball_query_wrapper (the ones GLENet's configs reach)
  pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda          points_in_boxes_cpu (augmentation helpers import it)
  the three remaining extension names                     importable, every call raises

The reference's wrappers allocate with `torch.cuda.IntTensor(..)` / `torch.cuda.FloatTensor(..)` and call `.cuda()`:
`cpu_placeholders()` below maps those to their host twins for the duration of a `with` block.  Disclosed in the
fixture's header as the existing goldens disclose theirs.
"""
import contextlib
import ctypes
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

import oracle


# ------------------------------------------------------------------------------------------------ spconv surface
def _triple(v):
    return tuple(int(x) for x in v) if isinstance(v, (list, tuple)) else (int(v),) * 3


class SparseConvTensor:
    """spconv.SparseConvTensor as spconv_backbone.py:141-146 constructs it and height_compression.py:21 / voxelrcnn_head.py
    read it: features (N,C) f32, indices (N,4) int32 [b,z,y,x]."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None, indice_dict=None,
                 benchmark=False):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = indice_dict if indice_dict is not None else {}
        self.grid, self.voxel_num, self.benchmark = grid, voxel_num, benchmark

    def replace_feature(self, new_features):
        t = SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size, self.grid, self.voxel_num,
                             self.indice_dict, self.benchmark)
        return t

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        return None if key is None else self.indice_dict.get(key)

    def dense(self, channels_first=True):
        """(B, C, D, H, W): scatter of the rows (differentiable: index_put on a zero tensor)."""
        idx = self.indices.long()
        d, h, w = self.spatial_shape
        out = self.features.new_zeros((self.batch_size, d, h, w, self.features.shape[1]))
        out = out.index_put((idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]), self.features)
        return out.permute(0, 4, 1, 2, 3).contiguous() if channels_first else out


class _OracleConv(torch.autograd.Function):
    """out = oracle.sconv_forward(features, W, rules); backward = oracle.sconv_backward (input and weight gradients)."""

    @staticmethod
    def forward(ctx, features, weight_kio, rules):
        ctx.rules = rules
        ctx.save_for_backward(features, weight_kio)
        out = oracle.sconv_forward(features.detach().numpy(), weight_kio.detach().numpy(), rules)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, grad_out):
        features, weight_kio = ctx.saved_tensors
        din, dw = oracle.sconv_backward(features.detach().numpy(), weight_kio.detach().numpy(),
                                        np.ascontiguousarray(grad_out.detach().numpy()), ctx.rules)
        return torch.from_numpy(din), torch.from_numpy(dw), None


class _Swapped:
    """The rule table of a strided convolution read backwards (SparseInverseConv3d: outputs are the forward conv's inputs)."""

    def __init__(self, r, in_indices):
        self.pairs_in, self.pairs_out, self.n_pairs = r.pairs_out, r.pairs_in, r.n_pairs
        self.out_indices, self.n_in = in_indices, len(r.out_indices)


class SparseModule(nn.Module):
    pass


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 fused_bn=False, use_hash=False, algo=None):
        super().__init__()
        assert ndim == 3 and groups == 1 and _triple(dilation) == (1, 1, 1)
        self.ndim, self.in_channels, self.out_channels = ndim, in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        self.subm, self.inverse, self.indice_key = subm, inverse, indice_key
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)
        fan_in = int(np.prod(self.kernel_size)) * in_channels
        with torch.no_grad():
            self.weight.uniform_(-1.0 / np.sqrt(fan_in), 1.0 / np.sqrt(fan_in))

    def forward(self, x):
        assert isinstance(x, SparseConvTensor)
        K = int(np.prod(self.kernel_size))
        w = self.weight.reshape(K, self.in_channels, self.out_channels)
        key = self.indice_key
        if self.inverse:
            r, in_idx, in_shape = x.indice_dict[key]
            rules, out_idx, out_shape = _Swapped(r, in_idx.numpy()), in_idx, in_shape
        else:
            hit = x.find_indice_pair(key)
            if hit is not None:
                r = hit[0]
                assert r.n_in == x.indices.shape[0], (key, r.n_in, x.indices.shape)
            else:
                r = oracle.build_rules(x.indices.numpy(), x.spatial_shape, self.kernel_size, self.stride, self.padding,
                                       subm=self.subm)
                if key is not None:
                    x.indice_dict[key] = (r, x.indices, list(x.spatial_shape))
            rules = r
            out_idx = x.indices if self.subm else torch.from_numpy(np.ascontiguousarray(r.out_indices))
            out_shape = x.spatial_shape if self.subm else list(r.out_shape)
        feats = _OracleConv.apply(x.features.contiguous().float(), w.contiguous(), rules)
        if self.bias is not None:
            feats = feats + self.bias
        return SparseConvTensor(feats, out_idx, out_shape, x.batch_size, x.grid, x.voxel_num, x.indice_dict, x.benchmark)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, use_hash=False, algo=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, use_hash=False, algo=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, False,
                         indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key=None, bias=True, algo=None):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key)


class SparseSequential(SparseModule):
    """spconv.SparseSequential: sparse modules take the tensor, dense nn.Modules take `.features` (spconv_backbone.py:21-25)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for k, m in args[0].items():
                self.add_module(k, m)
        else:
            for i, m in enumerate(args):
                self.add_module(str(i), m)
        for k, m in kwargs.items():
            self.add_module(k, m)

    def __getitem__(self, idx):
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        self.add_module(name if name is not None else str(len(self._modules)), module)

    def forward(self, x):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                x = m(x)
            elif isinstance(x, SparseConvTensor):
                if x.indices.shape[0] != 0:
                    x = x.replace_feature(m(x.features))
            else:
                x = m(x)
        return x


def _spconv_modules():
    names = dict(SparseConvTensor=SparseConvTensor, SparseModule=SparseModule, SparseConvolution=SparseConvolution,
                 SubMConv3d=SubMConv3d, SparseConv3d=SparseConv3d, SparseInverseConv3d=SparseInverseConv3d,
                 SparseSequential=SparseSequential)
    top = types.ModuleType("spconv")
    pyt = types.ModuleType("spconv.pytorch")
    conv = types.ModuleType("spconv.conv")
    utils = types.ModuleType("spconv.utils")
    for m in (top, pyt):
        m.__dict__.update(names)
        m.conv, m.utils = conv, utils
        m.__path__ = []
    conv.__dict__.update({k: names[k] for k in ("SparseConvolution", "SubMConv3d", "SparseConv3d", "SparseInverseConv3d")})
    top.pytorch = pyt
    top.__version__ = "2.1.0+oracle_refshim"
    return {"spconv": top, "spconv.pytorch": pyt, "spconv.conv": conv, "spconv.pytorch.conv": conv,
            "spconv.utils": utils, "spconv.pytorch.utils": utils}


# ------------------------------------------------------------------------------------------------ compiled-extension names
def _np32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


def _iou3d_nms_module():
    m = types.ModuleType("pcdet.ops.iou3d_nms.iou3d_nms_cuda")

    def boxes_overlap_bev_gpu(a, b, out):                        # iou3d_nms.cpp:49-68
        out.copy_(torch.from_numpy(oracle.boxes_overlap_bev(_np32(a), _np32(b))).reshape(out.shape))
        return 1

    def boxes_iou_bev_gpu(a, b, out):                            # iou3d_nms.cpp:70-88
        out.copy_(torch.from_numpy(oracle.boxes_iou_bev(_np32(a), _np32(b))).reshape(out.shape))
        return 1

    def boxes_iou_bev_cpu(a, b, out):                            # iou3d_cpu.cpp:232-252
        out.copy_(torch.from_numpy(oracle.boxes_iou_bev(_np32(a), _np32(b))).reshape(out.shape))
        return 1

    def _nms(boxes, keep, thresh, normal):                       # iou3d_nms.cpp:90-186: boxes sorted by the caller
        k = oracle.nms_sorted(_np32(boxes), float(thresh), normal=normal)
        keep[:len(k)] = torch.from_numpy(np.asarray(k, dtype=np.int64))
        return len(k)

    m.boxes_overlap_bev_gpu, m.boxes_iou_bev_gpu, m.boxes_iou_bev_cpu = boxes_overlap_bev_gpu, boxes_iou_bev_gpu, boxes_iou_bev_cpu
    m.nms_gpu = lambda boxes, keep, thresh: _nms(boxes, keep, thresh, False)
    m.nms_normal_gpu = lambda boxes, keep, thresh: _nms(boxes, keep, thresh, True)
    return m


def _pointnet2_stack_module():
    m = types.ModuleType("pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda")
    f, i = oracle._f, oracle._i

    def voxel_query_wrapper(M, Z, Y, X, nsample, radius, z_range, y_range, x_range, new_xyz, xyz, new_coords,
                            point_indices, idx):                 # voxel_query.cpp:28-44 (raw: -1 marks an empty query)
        nx, x_, nc, pi = _np32(new_xyz), _np32(xyz), oracle._i32(new_coords.numpy()), oracle._i32(point_indices.numpy())
        out = np.zeros((M, nsample), np.int32)
        oracle.lib().orc_voxel_query(M, Z, Y, X, nsample, ctypes.c_float(radius), z_range, y_range, x_range, f(nx), f(x_),
                                     i(nc), i(pi), i(out))
        idx.copy_(torch.from_numpy(out))
        return 1

    def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
        out.copy_(torch.from_numpy(oracle.group_points(_np32(features), features_batch_cnt.numpy(), idx.numpy(),
                                                       idx_batch_cnt.numpy())))
        return 1

    def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
        grad_features.copy_(torch.from_numpy(oracle.group_points_grad(_np32(grad_out), idx.numpy(), idx_batch_cnt.numpy(),
                                                                      features_batch_cnt.numpy(), N)))
        return 1

    def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
        out = np.zeros((M, nsample), np.int32)
        oracle.lib().orc_ball_query(B, M, ctypes.c_float(radius), nsample, f(_np32(new_xyz)),
                                    i(oracle._i32(new_xyz_batch_cnt.numpy())), f(_np32(xyz)),
                                    i(oracle._i32(xyz_batch_cnt.numpy())), i(out))
        idx.copy_(torch.from_numpy(out))
        return 1

    m.voxel_query_wrapper, m.group_points_wrapper = voxel_query_wrapper, group_points_wrapper
    m.group_points_grad_wrapper, m.ball_query_wrapper = group_points_grad_wrapper, ball_query_wrapper
    return m


def _roiaware_module():
    m = types.ModuleType("pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda")

    def points_in_boxes_cpu(boxes, pts, out):                    # roiaware_pool3d.cpp:142-168
        out.copy_(torch.from_numpy(oracle.points_in_boxes_cpu(_np32(pts), _np32(boxes))).reshape(out.shape))
        return 1
    m.points_in_boxes_cpu = points_in_boxes_cpu
    return m


class _Unreached(types.ModuleType):
    """An extension module no GLENet config reaches: importable, calling anything in it is an error."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def fail(*a, **k):
            raise NotImplementedError("%s.%s is not served by the oracle shim" % (self.__name__, name))
        return fail


def modules():
    table = _spconv_modules()
    table["pcdet.ops.iou3d_nms.iou3d_nms_cuda"] = _iou3d_nms_module()
    table["pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda"] = _pointnet2_stack_module()
    table["pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda"] = _roiaware_module()
    for name in ("pcdet.ops.iou3d.iou3d_cuda", "pcdet.ops.roipoint_pool3d.roipoint_pool3d_cuda",
                 "pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda"):
        table[name] = _Unreached(name)
    cumm = types.ModuleType("cumm")
    cumm.tensorview = types.ModuleType("cumm.tensorview")
    table["cumm"], table["cumm.tensorview"] = cumm, cumm.tensorview
    return table


def install():
    """Register the stand-ins under the names the reference imports (instead of glenet_amd.dropin.install())."""
    table = modules()
    sys.modules.update(table)
    return sorted(table)


@contextlib.contextmanager
def cpu_placeholders():
    """`.cuda()` is the identity, torch.cuda.{Int,Float,Long}Tensor are their host twins, while the reference runs."""
    saved = (torch.Tensor.cuda, nn.Module.cuda, torch.cuda.IntTensor, torch.cuda.FloatTensor, torch.cuda.LongTensor)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.IntTensor, torch.cuda.FloatTensor, torch.cuda.LongTensor = torch.IntTensor, torch.FloatTensor, torch.LongTensor
    try:
        yield
    finally:
        (torch.Tensor.cuda, nn.Module.cuda, torch.cuda.IntTensor, torch.cuda.FloatTensor, torch.cuda.LongTensor) = saved
