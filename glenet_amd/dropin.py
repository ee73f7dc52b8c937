"""Register glenet_amd under the import names the reference uses, so GLENet's own Python
(pcdet/models/backbones_3d/spconv_backbone.py, pcdet/ops/*/ *_utils.py, tools/train.py) runs
unmodified on the MI355X kernels.

    import glenet_amd.dropin; glenet_amd.dropin.install()
    # from here on:  import spconv.pytorch as spconv            -> glenet_amd.spconv
    #                from pcdet.ops.iou3d_nms import iou3d_nms_cuda -> glenet_amd.pcdet_ops...

Only the compiled-extension module names are aliased for pcdet.ops (the reference's *_utils.py
wrappers stay the reference's own files and call into these); `spconv` and the small part of
`cumm.tensorview` that data_processor.py:55 touches are provided whole.
"""
import importlib
import sys
import types

_EXT_MODULES = {
    "pcdet.ops.iou3d_nms.iou3d_nms_cuda": "glenet_amd.pcdet_ops.iou3d_nms.iou3d_nms_cuda",
    "pcdet.ops.iou3d.iou3d_cuda": "glenet_amd.pcdet_ops.iou3d.iou3d_cuda",
    "pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda": "glenet_amd.pcdet_ops.roiaware_pool3d.roiaware_pool3d_cuda",
    "pcdet.ops.roipoint_pool3d.roipoint_pool3d_cuda": "glenet_amd.pcdet_ops.roipoint_pool3d.roipoint_pool3d_cuda",
    "pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda":
        "glenet_amd.pcdet_ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda",
    "pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda":
        "glenet_amd.pcdet_ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda",
}
_SPCONV = {
    "spconv": "glenet_amd.spconv",
    "spconv.pytorch": "glenet_amd.spconv.pytorch",
    "spconv.conv": "glenet_amd.spconv.conv",
    "spconv.pytorch.conv": "glenet_amd.spconv.conv",
    "spconv.utils": "glenet_amd.spconv.utils",
    "spconv.pytorch.utils": "glenet_amd.spconv.utils",
}


def _tensorview_module():
    tv = types.ModuleType("cumm.tensorview")

    class _Array:
        def __init__(self, a):
            self._a = a

        def numpy(self):
            return self._a

        def numpy_view(self):
            return self._a

    tv.from_numpy = lambda a: _Array(a)
    tv.Tensor = _Array
    return tv


def install(spconv=True, ops=True, overwrite=False):
    """Alias the modules.  Returns the list of names that were registered."""
    done = []
    table = {}
    if spconv:
        table.update(_SPCONV)
    if ops:
        table.update(_EXT_MODULES)
    for alias, target in table.items():
        if alias in sys.modules and not overwrite:
            continue
        sys.modules[alias] = importlib.import_module(target)
        done.append(alias)
    if spconv and ("cumm" not in sys.modules or overwrite):
        cumm = types.ModuleType("cumm")
        cumm.tensorview = _tensorview_module()
        sys.modules["cumm"] = cumm
        sys.modules["cumm.tensorview"] = cumm.tensorview
        done += ["cumm", "cumm.tensorview"]
    return done


def accelerate(model, channels_last=True):
    """Give a network built from the REFERENCE'S OWN classes (pcdet.models.build_network over install()) the fast paths of
    this package, in place, without touching its parameters or state-dict keys.  What is re-classed / patched, each only
    where the instance has exactly the attributes the replacement reads (anything else is left as it is):

      BaseBEVBackbone            -> dense_path.BEVBackbone: 3x3 / strided / transposed convolutions on csrc/glx_conv2d.hip +
        (base_bev_backbone.py)      glx_deconv2d.hip (fp32 products on the bf16 matrix pipe), training BatchNorm on
                                    csrc/glx_bn.hip with the statistics in the conv epilogues, eval BatchNorm folded
      HeightCompression          -> backbone.HeightCompression: channels-last BEV map, or no map at all when the BEV
        (height_compression.py)     backbone above was re-classed (its first layer then runs on the sparse tensor)
      NeighborVoxelSAModuleMSG   -> pcdet_ops...voxel_pool_modules.NeighborVoxelSAModuleMSG: the row-major training path /
        (voxel_pool_modules.py)     the fused inference aggregation instead of (M, C, nsample) grouped tensors
      VoxelRCNNHead.roi_grid_pool   (voxelrcnn_head.py:106-191) -> roi_grid.RoIGridPool on the head's own layers: grid
                                    points in one kernel, the query through the sparse tensor's cell index (no dense -1
                                    voxel->point map, common_utils.py:226-243), no per-frame Python loops
      RoIHeadTemplate.proposal_layer (roi_head_template.py:52-128) -> detector.proposal_layer: batched top-k + NMS on the
                                    device for NMS_TYPE nms_gpu without MULTI_CLASSES_NMS (other settings: the original)
      ProposalTargetLayer        -> roi_targets.ProposalTargetLayer (matching + sampling + gathers in three launches)
        (proposal_target_layer.py)

    The sparse backbone needs nothing: its layers are spconv.SparseSequential(conv, BatchNorm1d, ReLU) and SparseSequential
    is ours (conv + BatchNorm + ReLU fused where the kernels cover the layer).  The anchor target assigner and the loss
    functions stay the reference's tensor statements (glenet_amd.target_assign / losses are called by glenet_amd.glenet_vr).
    channels_last: filters of the BEV backbone are moved to channels-last memory and the incoming `spatial_features` map is
    converted on entry (one copy); everything downstream sees logical NCHW tensors as before.
    Returns the names of what it changed.  On CPU tensors every re-classed module runs its layers one by one exactly as
    the reference does (tools/ref_dropin_check.py checks that on the reference's own GLENet-VR network)."""
    import types

    import torch
    from . import backbone as gb
    from . import dense_path as dp
    from . import roi_grid as rg
    from . import roi_targets as rt
    from .pcdet_ops.pointnet2.pointnet2_stack import voxel_pool_modules as vpm
    changed = []
    bev_ours = False
    for name, m in model.named_modules():
        if type(m).__name__ == "BaseBEVBackbone" and not isinstance(m, dp.BEVBackbone) \
                and isinstance(getattr(m, "blocks", None), torch.nn.ModuleList) \
                and isinstance(getattr(m, "deblocks", None), torch.nn.ModuleList):
            m.__class__ = dp.BEVBackbone
            if channels_last:
                m.to(memory_format=torch.channels_last)
                m.convert_input = True
            changed.append(name)
        bev_ours = bev_ours or isinstance(m, dp.BEVBackbone)
    for name, m in model.named_modules():
        cls = type(m).__name__
        if cls == "HeightCompression" and not isinstance(m, gb.HeightCompression) and hasattr(m, "num_bev_features"):
            m.__class__ = gb.HeightCompression
            m.channels_last = bool(channels_last)
            m.defer = bool(channels_last) and bev_ours and dp.SPARSE_FIRST_BEV_LAYER
            changed.append(name)
        elif cls == "NeighborVoxelSAModuleMSG" and not isinstance(m, vpm.NeighborVoxelSAModuleMSG) \
                and all(isinstance(getattr(m, a, None), torch.nn.ModuleList) for a in ("groupers", "mlps_in", "mlps_pos", "mlps_out")) \
                and all(hasattr(g, a) for g in m.groupers for a in ("max_range", "radius", "nsample")) \
                and hasattr(m, "pool_method"):
            m.__class__ = vpm.NeighborVoxelSAModuleMSG
            changed.append(name)
        elif cls == "ProposalTargetLayer" and not isinstance(m, rt.ProposalTargetLayer) and hasattr(m, "roi_sampler_cfg"):
            m.__class__ = rt.ProposalTargetLayer
            changed.append(name)
    for name, m in model.named_modules():
        layers = getattr(m, "roi_grid_pool_layers", None)
        cfg = getattr(m, "pool_cfg", None)
        if callable(getattr(m, "roi_grid_pool", None)) and isinstance(layers, torch.nn.ModuleList) and cfg is not None \
                and all(isinstance(l, vpm.NeighborVoxelSAModuleMSG) for l in layers) \
                and hasattr(m, "voxel_size") and hasattr(m, "point_cloud_range") and "_glx_pool" not in m.__dict__:
            pool = rg.RoIGridPool.__new__(rg.RoIGridPool)
            torch.nn.Module.__init__(pool)
            pool.grid_size = int(cfg.GRID_SIZE if hasattr(cfg, "GRID_SIZE") else cfg["GRID_SIZE"])
            pool.voxel_size, pool.point_cloud_range = [float(v) for v in m.voxel_size], [float(v) for v in m.point_cloud_range]
            pool.sources = list(cfg.FEATURES_SOURCE if hasattr(cfg, "FEATURES_SOURCE") else cfg["FEATURES_SOURCE"])
            pool.__dict__["roi_grid_pool_layers"] = layers              # shared, NOT registered twice: the keys stay the head's
            pool.num_features = sum(seq[0].out_channels for l in layers for seq in l.mlps_out)
            m.__dict__["_glx_pool"] = pool

            def roi_grid_pool(self, batch_dict):
                p = self.__dict__["_glx_pool"]
                p.training = self.training
                return p.forward(batch_dict["rois"], batch_dict["multi_scale_3d_features"],
                                 batch_dict["multi_scale_3d_strides"], batch_dict["batch_size"])
            m.roi_grid_pool = types.MethodType(roi_grid_pool, m)
            changed.append(name + ".roi_grid_pool")
        if callable(getattr(m, "proposal_layer", None)) and hasattr(m, "model_cfg") and "_glx_proposal_layer" not in m.__dict__ \
                and type(m).__name__.endswith("Head"):
            original = m.proposal_layer

            def proposal_layer(self, batch_dict, nms_config, _original=original):
                from . import detector as det
                boxes, cls = batch_dict.get("batch_box_preds"), batch_dict.get("batch_cls_preds")
                get = (lambda k, d=None: nms_config.get(k, d)) if hasattr(nms_config, "get") else (lambda k, d=None: getattr(nms_config, k, d))
                if (batch_dict.get("rois") is not None or batch_dict.get("batch_index") is not None or boxes is None
                        or not boxes.is_cuda or boxes.dim() != 3 or get("MULTI_CLASSES_NMS", False)
                        or get("NMS_TYPE") != "nms_gpu"):
                    return _original(batch_dict, nms_config=nms_config)
                with torch.no_grad():
                    rois, scores, labels = det.proposal_layer(boxes, cls, int(get("NMS_PRE_MAXSIZE")), int(get("NMS_POST_MAXSIZE")),
                                                              float(get("NMS_THRESH")),
                                                              normalized=bool(batch_dict.get("cls_preds_normalized", False)))
                batch_dict["rois"], batch_dict["roi_scores"], batch_dict["roi_labels"] = rois, scores, labels
                batch_dict["has_class_labels"] = cls.shape[-1] > 1
                batch_dict.pop("batch_index", None)
                return batch_dict
            m.__dict__["_glx_proposal_layer"] = True
            m.proposal_layer = types.MethodType(proposal_layer, m)
            changed.append(name + ".proposal_layer")
    return changed


_POINTWISE = {}


def pointwise_as_gemm(enable=True):
    """Opt-in, process-wide: torch.nn.Conv1d / Conv2d modules with a 1 x 1 kernel (stride 1, no padding, dilation 1, groups 1,
    zero padding mode) compute their output as ONE batched matrix product (W (Cout, Cin) @ x (B, Cin, L)) on device tensors
    instead of calling the vendor convolution.  The reference's RoI head feeds such layers tensors whose LENGTH is the number
    of active voxels / grid points of the batch (voxel_pool_modules.py:70-130: (1, C, M) and (1, C, M, nsample) inputs of
    `mlps_in` / `mlps_pos` / `mlps_out`), a new value every training step -- and MIOpen prepares a convolution per problem
    size (measured: 1.04 s per step with a new voxel count against 33 ms with a repeated one); a matrix product has no
    per-shape preparation.  Same arithmetic (fp32 dot products over Cin, bias added afterwards), autograd through torch's
    matmul.  Everything else -- other kernel sizes, CPU tensors -- takes the original forward.  BatchNorm1d / BatchNorm2d in
    training mode on such stacked tensors run on this package's channel-major kernels (glx_bn_cm_*: the vendor's BatchNorm
    builds a kernel per problem size as well); eval mode, image batches and (rows, C) inputs keep the original forward.
    Returns the list of patched classes; pointwise_as_gemm(False) restores the originals."""
    import torch
    import torch.nn.functional as F
    from torch import nn
    if not enable:
        for cls, orig in _POINTWISE.items():
            cls.forward = orig
        done = list(_POINTWISE)
        _POINTWISE.clear()
        return done

    def plain(m):
        n = len(m.kernel_size)
        return (m.kernel_size == (1,) * n and m.stride == (1,) * n and m.dilation == (1,) * n and m.groups == 1
                and m.padding in ((0,) * n, "valid") and m.padding_mode == "zeros")

    def make(orig):
        def forward(self, x):
            if not (x.is_cuda and plain(self) and x.dim() == len(self.kernel_size) + 2 and x.dtype == self.weight.dtype):
                return orig(self, x)
            b, c = x.shape[0], x.shape[1]
            w = self.weight.reshape(self.out_channels, c)
            y = torch.matmul(w, x if x.dim() == 3 else x.reshape(b, c, -1))   # (B, Cout, L); strided (1, C, M) views as they are
            if self.bias is not None:
                y = y + self.bias.view(1, -1, 1)
            return y.reshape((b, self.out_channels) + tuple(x.shape[2:]))
        return forward

    for cls in (nn.Conv1d, nn.Conv2d):
        if cls not in _POINTWISE:
            _POINTWISE[cls] = cls.forward
            cls.forward = make(cls.forward)
    def make_bn(orig):
        # The vendor's BatchNorm prepares per problem size too -- and on first sight BUILDS a kernel: 0.45 s per new voxel count
        # in a fresh process (round 5's bench: 508 ms per step with only the convolutions patched).  Stacked tensors (batch
        # dimension 1) in training mode take the channel-major kernels of this package (spconv.core.StackedBN); torch's native
        # batch_norm kernel (43 ms per step) and tensor statements (47 ms) were measured and are slower on these 200 MB tensors.
        def forward(self, x):
            from .spconv import core
            y = core.stacked_train_bn(self, x) if isinstance(x, torch.Tensor) else None
            return orig(self, x) if y is None else y
        return forward

    for cls in (nn.BatchNorm1d, nn.BatchNorm2d):
        if cls not in _POINTWISE:
            _POINTWISE[cls] = cls.forward
            cls.forward = make_bn(cls.forward)
    return list(_POINTWISE)


class reference_layout:
    """Context manager: the call sequence a network in the REFERENCE'S module layout makes through the drop-in WITHOUT
    accelerate() -- every switch that selects a fused / batched path of this package off, so that glenet_amd's own modules
    run layer by layer through the same `*_utils` wrappers, torch modules and per-frame loops the reference's Python drives:
    vendor (MIOpen) 2-D convolutions and torch BatchNorm in the BEV backbone, the dense BEV map, the per-frame proposal loop
    with its read-backs, RoI-grid pooling through VoxelQueryAndGrouping / grouping_operation and Conv1d / Conv2d modules, the
    loss terms as tensor statements.  (The sparse convolutions keep their conv + BatchNorm + ReLU fusion: that lives in
    spconv.SparseSequential, which the reference's backbone instantiates itself.)  bench.py times a training step under
    it (`dropin_step`) beside the same step with the fast paths (`dropin_accelerated_step`) and the recorded headline."""

    SWITCHES = (("glenet_amd.dense_path", "OWN_CONV3X3", False), ("glenet_amd.dense_path", "OWN_DECONV", False),
                ("glenet_amd.dense_path", "OWN_STRIDED_FORWARD", False), ("glenet_amd.dense_path", "SPARSE_FIRST_BEV_LAYER", False),
                ("glenet_amd.detector", "BATCHED_PROPOSALS", False), ("glenet_amd.detector", "FUSED_PREDICTED_BOXES", False),
                ("glenet_amd.detector", "FUSED_TOPK", False), ("glenet_amd.roi_targets", "FUSED_GATHER", False))
    CLASS_SWITCHES = (("glenet_amd.roi_grid", "RoIGridPool", "USE_ROWS", False), ("glenet_amd.roi_grid", "RoIGridPool", "USE_FUSED", False),
                      ("glenet_amd.pcdet_ops.pointnet2.pointnet2_stack.voxel_pool_modules", "NeighborVoxelSAModuleMSG", "USE_FUSED", False),
                      ("glenet_amd.pcdet_ops.pointnet2.pointnet2_stack.voxel_pool_modules", "NeighborVoxelSAModuleMSG", "USE_ROW_MAJOR", False),
                      ("glenet_amd.dense_path", "AnchorHead", "FUSE_HEADS", False), ("glenet_amd.dense_path", "BEVBackbone", "FUSE_EVAL", False),
                      ("glenet_amd.dense_path", "BEVBackbone", "FUSE_UPS_CAT", False))

    def __enter__(self):
        self.saved = []
        for mod, name, val in self.SWITCHES:
            m = importlib.import_module(mod)
            if hasattr(m, name):
                self.saved.append((m, name, getattr(m, name)))
                setattr(m, name, val)
        for mod, cls, name, val in self.CLASS_SWITCHES:
            c = getattr(importlib.import_module(mod), cls, None)
            if c is not None and hasattr(c, name):
                self.saved.append((c, name, getattr(c, name)))
                setattr(c, name, val)
        return self

    def __exit__(self, *exc):
        for obj, name, val in reversed(self.saved):
            setattr(obj, name, val)
        return False
