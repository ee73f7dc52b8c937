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


# ------------------------------------------------------------------------------------------------ the recorded training step
def _cfg_get(c, key, default=None):
    if c is None:
        return default
    if isinstance(c, dict):
        return c.get(key, default)
    return getattr(c, key, default)


def _translate_cfg(model_cfg):
    """(roi_cfg, head_cfg) for glenet_vr.GLENetVR from the MODEL block of a GLENet-VR configuration as the reference's
    cfg_from_yaml_file leaves it (tools/cfgs/kitti_models/GLENet_VR.yaml:32-166); raises on anything the recorded step does
    not implement (another detector / head class, several anchor classes, multi-class NMS, iou3d loss, ...)."""
    from .glenet_vr import DENSE_HEAD_CFG, ROI_HEAD_CFG
    g = _cfg_get
    want = dict(NAME="VoxelRCNN")
    names = dict(VFE="MeanVFE", BACKBONE_3D="VoxelBackBone8x", MAP_TO_BEV="HeightCompression", BACKBONE_2D="BaseBEVBackbone",
                 DENSE_HEAD="AnchorHeadSingle", ROI_HEAD="VoxelRCNNKLLabelIoUHead")
    if g(model_cfg, "NAME") != want["NAME"]:
        raise NotImplementedError("dropin.record: the recorded step is GLENet-VR's (MODEL.NAME VoxelRCNN), got %r" % g(model_cfg, "NAME"))
    for block, name in names.items():
        if g(g(model_cfg, block), "NAME") != name:
            raise NotImplementedError("dropin.record: MODEL.%s.NAME is %r, the recorded step implements %r"
                                      % (block, g(g(model_cfg, block), "NAME"), name))
    b2 = g(model_cfg, "BACKBONE_2D")
    if (list(g(b2, "LAYER_NUMS")), list(g(b2, "LAYER_STRIDES")), list(g(b2, "NUM_FILTERS")), list(g(b2, "UPSAMPLE_STRIDES")),
            list(g(b2, "NUM_UPSAMPLE_FILTERS"))) != ([5, 5], [1, 2], [64, 128], [1, 2], [128, 128]):
        raise NotImplementedError("dropin.record: BACKBONE_2D layout differs from GLENet_VR.yaml:46-52")
    dh = g(model_cfg, "DENSE_HEAD")
    ag = list(g(dh, "ANCHOR_GENERATOR_CONFIG"))
    if len(ag) != 1 or not g(dh, "USE_DIRECTION_CLASSIFIER") or g(dh, "NUM_DIR_BINS") != 2 \
            or abs(float(g(dh, "DIR_OFFSET")) - 0.78539) > 1e-6 or float(g(dh, "DIR_LIMIT_OFFSET")) != 0.0:
        raise NotImplementedError("dropin.record: one anchor class with the two-bin direction classifier (GLENet_VR.yaml:54-73)")
    ta = g(dh, "TARGET_ASSIGNER_CONFIG")
    if g(ta, "NAME") != "AxisAlignedTargetAssigner" or g(ta, "NORM_BY_NUM_EXAMPLES") or g(ta, "MATCH_HEIGHT") \
            or float(g(ta, "POS_FRACTION")) >= 0:
        raise NotImplementedError("dropin.record: AxisAlignedTargetAssigner without sampling (GLENet_VR.yaml:75-81)")
    a = ag[0]
    lw = g(g(dh, "LOSS_CONFIG"), "LOSS_WEIGHTS")
    head_cfg = dict(DENSE_HEAD_CFG, anchor_sizes=[list(map(float, s)) for s in g(a, "anchor_sizes")],
                    anchor_rotations=[float(v) for v in g(a, "anchor_rotations")],
                    anchor_bottom_heights=[float(v) for v in g(a, "anchor_bottom_heights")],
                    matched_threshold=float(g(a, "matched_threshold")), unmatched_threshold=float(g(a, "unmatched_threshold")),
                    cls_weight=float(g(lw, "cls_weight")), loc_weight=float(g(lw, "loc_weight")),
                    dir_weight=float(g(lw, "dir_weight")), code_weights=[float(v) for v in g(lw, "code_weights")])
    if g(a, "align_center") or int(g(a, "feature_map_stride")) != 8:
        raise NotImplementedError("dropin.record: anchors on the stride-8 map, align_center False")
    rh = g(model_cfg, "ROI_HEAD")
    pool = g(rh, "ROI_GRID_POOL")
    layers = g(pool, "POOL_LAYERS")
    POOL = {}
    for src in g(pool, "FEATURES_SOURCE"):
        lc = g(layers, src)
        if g(lc, "POOL_METHOD") != "max_pool":
            raise NotImplementedError("dropin.record: POOL_METHOD max_pool")
        # (c_mid, c_out) per group: VoxelRCNNHead.__init__ prepends the source's channel count to these lists IN PLACE
        # (voxelrcnn_head.py:26-28), so a configuration that has built a network carries three entries
        POOL[src] = dict(mlps=[list(m)[-2:] for m in g(lc, "MLPS")], query_ranges=[list(q) for q in g(lc, "QUERY_RANGES")],
                         radii=[float(v) for v in g(lc, "POOL_RADIUS")], nsamples=[int(v) for v in g(lc, "NSAMPLE")])

    def nms(c):
        if g(c, "NMS_TYPE") != "nms_gpu" or g(c, "MULTI_CLASSES_NMS"):
            raise NotImplementedError("dropin.record: proposals by class-agnostic nms_gpu (GLENet_VR.yaml:101-115)")
        return (int(g(c, "NMS_PRE_MAXSIZE")), int(g(c, "NMS_POST_MAXSIZE")), float(g(c, "NMS_THRESH")))
    tc = g(rh, "TARGET_CONFIG")
    lc = g(rh, "LOSS_CONFIG")
    if g(lc, "CLS_LOSS") != "BinaryCrossEntropy" or g(lc, "REG_LOSS") != "smooth-l1" or not g(lc, "CORNER_LOSS_REGULARIZATION") \
            or g(lc, "GRID_3D_IOU_LOSS") or not g(rh, "CLASS_AGNOSTIC"):
        raise NotImplementedError("dropin.record: the KL head's loss configuration differs from GLENet_VR.yaml:155-166")
    w = g(lc, "LOSS_WEIGHTS")
    roi_cfg = dict(ROI_HEAD_CFG, POOL=POOL, GRID_SIZE=int(g(pool, "GRID_SIZE")), SHARED_FC=tuple(g(rh, "SHARED_FC")),
                   CLS_FC=tuple(g(rh, "CLS_FC")), REG_FC=tuple(g(rh, "REG_FC")), DP_RATIO=float(g(rh, "DP_RATIO")),
                   NMS_TRAIN=nms(g(g(rh, "NMS_CONFIG"), "TRAIN")), NMS_TEST=nms(g(g(rh, "NMS_CONFIG"), "TEST")),
                   TARGET={k: g(tc, k) for k in ROI_HEAD_CFG["TARGET"]},
                   LOSS_WEIGHTS=dict(rcnn_cls_weight=float(g(w, "rcnn_cls_weight")), rcnn_reg_weight=float(g(w, "rcnn_reg_weight")),
                                     rcnn_corner_weight=float(g(w, "rcnn_corner_weight")),
                                     code_weights=[float(v) for v in g(w, "code_weights")]))
    return roi_cfg, head_cfg


def _share_state(twin, model):
    """Make `twin` (a glenet_vr.GLENetVR) hold the SAME Parameter / buffer objects as `model` (the reference-built network):
    the state-dict keys are equal by construction (tests/golden/ref_state_keys.npz), so every entry is re-pointed by name.
    Returns the number of shared tensors; raises when a key or shape does not match."""
    import torch
    src = dict(model.named_parameters())
    srcb = {k: v for k, v in model.named_buffers() if k != "global_step"}      # the detector's own step counter
    mine = dict(twin.named_parameters())                                        # (detector3d_template.py:21, model_func bumps it)
    mineb = dict(twin.named_buffers())
    if set(src) != set(mine) or set(srcb) != set(mineb):
        odd = sorted((set(src) ^ set(mine)) | (set(srcb) ^ set(mineb)))
        raise ValueError("dropin.record: the network's state differs from GLENet-VR's in %d entries, e.g. %s" % (len(odd), odd[:6]))
    n = 0
    for name, p in list(src.items()) + list(srcb.items()):
        path = name.split(".")
        mod = twin
        for part in path[:-1]:
            mod = getattr(mod, part)
        table = mod._parameters if path[-1] in mod._parameters else mod._buffers
        old = table[path[-1]]
        if old is not None and tuple(old.shape) != tuple(p.shape):
            raise ValueError("dropin.record: %s is %s here, %s in the network" % (name, tuple(old.shape), tuple(p.shape)))
        table[path[-1]] = p
        n += 1
    assert all(a is b for a, b in zip(twin.parameters(), (dict(model.named_parameters())[k] for k, _ in twin.named_parameters())))
    return n


class RecordedStep:
    """What dropin.record() returns: a callable with the contract of the reference's network in training mode
    (pcdet/models/detectors/voxel_rcnn.py:forward: `ret_dict, tb_dict, disp_dict = model(batch_dict)`, consumed by
    pcdet/models/__init__.py:37-52 model_func and tools/train_utils/train_utils.py:45-76), backed by ONE recorded HIP graph of
    forward + backward on the network's OWN parameters:

        step = glenet_amd.dropin.record(model, batch_size=4, max_points=..., max_gt=...)
        for batch in loader:
            optimizer.zero_grad()                          # either form
            ret, tb, disp = step(batch)                    # copies the batch into the static inputs, replays the graph
            ret["loss"].backward()                         # hands every parameter its gradient (views of one flat buffer)
            clip_grad_norm_(model.parameters(), 10); optimizer.step()          # the caller's, untouched

    The graph runs glenet_amd.glenet_vr.GLENetVR's shape-static step (capacity-padded buffers, live counts on the device, the
    RoI branch on its own stream, weight gradients written in place) over Parameter objects SHARED with `model`; voxelization
    happens on the device from batch["points"] (the reference's DataProcessor voxels of the same points are bit-identical,
    tests/test_sparse_gpu.py).  Gradient accumulation over several calls is not supported (a replay overwrites the flat
    gradient buffer): call zero_grad() between steps as tools/train.py does."""

    def __init__(self, model, twin, pipe):
        import torch
        self.model, self.twin, self.pipe = model, twin, pipe
        self._anchor = torch.zeros((), device=pipe.points.device, requires_grad=True)
        self.params = pipe.flat_grads.params
        self.views = pipe.flat_grads.grad_views
        self.flat_grad = pipe.flat_grads.flat_grad

    def tensors(self, batch_dict):
        """(points (P, C), frame ids (P,) int32, gt_boxes (B, G, 8), gt_uncertaintys (B, G, 7) or None) on the device from a
        collated batch (pcdet/datasets/dataset.py:170-250: `points` (P, 1 + C) with the frame id in column 0)."""
        import torch
        dev = self.pipe.points.device
        pts = torch.as_tensor(batch_dict["points"]).to(dev, non_blocking=True)
        gt = torch.as_tensor(batch_dict["gt_boxes"]).to(dev, torch.float32, non_blocking=True).contiguous()
        unc = batch_dict.get("gt_uncertaintys")
        if unc is not None:
            unc = torch.as_tensor(unc).to(dev, torch.float32, non_blocking=True)[:, :, :7]
            unc = torch.where(gt[:, :, 7:8] > 0, unc, torch.zeros_like(unc)).contiguous()      # dataset.py:188 pads with -1
        return pts[:, 1:].float().contiguous(), pts[:, 0].to(torch.int32).contiguous(), gt, unc

    def load(self, batch_dict):
        self.pipe.load(*self.tensors(batch_dict))

    def __call__(self, batch_dict):
        import torch
        if not (self.model.training and torch.is_grad_enabled()):
            raise RuntimeError("dropin.record: the recorded step is the TRAINING step; run evaluation through the network itself")
        self.load(batch_dict)
        self.pipe.step()
        # (the pipeline's scalar: already detached after a staged backward, a consumed autograd graph behind it otherwise)
        loss = _recorded_loss().apply(self._anchor, self.pipe.loss.detach(), self)
        tb = {k: v for k, v in self.pipe.parts.items()}
        tb["rpn_loss"] = tb.get("loss_rpn")
        return {"loss": loss}, tb, {}

    def check(self):
        """One read-back: did the last batch fit the recorded capacities?  Raises (spconv.core.check_static) when a strided
        output set or the voxel cap overflowed -- that step's gradients are then not the batch's; returns True otherwise."""
        self.pipe.check()
        return True


def _recorded_loss_class():
    import torch

    class _RecordedLoss(torch.autograd.Function):
        """The scalar of a replayed step: backward() hands every parameter its gradient -- the views of the flat buffer the
        replay has filled -- scaled by the incoming gradient (1 for `loss.backward()`; `loss.mean()` of a scalar is 1 as well)."""

        @staticmethod
        def forward(ctx, anchor, value, rec):
            ctx.rec = rec
            return value.detach().clone()

        @staticmethod
        def backward(ctx, g):
            rec = ctx.rec
            rec.flat_grad.mul_(g)                                   # one launch on the flat buffer (g is a device scalar)
            for p, v in zip(rec.params, rec.views):
                if p.grad is None:
                    p.grad = v
                elif p.grad.data_ptr() != v.data_ptr():
                    p.grad.add_(v)
            return None, None, None
    return _RecordedLoss


_RL = None


def _recorded_loss():
    global _RL
    if _RL is None:
        _RL = _recorded_loss_class()
    return _RL


def record(model, batches=None, batch_size=None, max_points=None, max_gt=32, data_cfg=None, capacities=None, warmup=2,
           seed_rois_with_gt=None, dry_run=False):
    """Record the training step of a GLENet-VR network built by the REFERENCE'S OWN `build_network` (over install()) as one HIP
    graph of forward + backward on the network's own parameters, optimizer left to the caller: returns a RecordedStep.

    model        the network (pcdet.models.detectors.VoxelRCNN with the GLENet_VR.yaml module layout); its `model_cfg` and
                 `dataset` (point_cloud_range, voxel_size) are read, anything the recorded step does not implement raises
    batches      one collated batch_dict or several (`points` (P, 1 + C), `gt_boxes` (B, G, 8)[, `gt_uncertaintys` (B, G, 7)]):
                 representative inputs -- the warm-up passes run on the first, the strided output sets' capacities are sized
                 1.3 x the largest seen (backbone.StaticFramePipeline.calibrate; RecordedStep.check() tells whether a later
                 batch fitted), batch_size / max_points default to what they show (points + 10 %)
    max_gt       capacity of the ground-truth rows per frame
    data_cfg     optional dict(max_points=5, max_voxels_train=16000, num_features=4): DATA_PROCESSOR's voxel settings
                 (kitti_dataset.yaml:65-72); default: the dataset's own config when the network carries it, else those values
    dry_run      build the parameter-sharing twin and translate the configuration only (no device, no graph): what
                 tools/ref_dropin_check.py runs on CPU against the reference's own network
    What is changed on `model`: the 4-D filters of its BEV backbone and dense head move to channels-last memory (values and
    keys unchanged, as accelerate() does); nothing else."""
    import torch
    from . import glenet_vr as gvr
    roi_cfg, head_cfg = _translate_cfg(model.model_cfg)
    ds = getattr(model, "dataset", None)
    cfg = dict(point_cloud_range=[round(float(v), 5) for v in ds.point_cloud_range],      # the dataset keeps a float32 array
               voxel_size=[round(float(v), 6) for v in ds.voxel_size],
               max_points=5, max_voxels_train=16000, max_voxels_test=40000,
               num_features=int(getattr(getattr(ds, "point_feature_encoder", None), "num_point_features", 4)))
    dc = getattr(ds, "dataset_cfg", None)
    for p in (_cfg_get(dc, "DATA_PROCESSOR", None) or []):
        if _cfg_get(p, "NAME") == "transform_points_to_voxels":
            cfg["max_points"] = int(_cfg_get(p, "MAX_POINTS_PER_VOXEL"))
            mv = _cfg_get(p, "MAX_NUMBER_OF_VOXELS")
            cfg["max_voxels_train"], cfg["max_voxels_test"] = int(_cfg_get(mv, "train")), int(_cfg_get(mv, "test"))
    cfg.update(data_cfg or {})
    dev = next(model.parameters()).device
    with torch.device(dev):
        twin = gvr.GLENetVR(cfg, cfg["num_features"], roi_cfg=roi_cfg, head_cfg=head_cfg, bev_channels_last=True)
    if not dry_run:
        with torch.no_grad():
            for name in ("backbone_2d", "dense_head"):
                for p in getattr(model, name).parameters():
                    if p.dim() == 4:
                        p.data = p.data.contiguous(memory_format=torch.channels_last)
    shared = _share_state(twin, model)
    twin.train(model.training)
    if dry_run:
        return dict(shared=shared, roi_cfg=roi_cfg, head_cfg=head_cfg, cfg=cfg, twin=twin)
    if dev.type != "cuda":
        raise RuntimeError("dropin.record: the network must live on the GPU (no CPU path)")
    if isinstance(batches, dict):
        batches = [batches]
    batches = list(batches or [])
    if not batches:
        raise ValueError("dropin.record: at least one representative batch_dict (the warm-up passes of the recording run on it)")
    if batch_size is None:
        batch_size = int(batches[0]["gt_boxes"].shape[0])
    if max_points is None:
        max_points = (int(max(len(b["points"]) for b in batches) * 1.1) + 1023) // 1024 * 1024
    max_gt = max(int(max_gt), max(int(b["gt_boxes"].shape[1]) for b in batches))
    pipe = gvr.StaticTrainStep(twin, batch_size, max_points, cfg["num_features"], max_gt=max_gt, optimizer="external",
                               seed_rois_with_gt=seed_rois_with_gt, capacities=capacities, device=dev)
    rec = RecordedStep(model, twin, pipe)
    if capacities is None:
        caps = {}
        for bd in batches:
            pts, bidx, _, _ = rec.tensors(bd)
            for k, v in pipe.calibrate(pts, bidx).items():
                caps[k] = max(caps.get(k, 0), v)
        pipe.capacities = caps
    rec.load(batches[0])
    pipe.capture(warmup=warmup, split=True)
    return rec


# ------------------------------------------------------------------------------------------------ the recorded inference pass
def _translate_post_cfg(model_cfg):
    """det.post_processing's settings from MODEL.POST_PROCESSING (GLENet_VR.yaml:168-181)."""
    g = _cfg_get
    pp = g(model_cfg, "POST_PROCESSING")
    nms = g(pp, "NMS_CONFIG")
    if g(nms, "NMS_TYPE") != "new_nms_gpu" or g(nms, "MULTI_CLASSES_NMS") or g(pp, "OUTPUT_RAW_SCORE"):
        raise NotImplementedError("dropin.record_inference: class-agnostic variance-voting NMS (NMS_TYPE new_nms_gpu) of "
                                  "normalised scores (GLENet_VR.yaml:168-181)")
    cfg = dict(SCORE_THRESH=g(pp, "SCORE_THRESH"), POST_SCORE_THRESH=g(pp, "POST_SCORE_THRESH", None),
               NMS_THRESH=float(g(nms, "NMS_THRESH")), NMS_PRE_MAXSIZE=int(g(nms, "NMS_PRE_MAXSIZE")),
               NMS_POST_MAXSIZE=int(g(nms, "NMS_POST_MAXSIZE")))
    return cfg, [float(t) for t in (g(pp, "RECALL_THRESH_LIST") or [])]


def recall_counts(box_preds, rois, gt_boxes, thresh_list):
    """Detector3DTemplate.generate_recall_record's counts (detector3d_template.py:318-362) for a whole batch ON THE DEVICE:
    box_preds (B, R, 7+) the refined boxes, rois (B, R, 7+) or None, gt_boxes (B, G, 7+) with zero rows behind a frame's last box.
    Returns an int64 tensor [gt, rcnn_t0, roi_t0, rcnn_t1, roi_t1, ...]: the rows up to a frame's last non-zero row count as ground
    truth (as the reference's trimming loop leaves them); an all-zero row overlaps nothing, so it is never recalled."""
    import torch
    from .pcdet_ops.iou3d_nms import iou3d_nms_utils
    B, G = gt_boxes.shape[0], gt_boxes.shape[1]
    nz = gt_boxes.abs().sum(dim=-1) != 0                                          # (B, G)
    idx = torch.arange(1, G + 1, device=gt_boxes.device)
    n_gt = (nz * idx).max(dim=1)[0]                                               # last non-zero row + 1 per frame
    inside = idx[None, :] <= n_gt[:, None]
    out = [n_gt.sum()]
    rc, ro = [], []
    for b in range(B):
        g = gt_boxes[b, :, 0:7].contiguous()
        rc.append(iou3d_nms_utils.boxes_iou3d_gpu(box_preds[b, :, 0:7].contiguous(), g).max(dim=0)[0])
        if rois is not None:
            ro.append(iou3d_nms_utils.boxes_iou3d_gpu(rois[b, :, 0:7].contiguous(), g).max(dim=0)[0])
    rc = torch.stack(rc)
    ro = torch.stack(ro) if ro else None
    for t in thresh_list:
        out.append(((rc > t) & inside).sum())
        out.append(((ro > t) & inside).sum() if ro is not None else torch.zeros((), dtype=torch.int64, device=gt_boxes.device))
    return torch.stack([o.to(torch.int64) for o in out])


def recall_record(box_preds, rois, gt_boxes, thresh_list, recall_dict=None):
    """The dict Detector3DTemplate.post_processing returns beside pred_dicts, for a batch (one read-back)."""
    counts = recall_counts(box_preds, rois, gt_boxes, thresh_list).tolist()
    if not recall_dict:
        recall_dict = {"gt": 0}
        for t in thresh_list:
            recall_dict["roi_%s" % str(t)] = 0
            recall_dict["rcnn_%s" % str(t)] = 0
    recall_dict["gt"] += counts[0]
    for i, t in enumerate(thresh_list):
        recall_dict["rcnn_%s" % str(t)] += counts[1 + 2 * i]
        recall_dict["roi_%s" % str(t)] += counts[2 + 2 * i]
    return recall_dict


class RecordedInference:
    """What dropin.record_inference() returns: a callable with the contract of the reference's network in EVAL mode
    (pcdet/models/detectors/voxel_rcnn.py:forward: `pred_dicts, recall_dicts = model(batch_dict)`, consumed by
    tools/eval_utils/eval_utils.py:53-66), backed by one recorded HIP graph of the whole pass -- voxelization, sparse backbone,
    BEV backbone + anchor head with eval-mode BatchNorm folded, proposals, RoI-grid pooling, FC towers, box refinement and
    Detector3DTemplate.post_processing's variance-voting NMS on the device -- over the network's own parameters."""

    def __init__(self, model, twin, pipe, thresh_list):
        self.model, self.twin, self.pipe, self.thresh_list = model, twin, pipe, thresh_list

    def __call__(self, batch_dict):
        import torch
        from . import detector as det
        if self.model.training:
            raise RuntimeError("dropin.record_inference: the recorded pass is the EVAL pass (model.eval())")
        dev = self.pipe.points.device
        pts = torch.as_tensor(batch_dict["points"]).to(dev, non_blocking=True)
        self.pipe.load(pts[:, 1:].float().contiguous(), pts[:, 0].to(torch.int32).contiguous())
        out = self.pipe.replay()
        pred = det.pred_dicts(out["post"])                       # the pass's one read-back
        recall = {}
        if "gt_boxes" in batch_dict and self.thresh_list:
            gt = torch.as_tensor(batch_dict["gt_boxes"]).to(dev, torch.float32)
            recall = recall_record(out["batch_box_preds"], out["rois"], gt, self.thresh_list)
        return pred, recall

    def check(self):
        self.pipe.check()
        return True


def record_inference(model, batches, batch_size=None, max_points=None, data_cfg=None):
    """The eval-mode counterpart of record(): the inference pass of a reference-built GLENet-VR network as one recorded graph on
    the network's own parameters (a parameter-sharing twin, the configuration -- NMS_CONFIG.TEST, POST_PROCESSING -- translated
    from model.model_cfg).  batches: representative collated batch_dicts (`points` (P, 1 + C)); returns a RecordedInference.
    A change of the weights (load_state_dict, a training step) is noticed by the pipeline's weight tag: the pass re-records."""
    import torch
    from . import detector as det
    from . import glenet_vr as gvr
    roi_cfg, head_cfg = _translate_cfg(model.model_cfg)
    post_cfg, thresh = _translate_post_cfg(model.model_cfg)
    ds = getattr(model, "dataset", None)
    cfg = dict(point_cloud_range=[round(float(v), 5) for v in ds.point_cloud_range],
               voxel_size=[round(float(v), 6) for v in ds.voxel_size], max_points=5, max_voxels_train=16000, max_voxels_test=40000,
               num_features=int(getattr(getattr(ds, "point_feature_encoder", None), "num_point_features", 4)))
    dc = getattr(ds, "dataset_cfg", None)
    for p in (_cfg_get(dc, "DATA_PROCESSOR", None) or []):
        if _cfg_get(p, "NAME") == "transform_points_to_voxels":
            cfg["max_points"] = int(_cfg_get(p, "MAX_POINTS_PER_VOXEL"))
            mv = _cfg_get(p, "MAX_NUMBER_OF_VOXELS")
            cfg["max_voxels_train"], cfg["max_voxels_test"] = int(_cfg_get(mv, "train")), int(_cfg_get(mv, "test"))
    cfg.update(data_cfg or {})
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("dropin.record_inference: the network must live on the GPU (no CPU path)")
    with torch.device(dev):
        twin = gvr.GLENetVR(cfg, cfg["num_features"], roi_cfg=roi_cfg, head_cfg=head_cfg, bev_channels_last=True)
    with torch.no_grad():
        for name in ("backbone_2d", "dense_head"):
            for p in getattr(model, name).parameters():
                if p.dim() == 4:
                    p.data = p.data.contiguous(memory_format=torch.channels_last)
    _share_state(twin, model)
    twin.eval()
    if isinstance(batches, dict):
        batches = [batches]
    batches = list(batches or [])
    if not batches:
        raise ValueError("dropin.record_inference: at least one representative batch_dict")
    if batch_size is None:
        batch_size = int(batches[0].get("batch_size") or int(torch.as_tensor(batches[0]["points"])[:, 0].max().item()) + 1)
    if max_points is None:
        max_points = (int(max(len(b["points"]) for b in batches) * 1.1) + 1023) // 1024 * 1024
    pipe = det.StaticDetectorPipeline(twin, batch_size, max_points, cfg["num_features"], device=dev)
    pipe.post_cfg = post_cfg
    caps = {}
    first = None
    for bd in batches:
        pts = torch.as_tensor(bd["points"]).to(dev)
        xyz, bidx = pts[:, 1:].float().contiguous(), pts[:, 0].to(torch.int32).contiguous()
        first = first if first is not None else (xyz, bidx)
        for k, v in pipe.calibrate(xyz, bidx).items():
            caps[k] = max(caps.get(k, 0), v)
    pipe.capacities = caps
    pipe.load(*first)
    pipe.capture()
    return RecordedInference(model, twin, pipe, thresh)
