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
    """Put the reference's BEV backbones on the own dense kernels: every `BaseBEVBackbone` in `model`
    (pcdet/models/backbones_2d/base_bev_backbone.py:6-112 -- `blocks`, `deblocks`, `forward(data_dict)`) is re-classed to
    glenet_amd.dense_path.BEVBackbone, which keeps the module lists, parameter names and data_dict keys and runs the 3x3
    / strided / transposed convolutions on csrc/glx_conv2d.hip + glx_deconv2d.hip (fp32 products on the bf16 matrix
    pipe), training-mode BatchNorm on csrc/glx_bn.hip with the statistics in the conv epilogues, eval-mode BatchNorm
    folded into the epilogues.  channels_last: filters are moved to channels-last memory and the incoming
    `spatial_features` map is converted on entry (one copy); everything downstream sees logical NCHW tensors as before.
    Returns the names of the modules it changed.  Nothing else of the reference is touched; on CPU tensors the module
    runs its layers one by one exactly as the reference does."""
    import torch
    from . import dense_path as dp
    changed = []
    for name, m in model.named_modules():
        if type(m).__name__ == "BaseBEVBackbone" and not isinstance(m, dp.BEVBackbone) \
                and isinstance(getattr(m, "blocks", None), torch.nn.ModuleList) \
                and isinstance(getattr(m, "deblocks", None), torch.nn.ModuleList):
            m.__class__ = dp.BEVBackbone
            if channels_last:
                m.to(memory_format=torch.channels_last)
                m.convert_input = True
            changed.append(name)
    return changed
