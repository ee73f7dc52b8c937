"""glenet_amd: MI355X-native implementation of GLENet's data-parallel detection hot path.

Sub-packages mirror the reference interfaces they replace:
  glenet_amd.spconv      -> the `spconv` surface (SparseConvTensor, SubMConv3d, ...)
  glenet_amd.pcdet_ops   -> `pcdet.ops.*` (iou3d_nms, pointnet2_stack, roiaware/roipoint pools)
  glenet_amd.voxelize    -> device voxelizers (VoxelGeneratorWrapper / DynamicMeanVFE semantics)
  glenet_amd.dropin      -> registers the above under the reference's import names
All compute goes through csrc/libglenet_hip.so (C ABI in include/glenet_hip.h).
"""
import os as _os

__version__ = "0.1.0"

# ROCm 7.2's HIP-graph executor replays the branches of a recorded graph on DEBUG_HIP_FORCE_GRAPH_QUEUES streams (default
# 4).  The recorded steps here have three branches (main, RoI, rule tables / weight gradients), of which two are long: with
# TWO executor queues the GLENet-VR training step replays in 6.50 ms instead of 6.67 (profiles/r04_summary.md, six alternating
# runs on one box; 1 queue = no overlap: 8.17 ms; 3: 6.63; 6 / 8: 6.67).  The HIP runtime reads the variable when it
# initialises (the first HIP call of the process), so it is set here, when the package is imported, unless the caller's
# environment already says something else.
_os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", "2")
