"""glenet_amd: MI355X-native implementation of GLENet's data-parallel detection hot path.

Sub-packages mirror the reference interfaces they replace:
  glenet_amd.spconv      -> the `spconv` surface (SparseConvTensor, SubMConv3d, ...)
  glenet_amd.pcdet_ops   -> `pcdet.ops.*` (iou3d_nms, pointnet2_stack, roiaware/roipoint pools)
  glenet_amd.voxelize    -> device voxelizers (VoxelGeneratorWrapper / DynamicMeanVFE semantics)
  glenet_amd.dropin      -> registers the above under the reference's import names
  glenet_amd.runtime     -> opt-in process-level settings (HIP-graph executor queues)
All compute goes through csrc/libglenet_hip.so (C ABI in include/glenet_hip.h).
"""
__version__ = "0.1.0"

# Importing the package changes nothing in the process: runtime switches (the HIP-graph executor's queue count) are
# explicit calls of the entry points -- glenet_amd.runtime.configure_graph_executor(), which bench.py calls and reports.
