"""glenet_amd: MI355X-native implementation of GLENet's data-parallel detection hot path.

Sub-packages mirror the reference interfaces they replace:
  glenet_amd.spconv      -> the `spconv` surface (SparseConvTensor, SubMConv3d, ...)
  glenet_amd.pcdet_ops   -> `pcdet.ops.*` (iou3d_nms, pointnet2_stack, roiaware/roipoint pools)
  glenet_amd.voxelize    -> device voxelizers (VoxelGeneratorWrapper / DynamicMeanVFE semantics)
  glenet_amd.dropin      -> registers the above under the reference's import names
All compute goes through csrc/libglenet_hip.so (C ABI in include/glenet_hip.h).
"""
__version__ = "0.1.0"
