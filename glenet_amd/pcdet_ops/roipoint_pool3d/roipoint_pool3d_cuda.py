"""Drop-in for `pcdet.ops.roipoint_pool3d.roipoint_pool3d_cuda` (roipoint_pool3d.cpp:57-59)."""
from ... import _lib
from ..._lib import call


def forward(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
    _lib.check_cuda(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag)
    b, n, _ = xyz.shape
    m, c, s = boxes3d.shape[1], pts_feature.shape[2], pooled_features.shape[2]
    call("glx_roipoint_pool3d", xyz, boxes3d, pts_feature, b, n, m, c, s, pooled_features,
         pooled_empty_flag)
    return 1
