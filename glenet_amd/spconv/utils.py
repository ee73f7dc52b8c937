"""`spconv.utils` voxel generators with the constructor/generate signatures the reference
calls (pcdet/datasets/processor/data_processor.py:15-60).  numpy in, numpy out, HOST arithmetic
(libglenet_host.so through glenet_amd._host): the reference builds these inside Dataset objects and
calls them from forked DataLoader workers, so they must not touch the GPU runtime.  (The device
voxelizer of the training pipeline is glenet_amd.voxelize.hard_voxelize / data_pipeline.)"""
import numpy as np

from .. import _host


class _TV:
    """Minimal stand-in for the cumm.tensorview arrays Point2VoxelCPU3d returns: `.numpy()`."""

    def __init__(self, a):
        self._a = a

    def numpy(self):
        return self._a

    def numpy_view(self):
        return self._a


class VoxelGeneratorV2:
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000,
                 full_mean=False, block_filtering=False, block_factor=8, block_size=3,
                 height_threshold=0.1, height_high_threshold=2.0):
        assert not full_mean and not block_filtering, "not used by the reference configs"
        self._voxel_size = [float(v) for v in voxel_size]
        self._point_cloud_range = [float(v) for v in point_cloud_range]
        self._max_num_points = int(max_num_points)
        self._max_voxels = int(max_voxels)

    def generate(self, points, max_voxels=None):
        v, c, n = _host.voxelize_hard(np.asarray(points), self._voxel_size, self._point_cloud_range,
                                      self._max_num_points, max_voxels or self._max_voxels)
        return {"voxels": v, "coordinates": c, "num_points_per_voxel": n}

    @property
    def voxel_size(self):
        return self._voxel_size

    @property
    def point_cloud_range(self):
        return self._point_cloud_range


VoxelGenerator = VoxelGeneratorV2


class Point2VoxelCPU3d:
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels,
                 max_num_points_per_voxel):
        self._gen = VoxelGeneratorV2(vsize_xyz, coors_range_xyz, max_num_points_per_voxel,
                                     max_num_voxels)
        self.num_point_features = num_point_features

    def point_to_voxel(self, pc):
        pts = pc.numpy() if hasattr(pc, "numpy") else np.asarray(pc)
        out = self._gen.generate(pts)
        return _TV(out["voxels"]), _TV(out["coordinates"]), _TV(out["num_points_per_voxel"])
