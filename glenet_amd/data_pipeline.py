"""GPU-resident data step in front of the path: the three point
operations of DataProcessor (pcdet/datasets/processor/data_processor.py:78-152) on stacked device
tensors, so the DataLoader workers only read files and the reference's own
`transform_points_to_voxels_placeholder` hook (:107-115) can leave voxelization to the device.

  mask_points_by_range     common_utils.py:60-63   (x/y only, closed interval, as the reference)
  shuffle_points           data_processor.py:95-105 (per frame, torch generator instead of np.random)
  DeviceDataProcessor      mask -> shuffle -> hard voxelize (+ the voxelizer's cell index)
"""
import torch

from . import backbone as gb


def mask_points_by_range(points, limit_range):
    return ((points[:, 0] >= limit_range[0]) & (points[:, 0] <= limit_range[3]) &
            (points[:, 1] >= limit_range[1]) & (points[:, 1] <= limit_range[4]))


def shuffle_points(points, batch_idx, batch_size, generator=None):
    """Random permutation of the points of every frame; frames stay stacked in order (the hard
    voxelizer needs non-decreasing frame ids).  One sort of (frame id + uniform noise) keys."""
    noise = torch.rand(points.shape[0], device=points.device, generator=generator)
    order = torch.argsort(batch_idx.to(torch.float32) + noise * 0.5)
    return points[order], batch_idx[order]


class DeviceDataProcessor:
    def __init__(self, cfg, training=True, shuffle=True, seed=None):
        self.cfg, self.training, self.shuffle = cfg, training, shuffle
        self.generator = None
        self.seed = seed

    def __call__(self, points, batch_idx, batch_size, static=False):
        if self.seed is not None and self.generator is None:
            self.generator = torch.Generator(device=points.device).manual_seed(self.seed)
        keep = mask_points_by_range(points, self.cfg["point_cloud_range"])
        points, batch_idx = points[keep], batch_idx[keep]
        if self.shuffle:
            points, batch_idx = shuffle_points(points, batch_idx, batch_size, self.generator)
        bd = gb.voxelize_batch(points.contiguous(), batch_idx.contiguous(), batch_size, self.cfg,
                               train=self.training, static=static)
        bd["points"], bd["point_batch_idx"] = points, batch_idx
        return bd
