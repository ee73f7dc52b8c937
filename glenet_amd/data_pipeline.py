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


def mask_and_shuffle_static(points, batch_idx, batch_size, limit_range, state, shuffle=True):
    """The same two operations on a CAPACITY-sized stacked buffer without changing its shape or reading anything
    back (capturable in a HIP graph; csrc/glx_voxelize.hip: glx_mask_shuffle): rows outside the x / y range (and
    padding rows, frame id == batch_size) are dropped, the kept rows of every frame are written in a keyed
    pseudo-random order, the rest of the buffer becomes padding.  state: device int64[2] = (seed, calls so far).
    -> (points, batch_idx, order): order[i] = source row of output row i (int32, -1 for padding)."""
    import ctypes
    from . import _lib
    _lib.check_cuda(points, batch_idx, state)
    p, c = points.shape
    out_p, out_b = torch.empty_like(points), torch.empty_like(batch_idx)
    order = torch.empty(p, dtype=torch.int32, device=points.device)
    ws = _lib.workspace.get(_lib.query("glx_mask_shuffle_workspace_bytes", p, int(batch_size)), points.device)
    rng = (ctypes.c_float * 4)(float(limit_range[0]), float(limit_range[1]), float(limit_range[3]), float(limit_range[4]))
    _lib.call("glx_mask_shuffle", points, batch_idx, p, c, int(batch_size), rng, 1 if shuffle else 0, state, out_p, out_b,
              order, ws, _lib.size_arg(ws.numel()))
    return out_p, out_b, order


class DeviceDataProcessor:
    def __init__(self, cfg, training=True, shuffle=True, seed=None):
        self.cfg, self.training, self.shuffle = cfg, training, shuffle
        self.generator = None
        self.seed = seed
        self._state = None          # device (seed, call counter) of the static step's keyed permutation

    def static_step(self, points, batch_idx, batch_size):
        """mask + shuffle on capacity-sized buffers (see mask_and_shuffle_static); voxelization stays with the caller
        (StaticTrainPipeline runs it right after, inside the same recorded step)."""
        if self._state is None or self._state.device != points.device:
            self._state = torch.tensor([self.seed if self.seed is not None else 0x5EED, 0], dtype=torch.int64,
                                       device=points.device)
        return mask_and_shuffle_static(points, batch_idx, batch_size, self.cfg["point_cloud_range"], self._state,
                                       self.shuffle)[:2]

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
