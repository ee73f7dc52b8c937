"""ctypes binding of libglenet_host.so (include/glenet_host.h): the entry points whose contract is host memory +
host arithmetic + callable from forked DataLoader workers.  This module imports neither torch nor
the HIP library; numpy arrays in, numpy arrays out."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libglenet_host.so")
_lib = None


class GlxHostError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GlxHostError("libglenet_host.so not found at %s -- run `python -c 'import __graft_entry__ as g; "
                               "g.build()'`" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def _f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None and (a.ndim != 2 or a.shape[1] != cols):
        raise GlxHostError("expected an (N, %d) float32 array, got %s" % (cols, a.shape))
    return a


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _check(rc, name):
    if rc != 0:
        raise GlxHostError("%s failed (%d)" % (name, rc))


def boxes_iou_bev(boxes_a, boxes_b, out=None):
    a, b = _f32(boxes_a, 7), _f32(boxes_b, 7)
    if out is None:
        out = np.empty((len(a), len(b)), np.float32)
    _check(load().glxh_boxes_iou_bev(_p(a), len(a), _p(b), len(b), _p(out)), "glxh_boxes_iou_bev")
    return out


def iou3d_boxes_bev(boxes_a, boxes_b, iou=True):
    a, b = _f32(boxes_a, 5), _f32(boxes_b, 5)
    out = np.empty((len(a), len(b)), np.float32)
    fn = load().glxh_iou3d_boxes_iou_bev if iou else load().glxh_iou3d_boxes_overlap_bev
    _check(fn(_p(a), len(a), _p(b), len(b), _p(out)), "glxh_iou3d_boxes_*_bev")
    return out


def points_in_boxes(boxes, points, out=None):
    b, p = _f32(boxes, 7), _f32(points, 3)
    if out is None:
        out = np.empty((len(b), len(p)), np.int32)
    _check(load().glxh_points_in_boxes(_p(b), len(b), _p(p), len(p), _p(out)), "glxh_points_in_boxes")
    return out


def voxelize_hard(points, voxel_size, point_cloud_range, max_points, max_voxels):
    """-> voxels (Nv, max_points, C), coords (Nv, 3) [z,y,x], num_points (Nv,)."""
    pts = _f32(points)
    if pts.ndim != 2 or pts.shape[1] < 3:
        raise GlxHostError("expected (P, C >= 3) points, got %s" % (pts.shape,))
    rng = np.asarray(point_cloud_range, np.float32)
    vs = np.asarray(voxel_size, np.float32)
    grid = np.round((np.asarray(point_cloud_range, np.float64)[3:6] - np.asarray(point_cloud_range, np.float64)[0:3])
                    / np.asarray(voxel_size, np.float64)).astype(np.int32)      # data_processor.py:119-120
    P, C = pts.shape
    voxels = np.empty((max_voxels, max_points, C), np.float32)
    coords = np.empty((max_voxels, 3), np.int32)
    num = np.empty((max_voxels,), np.int32)
    nv = ctypes.c_int(0)
    _check(load().glxh_voxelize_hard(_p(pts), P, C, _p(rng), _p(vs), _p(grid), int(max_points), int(max_voxels),
                                     _p(voxels), _p(coords), _p(num), ctypes.byref(nv)), "glxh_voxelize_hard")
    n = nv.value
    return voxels[:n], coords[:n], num[:n]
