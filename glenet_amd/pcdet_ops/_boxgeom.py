"""Shared box arithmetic of the pcdet.ops wrappers (torch, any device)."""
import torch


def z_overlap(za, ha, zb, hb, pairwise):
    """Overlap of the vertical extents [z - h/2, z + h/2]; (N,M) if pairwise else (N,1)."""
    lo_a, hi_a = (za - ha / 2).view(-1, 1), (za + ha / 2).view(-1, 1)
    shape = (1, -1) if pairwise else (-1, 1)
    lo_b, hi_b = (zb - hb / 2).view(*shape), (zb + hb / 2).view(*shape)
    return (torch.min(hi_a, hi_b) - torch.max(lo_a, lo_b)).clamp(min=0)


def iou3d_from_bev(bev_overlap, a, b, pairwise, eps, h_col=5):
    """3-D IoU from a BEV overlap area: intersect heights, divide by the clamped union volume."""
    inter = bev_overlap * z_overlap(a[:, 2], a[:, h_col], b[:, 2], b[:, h_col], pairwise)
    va = (a[:, 3] * a[:, 4] * a[:, 5]).view(-1, 1)
    vb = (b[:, 3] * b[:, 4] * b[:, 5]).view(*((1, -1) if pairwise else (-1, 1)))
    return inter / (va + vb - inter).clamp(min=eps)
