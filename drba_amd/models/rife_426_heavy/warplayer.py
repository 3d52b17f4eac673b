"""Backward warp, reference signature warp(tenInput, tenFlow) (models/rife_426_heavy/warplayer.py:8).

grid_sample(bilinear, padding_mode='border', align_corners=True) on base grid + normalised
flow, evaluated by the HIP gather kernel (drba_amd/csrc/splat_warp.hip); no grid tensor is
built or cached.
"""
from drba_amd import ops as _ops


def warp(tenInput, tenFlow):
    return _ops.backwarp(tenInput, tenFlow, "border")
