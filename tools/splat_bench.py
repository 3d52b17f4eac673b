#!/usr/bin/env python3
"""Time softsplat('soft') at the GMFSS pyramid shapes for a 1080p input (net 1152x1920 -> 576x960 and below).
    python tools/splat_bench.py [flow_px]      flow_px = amplitude of the smooth synthetic flow (default 6)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import ops  # noqa: E402

amp = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (C, H, W) in ((1, 576, 960), (3, 576, 960), (64, 576, 960), (128, 288, 480), (192, 144, 240), (2, 1088, 1920)):
    x = torch.randn(1, C, H, W, generator=g).to(dev)
    flow = torch.nn.functional.interpolate(torch.randn(1, 2, H // 32, W // 32, generator=g) * amp, size=(H, W), mode="bilinear").to(dev).contiguous()
    m = torch.randn(1, 1, H, W, generator=g).to(dev)
    for mode, mm in (("soft", m), ("avg", None)):
        us = timeit(lambda: ops.softsplat(x, flow, mm, mode))
        nbytes = 4.0 * H * W * (2 * C + 3)  # read in + flow + metric, write out
        print(f"softsplat {mode:4s} C={C:3d} {H}x{W} |flow|~{amp:g}px: {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s algorithmic")
