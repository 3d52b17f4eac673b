#!/usr/bin/env python3
"""Launch the roofline-target kernels a few times each (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes).
A plain elementwise kernel with a known byte count (drba_affine on 64 Mi floats: 256 MiB read + 256 MiB written,
4 B per lane like the targets) calibrates the counters, as MI355X_MICROARCH.md 'HBM' prescribes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
H, W = 1088, 1920
ops.AUTOTUNE = False
a = torch.randn(64 << 20, generator=g).to(dev)
for _ in range(3):
    ops.affine(a, 1.5, 0.25)
img0, img1 = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 3, H, W, generator=g).to(dev)
f0, f1 = torch.randn(1, 16, H, W, generator=g).to(dev), torch.randn(1, 16, H, W, generator=g).to(dev)
flow = torch.nn.functional.interpolate(torch.randn(1, 4, H // 32, W // 32, generator=g) * 6, size=(H, W), mode="bilinear").to(dev).contiguous()
tmap = torch.rand(1, 1, H, W, generator=g).to(dev)
for s in (1.0, 2.0):
    tprev = torch.randn(1, 13, int(H / (2 * s)), int(W / (2 * s)), generator=g).to(dev)
    for _ in range(3):
        ops.ifblock_input(img0, img1, f0, f1, tmap, flow, tprev, 2 * s, s)
for (c, h, w, cfg) in ((32, 272, 480, 2), (64, 136, 240, 2)):
    x = torch.randn(1, c, h, w, generator=g).to(dev)
    layer = ops.Conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.05, torch.zeros(c), 1, True, torch.ones(1, c, 1, 1), device=dev, cfg=cfg)
    out = torch.empty_like(x)
    for _ in range(3):
        layer(x, residual=x, out=out)
torch.cuda.synchronize()
