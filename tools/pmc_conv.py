#!/usr/bin/env python3
"""Launch the RIFE ResConv shapes (N=2 stacked batch) with chosen configurations a few times, for rocprofv3 --pmc passes.
    python tools/pmc_conv.py [cfg ...]        default: 2 6 0 1 3"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
ops.AUTOTUNE = False
cfgs = [int(a) for a in sys.argv[1:]] or [2, 6, 0, 1, 3]
for (c, h, w, n) in ((64, 136, 240, 2), (32, 272, 480, 2), (64, 576, 960, 1)):
    x = torch.randn(n, c, h, w, generator=g).to(dev)
    wt = torch.randn(c, c, 3, 3, generator=g) * 0.05
    for cfg in cfgs:
        layer = ops.Conv3x3(wt, torch.zeros(c), 1, True, torch.ones(1, c, 1, 1), device=dev, cfg=cfg)
        out = torch.empty_like(x)
        for _ in range(3):
            layer(x, residual=x, out=out)
torch.cuda.synchronize()
