#!/usr/bin/env python3
"""Three launches of drba_stage_conv0_batch at 1080p, 8 samples (for rocprofv3 --pmc passes).  python tools/exp/stage_conv_pmc.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
H, W, B = 1088, 1920, 8
conv = ops.Conv3x3(torch.randn(16, 52, 3, 3, generator=g) * 0.05, torch.zeros(16), 2, True, None, device=dev)
items, flows = [], []
for _ in range(B):
    i0, i1 = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 3, H, W, generator=g).to(dev)
    f0, f1 = torch.randn(1, 16, H, W, generator=g).to(dev), torch.randn(1, 16, H, W, generator=g).to(dev)
    items.append((i0, i1, torch.rand(1, 1, H, W, generator=g).to(dev), f0, f1))
    lo = torch.randn(1, 4, H // 16, W // 16, generator=g) * 5
    flows.append(torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear").to(dev).contiguous())
tprev = torch.randn(B, 13, H // 2, W // 2, generator=g).to(dev)
for _ in range(3):
    ops.stage_conv0(items, flows, tprev, 2.0, conv, fold=True)
torch.cuda.synchronize()
