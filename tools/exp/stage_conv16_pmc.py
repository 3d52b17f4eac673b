#!/usr/bin/env python3
"""Three launches each of drba_stage_conv16_batch and drba_stage_conv0_batch at 1080p, 8 samples, the flow as three terms (the
loop's launch), for rocprofv3 --pmc passes.    python tools/exp/stage_conv16_pmc.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
H, W, B = 1088, 1920, 8
conv = ops.Conv3x3(torch.randn(16, 52, 3, 3, generator=g) * 0.05, torch.zeros(16), 2, True, None, device=dev)
# eight items over six frames, each frame with its [H,W,4] copy: the launch the pipeline's groups make
fr = []
for _ in range(B // 2 + 2):
    im, ft = torch.rand(1, 3, H, W, generator=g).to(dev), torch.randn(1, 16, H, W, generator=g).to(dev)
    ops.rgbx(im)
    fr.append((im, ft))
items = []
for j in range(B // 2):
    (a, fa), (b, fb), (c, fc) = fr[j], fr[j + 1], fr[j + 2]
    items += [(b, a, torch.rand(1, 1, H, W, generator=g).to(dev), fb, fa), (b, c, torch.rand(1, 1, H, W, generator=g).to(dev), fb, fc)]


def head(st, amp):
    t = torch.randn(B, 13, H // st, W // st, generator=g)
    lo = torch.randn(B, 4, max(H // st // 8, 2), max(W // st // 8, 2), generator=g) * amp
    t[:, :4] = torch.nn.functional.interpolate(lo, size=t.shape[2:], mode="bicubic", align_corners=False)
    return t.to(dev)


terms = [(head(16, 1.0), 16.0), (head(8, 0.4), 8.0), (head(4, 0.3), 4.0)]
tprev = head(2, 0.3)
for two in (True, False):
    ops.STAGE_CONV_TWO_TERM = two
    for _ in range(3):
        ops.stage_conv0(items, None, tprev, 2.0, conv, terms=terms)
    torch.cuda.synchronize()
