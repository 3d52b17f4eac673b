#!/usr/bin/env python3
"""Launch the RIFE lastconv / encoder deconvolutions a few times (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (cin, cout, h, w, n, ps) in ((32, 52, 272, 480, 2, True), (64, 52, 136, 240, 2, True), (16, 16, 544, 960, 1, False)):
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    layer = ops.Deconv4x4(torch.randn(cin, cout, 4, 4, generator=g) * 0.05, torch.zeros(cout), pixel_shuffle=ps, device=dev)
    for _ in range(4):
        y = layer(x)
torch.cuda.synchronize()
