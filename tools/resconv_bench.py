#!/usr/bin/env python3
"""ResConv layers (lrelu(conv3x3(x) * beta + x), IFNet_HDv3.py:50-59) of the 1080p / 4K RIFE path and the GridNet-sized
stride-1 layers: every split-bf16 configuration and the best fp32 one, back-to-back launches timed with events.
    python tools/resconv_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


layers = [("b4.res 1080p N2", 2, 32, 272, 480), ("b3.res 1080p N2", 2, 64, 136, 240), ("b2.res 1080p N2", 2, 96, 68, 120),
          ("b1.res 1080p N2", 2, 128, 34, 60), ("b0.res 1080p N2", 2, 192, 17, 30), ("b4.res 4K N2", 2, 32, 544, 960),
          ("b3.res 4K N2", 2, 64, 272, 480), ("grid 32 full", 1, 32, 1152, 1920), ("grid 64 half", 1, 64, 576, 960),
          ("grid 96 quarter", 1, 96, 288, 480)]
g = torch.Generator().manual_seed(0)
for name, n, c, h, w in layers:
    x = torch.randn(n, c, h, w, generator=g).to(dev)
    wt = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    b = torch.randn(c, generator=g) * 0.1
    beta = torch.rand(1, c, 1, 1, generator=g) + 0.5
    flop = 2.0 * n * c * c * 9 * h * w
    out = torch.empty_like(x)
    res = []
    for cfg in range(lib.drba_conv3x3_num_cfgs()):
        if lib.drba_conv3x3_cfg_stride(cfg) != 1 or lib.drba_conv3x3_packed_floats(c, c, cfg) == 0:
            continue
        layer = ops.Conv3x3(wt, b, 1, True, beta, device=dev, cfg=cfg)
        res.append((timeit(lambda: layer(x, residual=x, out=out)), cfg))
    best32 = min(r for r in res if r[1] < 14)
    line = f"{name:18s} {flop / 1e9:6.2f} GF | fp32 cfg{best32[1]:2d} {best32[0]:6.1f} us {flop / best32[0] / 1e6:6.1f} TF/s | split: "
    line += "  ".join(f"cfg{cfg} {us:6.1f} us {flop / us / 1e6:6.1f} TF/s" for us, cfg in res if cfg >= 14)
    print(line, flush=True)
