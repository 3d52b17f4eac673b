#!/usr/bin/env python3
"""fp32 MFMA configs against the split-bf16 family on stride-1 3x3 layers (GridNet / FeatureNet / IFBlock shapes).
    python tools/conv_split_bench.py
Prints per layer the time and effective fp32 TFLOP/s of every config that can run it, and max|split - fp32|."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()


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


layers = [("grid 32->32 full", 1, 32, 32, 1152, 1920), ("grid 64->64 half", 1, 64, 64, 576, 960), ("grid 96->96 quarter", 1, 96, 96, 288, 480),
          ("feat 32->32 full", 1, 32, 32, 1152, 1920), ("b4.res 1080p N2", 2, 32, 32, 272, 480), ("b3.res 1080p N2", 2, 64, 64, 136, 240),
          ("b2.res 1080p N2", 2, 96, 96, 68, 120), ("b1.res 1080p N2", 2, 128, 128, 34, 60), ("b2.res 1080p N1", 1, 96, 96, 68, 120),
          ("b4.res 4K N2", 2, 32, 32, 544, 960), ("b3.res 4K N2", 2, 64, 64, 272, 480), ("b2.res 4K N2", 2, 96, 96, 136, 240)]
g = torch.Generator().manual_seed(0)
for name, n, cin, cout, h, w in layers:
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    flop = 2.0 * n * cout * cin * 9 * h * w
    res, outs = [], {}
    for cfg in range(lib.drba_conv3x3_num_cfgs()):
        if lib.drba_conv3x3_cfg_stride(cfg) != 1 or lib.drba_conv3x3_packed_floats(cin, cout, cfg) == 0:
            continue
        layer = ops.Conv3x3(wt, b, 1, True, None, device=dev, cfg=cfg)
        us = timeit(lambda: layer(x))
        res.append((us, cfg))
        outs[cfg] = layer(x)
    best32 = min(r for r in res if r[1] < 14)
    bests = min(r for r in res if r[1] >= 14)
    d = float((outs[bests[1]] - outs[best32[1]]).abs().max())
    print(" ".join(f"{c}:{u:.0f}" for u, c in res if c < 14), end=" || ")
    print(f"{name:22s} fp32 cfg{best32[1]:2d} {best32[0]:7.1f} us {flop / best32[0] / 1e6:6.1f} TF/s | split cfg{bests[1]:2d} {bests[0]:7.1f} us "
          f"{flop / bests[0] / 1e6:6.1f} TF/s | " + " ".join(f"{c}:{u:.0f}" for u, c in res if c >= 14) + f" | max diff {d:.2e}", flush=True)
