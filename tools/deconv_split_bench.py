#!/usr/bin/env python3
"""fp32 MFMA deconv configs against the split-bf16 ones on the transposed convolutions of the 1080p paths.
    python tools/deconv_split_bench.py"""
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


layers = [("b4.last N2", 2, 32, 52, 272, 480, True), ("b3.last N2", 2, 64, 52, 136, 240, True), ("b2.last N2", 2, 96, 52, 68, 120, True),
          ("grid up 96->64", 1, 96, 64, 288, 480, False), ("grid up 64->32", 1, 64, 32, 576, 960, False)]
g = torch.Generator().manual_seed(0)
for name, n, cin, cout, h, w, ps in layers:
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    flop = 2.0 * n * cout * cin * 16 * h * w
    res, outs = [], {}
    for cfg in range(lib.drba_deconv4x4_num_cfgs()):
        if lib.drba_deconv4x4_packed_floats(cin, cout, cfg) == 0:
            continue
        layer = ops.Deconv4x4(wt, b, ps, device=dev, cfg=cfg)
        res.append((timeit(lambda: layer(x)), cfg))
        outs[cfg] = layer(x)
    best32, bests = min(r for r in res if r[1] < 6), min(r for r in res if r[1] >= 6)
    d = float((outs[bests[1]] - outs[best32[1]]).abs().max())
    print(f"{name:16s} fp32 cfg{best32[1]} {best32[0]:7.1f} us {flop / best32[0] / 1e6:6.1f} TF/s | split cfg{bests[1]} {bests[0]:7.1f} us "
          f"{flop / bests[0] / 1e6:6.1f} TF/s | " + " ".join(f"{c}:{u:.0f}" for u, c in res) + f" | max diff {d:.2e}", flush=True)
