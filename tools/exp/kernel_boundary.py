#!/usr/bin/env python3
"""Cost of a dependent kernel boundary on one stream: back-to-back launches of a trivial library kernel (drba_clamp on 64
floats) and of a small / a large convolution, total time / launches, untraced.  python tools/exp/kernel_boundary.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.rand(1, 1, 8, 8, device=dev)


def run(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, t_host / n * 1e6


print("clamp 64 floats: %.2f us per launch (host enqueue %.2f)" % run(lambda: ops.clamp(x, 0.0, 1.0), 2000))
g = torch.Generator().manual_seed(0)
for (c, h, w) in ((32, 8, 32), (32, 272, 480), (192, 17, 30)):
    wt = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    layers = [(ops.Conv3x3(wt, torch.zeros(c), 1, True, torch.ones(1, c, 1, 1), device=dev), True) for _ in range(8)]
    chain = ops.ConvChain(layers)
    xx = torch.randn(2, c, h, w, generator=g).to(dev)
    chain(xx)
    chain(xx)
    us, host = run(lambda: chain(xx), 200)
    print(f"chain of 8 ResConv {c}ch {h}x{w} N2: {us / 8:.2f} us per layer (host {host / 8:.2f})")
