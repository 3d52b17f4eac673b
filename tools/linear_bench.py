#!/usr/bin/env python3
"""drba_linear_split against torch.matmul (rocBLAS / hipBLASLt fp32) on the GMFlow transformer's GEMM shapes at 1080p.
    python tools/linear_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
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


for name, m, k, n, gelu in (("qkv fine", 69120, 128, 384, False), ("proj fine", 69120, 128, 128, False), ("kv fine", 69120, 128, 256, False),
                            ("mlp0 fine", 69120, 256, 1024, True), ("mlp2 fine", 69120, 1024, 128, False),
                            ("qkv coarse", 17280, 128, 384, False), ("mlp0 coarse", 17280, 256, 1024, True), ("mlp2 coarse", 17280, 1024, 128, False)):
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    lin = ops.LinearSplit(w, None, gelu=gelu, device=dev)
    blas = (lambda: ops.gelu(torch.matmul(x, w.t()))) if gelu else (lambda: torch.matmul(x, w.t()))
    ts, tb = timeit(lambda: lin(x)), timeit(blas)
    d = float((lin(x) - blas()).abs().max())
    flop = 2.0 * m * k * n
    print(f"{name:12s} [{m}x{k}]->{n}: split {ts:7.1f} us ({flop / ts / 1e6:6.1f} TF/s)   BLAS{'+gelu' if gelu else ''} {tb:7.1f} us ({flop / tb / 1e6:6.1f} TF/s)   max diff {d:.2e}", flush=True)
