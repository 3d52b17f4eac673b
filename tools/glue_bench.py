#!/usr/bin/env python3
"""Time the non-GEMM kernels at a given net size (default 1088x1920).
    DRBA_IFIN_VARIANT=<0|1|2> python tools/glue_bench.py [H W]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import ops  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1088, 1920)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
P = H * W


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev)


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


img0, img1 = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 3, H, W, generator=g).to(dev)
f0, f1 = rnd(1, 16, H, W), rnd(1, 16, H, W)
# smooth flow of a few pixels (what the network produces), not white noise
flow = torch.nn.functional.interpolate(torch.randn(1, 4, H // 32, W // 32, generator=g) * 6, size=(H, W), mode="bilinear").to(dev).contiguous()
tmap = torch.rand(1, 1, H, W, generator=g).to(dev)
print(f"variant={os.environ.get('DRBA_IFIN_VARIANT', 'default')} size {H}x{W}")
for s in (16.0, 8.0, 4.0, 2.0, 1.0):
    sp = 2 * s
    tprev = rnd(1, 13, int(H / sp), int(W / sp))
    us = timeit(lambda: ops.ifblock_input(img0, img1, f0, f1, tmap, flow, tprev, sp, s))
    h, w = int(H / s), int(W / s)
    pts = P if s <= 2 else 4 * h * w
    nbytes = 4.0 * (43 * pts + 52 * h * w)
    print(f"ifblock_input s={s:4.0f}: {us:7.1f} us  {nbytes / us / 1e3:7.1f} GB/s")
us = timeit(lambda: ops.ifblock_input(img0, img1, f0, f1, 0.5, None, None, 1.0, 16.0))
print(f"ifblock_input first s=16: {us:7.1f} us")
for s in (16.0, 4.0, 1.0):
    tmp = rnd(1, 13, int(H / s), int(W / s))
    us = timeit(lambda: ops.ifblock_update(tmp, flow, H, W, s))
    print(f"ifblock_update s={s:4.0f}: {us:7.1f} us  {4.0 * 8 * P / us / 1e3:7.1f} GB/s")
tl = rnd(1, 13, H, W)
us = timeit(lambda: ops.warp_blend(img0, img1, flow, tl, 1.0))
print(f"warp_blend: {us:7.1f} us  {4.0 * 14 * P / us / 1e3:7.1f} GB/s")
fl2 = flow[:, :2].contiguous()
us = timeit(lambda: ops.flow_reverse(fl2))
print(f"flow_reverse: {us:7.1f} us")
us = timeit(lambda: ops.drm_rife_linear(fl2, flow[:, 2:].contiguous(), 0.25))
print(f"drm_rife_linear: {us:7.1f} us")
us = timeit(lambda: ops.softsplat(fl2, fl2, None, "avg"))
print(f"softsplat avg C=2 (generic, global atomics): {us:7.1f} us")
us = timeit(lambda: ops.backwarp(f0, fl2))
print(f"backwarp C=16: {us:7.1f} us  {4.0 * 34 * P / us / 1e3:7.1f} GB/s")
us = timeit(lambda: ops.resize_bilinear(img0, (1080, 1920)))
print(f"resize 3ch: {us:7.1f} us")
