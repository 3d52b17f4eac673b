#!/usr/bin/env python3
"""Time every kernel configuration on the IFNet layer shapes (1080p net size by default).
    python tools/conv_bench.py [H W]
Prints per layer: the picked config, and for each valid config the average microseconds and TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import _lib, ops  # noqa: E402

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("H", nargs="?", type=int, default=1088)
ap.add_argument("W", nargs="?", type=int, default=1920)
ap.add_argument("--only", default=None, help="comma list of layer names (e.g. b4.res,b3.res)")
ap.add_argument("--cfg", type=int, default=None, help="run only this config id")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
H, W = args.H, args.W
dev = torch.device("cuda:0")
lib = _lib.load()
S1, S2, NDC = list(range(0, 8)), list(range(8, 14)), 6


def timeit(fn, reps=None):
    reps = reps or args.reps
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


layers = []
C, IN = (192, 128, 96, 64, 32), (39, 52, 52, 52, 52)
for i, (c, cin) in enumerate(zip(C, IN)):
    s = (16, 8, 4, 2, 1)[i]
    h, w = H // s, W // s
    layers.append((f"b{i}.conv0.0", "conv", cin, c // 2, h, w, 2))
    layers.append((f"b{i}.conv0.1", "conv", c // 2, c, h // 2, w // 2, 2))
    layers.append((f"b{i}.res", "res", c, c, h // 4, w // 4, 1))
    layers.append((f"b{i}.last", "deconv", c, 52, h // 4, w // 4, 1))
layers += [("enc.cnn0", "conv", 3, 16, H, W, 2), ("enc.cnn1", "conv", 16, 16, H // 2, W // 2, 1),
           ("enc.cnn3", "deconv16", 16, 16, H // 2, W // 2, 1)]

g = torch.Generator().manual_seed(0)
only = set(args.only.split(",")) if args.only else None
for name, kind, cin, cout, h, w, stride in layers:
    if only and name not in only:
        continue
    x = torch.randn(1, cin, h, w, generator=g).to(dev)
    b = torch.zeros(cout)
    if kind in ("conv", "res"):
        wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        flops = 2.0 * cout * cin * 9 * ho * wo
        picked = lib.drba_conv3x3_pick_cfg(cin, cout, ho, wo, stride)
        res = []
        for cfg in ([args.cfg] if args.cfg is not None else (S1 if stride == 1 else S2)):
            beta = torch.ones(1, cout, 1, 1) if kind == "res" else None
            layer = ops.Conv3x3(wt, b, stride, True, beta, device=dev, cfg=cfg)
            out = torch.empty((1, cout, ho, wo), device=dev)
            us = timeit(lambda: layer(x, residual=x if kind == "res" else None, out=out))
            res.append((cfg, us))
    else:
        wt = torch.randn(cin, cout, 4, 4, generator=g) * 0.05
        flops = 2.0 * cout * cin * 16 * h * w
        picked = lib.drba_deconv4x4_pick_cfg(cin, cout, h, w)
        res = []
        for cfg in ([args.cfg] if args.cfg is not None else range(NDC)):
            layer = ops.Deconv4x4(wt, b, kind == "deconv", device=dev, cfg=cfg)
            us = timeit(lambda: layer(x))
            res.append((cfg, us))
    best = min(res, key=lambda r: r[1])
    line = " ".join(f"{'*' if c == picked else ''}c{c}:{us:.0f}us" for c, us in res)
    pk = dict(res).get(picked, best[1])
    print(f"{name:12s} {cin:3d}->{cout:3d} {h}x{w} s{stride} {flops / 1e9:6.2f}GF picked c{picked} {pk:.0f}us {flops / pk / 1e6:6.1f}TF/s"
          f" | best c{best[0]} {best[1]:.0f}us {flops / best[1] / 1e6:6.1f}TF/s | {line}")
