#!/usr/bin/env python3
"""Two-term fp16 split (cfg family 4) against the three-term bf16 families on the step's stride-1 layers: time per launch
and max error against the fp64 convolution (a 1-item slice), per configuration.
    python tools/exp/f16_split_probe.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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


layers = [("b4.res 32ch 272x480 N8", 8, 32, 272, 480), ("b3.res 64ch 136x240 N8", 8, 64, 136, 240), ("b2.res 96ch 68x120 N8", 8, 96, 68, 120),
          ("b1.res 128ch 34x60 N8", 8, 128, 34, 60), ("b0.res 192ch 17x30 N8", 8, 192, 17, 30), ("b0.res 192ch 17x30 N4", 4, 192, 17, 30),
          ("grid 32ch 1152x1920", 1, 32, 1152, 1920), ("grid 64ch 576x960", 1, 64, 576, 960), ("grid 96ch 288x480", 1, 96, 288, 480)]
g = torch.Generator().manual_seed(0)
fam_name = {1: "bf16x3 reg", 2: "bf16x3 dma", 3: "bf16x3 ks", 4: "f16x2"}
for name, n, c, h, w in layers:
    x = (torch.randn(n, c, h, w, generator=g) * 2.0).to(dev)
    wt = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    b = torch.randn(c, generator=g) * 0.1
    beta = torch.rand(1, c, 1, 1, generator=g) + 0.5
    hs = min(h, 64)
    xs = x[:1, :, :hs].double().cpu()
    ref = F.leaky_relu(F.conv2d(xs, wt.double(), b.double(), padding=1) * beta.double() + xs, 0.2)[:, :, :hs - 1]
    best = {}
    for cfg in range(lib.drba_conv3x3_num_cfgs()):
        fam = lib.drba_conv3x3_cfg_family(cfg)
        if fam == 0 or lib.drba_conv3x3_packed_floats(c, c, cfg) == 0:
            continue
        layer = ops.Conv3x3(wt, b, 1, True, beta, device=dev, cfg=cfg)
        try:
            us = timeit(lambda: layer(x, residual=x))
        except Exception as e:  # noqa: BLE001
            print(name, cfg, "EXC", e)
            continue
        got = layer(x, residual=x)[:1, :, :hs - 1].double().cpu()
        err = float((got - ref).abs().max())
        if fam not in best or us < best[fam][0]:
            best[fam] = (us, cfg, err)
    print(f"{name:26s} |ref|max {float(ref.abs().max()):6.2f} | " + " | ".join(
        f"{fam_name[f]} cfg{v[1]} {v[0]:7.1f} us err {v[2]:.2e}" for f, v in sorted(best.items())), flush=True)
