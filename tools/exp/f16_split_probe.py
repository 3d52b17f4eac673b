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
        if fam == 0 or lib.drba_conv3x3_cfg_stride(cfg) != 1 or lib.drba_conv3x3_packed_floats(c, c, cfg) == 0:
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

# the stride-2 conv0 layers of the step: fp32 MFMA configurations against the two-term stride-2 tiles
print()
for name, n, cin, cout, h, w in (("b4.conv0.1 16->32 544x960 N8", 8, 16, 32, 544, 960), ("b3.conv0.0 52->32 544x960 N8", 8, 52, 32, 544, 960),
                                 ("b3.conv0.1 32->64 272x480 N8", 8, 32, 64, 272, 480), ("b2.conv0.0 52->48 272x480 N8", 8, 52, 48, 272, 480),
                                 ("b2.conv0.1 48->96 136x240 N8", 8, 48, 96, 136, 240), ("b1.conv0.0 52->64 136x240 N8", 8, 52, 64, 136, 240),
                                 ("b1.conv0.1 64->128 68x120 N8", 8, 64, 128, 68, 120), ("b0.conv0.0 39->96 68x120 N8", 8, 39, 96, 68, 120),
                                 ("b0.conv0.1 96->192 34x60 N8", 8, 96, 192, 34, 60)):
    x = (torch.randn(n, cin, h, w, generator=g) * 2.0).to(dev)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    hs = min(h, 64)
    ref = F.leaky_relu(F.conv2d(x[:1, :, :hs].double().cpu(), wt.double(), b.double(), stride=2, padding=1), 0.2)[:, :, :hs // 2 - 1]
    best = {}
    for cfg in range(lib.drba_conv3x3_num_cfgs()):
        if lib.drba_conv3x3_cfg_stride(cfg) != 2 or lib.drba_conv3x3_packed_floats(cin, cout, cfg) == 0:
            continue
        fam = lib.drba_conv3x3_cfg_family(cfg)
        layer = ops.Conv3x3(wt, b, 2, True, None, device=dev, cfg=cfg)
        try:
            us = timeit(lambda: layer(x))
        except Exception as e:  # noqa: BLE001
            print(name, cfg, "EXC", e)
            continue
        err = float((layer(x)[:1, :, :hs // 2 - 1].double().cpu() - ref).abs().max())
        best.setdefault(fam, []).append((us, cfg, err))
    print(f"{name:30s} |ref|max {float(ref.abs().max()):6.2f} | " + " | ".join(
        ("fp32" if f == 0 else "f16x2") + " " + " ".join(f"cfg{c}:{u:.0f}us({e:.1e})" for u, c, e in sorted(v)[:4]) for f, v in sorted(best.items())), flush=True)
