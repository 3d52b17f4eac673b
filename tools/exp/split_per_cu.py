#!/usr/bin/env python3
"""conv_split_mfma's persistent grid: workgroups per CU by LDS alone (3) against what registers AND LDS keep resident.
Run on the tuning library (tools/exp/build_tuning.sh), once per DRBA_SPLIT_PER_CU value:
    DRBA_SPLIT_PER_CU=3 python tools/exp/split_per_cu.py ; DRBA_SPLIT_PER_CU=2 python tools/exp/split_per_cu.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
reps = 20


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


layers = [("b4 32ch 1080p N8", 8, 32, 272, 480), ("b3 64ch 1080p N8", 8, 64, 136, 240), ("b2 96ch 1080p N8", 8, 96, 68, 120),
          ("b1 128ch 1080p N8", 8, 128, 34, 60), ("b0 192ch 1080p N8", 8, 192, 17, 30), ("b3 64ch 4K N8", 8, 64, 272, 480),
          ("b2 96ch 4K N8", 8, 96, 136, 240), ("grid 32 full", 1, 32, 1152, 1920), ("grid 64 half", 1, 64, 576, 960),
          ("grid 96 quarter", 1, 96, 288, 480)]
g = torch.Generator().manual_seed(0)
first = lib.drba_conv3x3_num_cfgs()
print("DRBA_SPLIT_PER_CU =", os.environ.get("DRBA_SPLIT_PER_CU", "(resident)"))
for name, n, c, h, w in layers:
    x = torch.randn(n, c, h, w, generator=g).to(dev)
    wt = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    b = torch.randn(c, generator=g) * 0.1
    beta = torch.rand(1, c, 1, 1, generator=g) + 0.5
    out = torch.empty_like(x)
    res = []
    for cfg in range(lib.drba_conv3x3_num_cfgs()):
        if lib.drba_conv3x3_cfg_stride(cfg) != 1 or lib.drba_conv3x3_cfg_family(cfg) != 4 or lib.drba_conv3x3_packed_floats(c, c, cfg) == 0:
            continue
        layer = ops.Conv3x3(wt, b, 1, True, beta, device=dev, cfg=cfg)
        res.append((timeit(lambda: layer(x, residual=x, out=out)), cfg))
    print(f"{name:18s} " + "  ".join(f"cfg{cfg} {us:6.1f}" for us, cfg in res), flush=True)
# stride 2 (IFBlock conv0), two-term tiles only
for name, n, cin, cout, h, w in [("conv0 52->48 s2 N8", 8, 52, 48, 272, 480), ("conv0 48->96 s2 N8", 8, 48, 96, 136, 240),
                                 ("conv0 32->64 s2 N8", 8, 32, 64, 272, 480), ("conv0 52->64 s2 N8", 8, 52, 64, 136, 240)]:
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    res = []
    for cfg in range(lib.drba_conv3x3_num_cfgs()):
        if lib.drba_conv3x3_cfg_stride(cfg) != 2 or lib.drba_conv3x3_cfg_family(cfg) != 4 or lib.drba_conv3x3_packed_floats(cin, cout, cfg) == 0:
            continue
        layer = ops.Conv3x3(wt, b, 2, True, None, device=dev, cfg=cfg)
        res.append((timeit(lambda: layer(x)), cfg))
    print(f"{name:22s} " + "  ".join(f"cfg{cfg} {us:6.1f}" for us, cfg in res), flush=True)
# the transposed form (IFBlock lastconv): 32 -> 20 at 272x480 and 64 -> 52 at 136x240, N8
for name, n, cin, cout, h, w in [("lastconv b4 32->20 N8", 8, 32, 20, 272, 480), ("lastconv b3 64->52 N8", 8, 64, 52, 136, 240),
                                 ("lastconv b2 96->52 N8", 8, 96, 52, 68, 120)]:
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    res = []
    for cfg in range(lib.drba_deconv4x4_num_cfgs()):
        if lib.drba_deconv4x4_cfg_family(cfg) != 4 or lib.drba_deconv4x4_packed_floats(cin, cout, cfg) == 0:
            continue
        layer = ops.Deconv4x4(wt, b, pixel_shuffle=True, device=dev, cfg=cfg)
        res.append((timeit(lambda: layer(x)), cfg))
    print(f"{name:22s} " + "  ".join(f"cfg{cfg} {us:6.1f}" for us, cfg in res), flush=True)
