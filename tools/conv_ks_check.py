#!/usr/bin/env python3
"""conv_ks.hip (split-bf16 convolution, K split across the waves of a workgroup) on its own: error against an
fp64-accumulated convolution (2..6 chunks, ResConv from LDS, residuals from memory, PReLU pre-activation, ragged maps,
widths that are / are not multiples of 4, batch, cout padding), then timing against every other configuration on the
small-map layer shapes of the 1080p path.    python tools/conv_ks_check.py [reps] [--no-time]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
ks = [c for c in range(lib.drba_conv3x3_num_cfgs()) if lib.drba_conv3x3_cfg_family(c) == 3]
print("ks cfgs", ks, flush=True)
g = torch.Generator().manual_seed(11)
bad = 0
cases = [(1, 64, 64, 2, 16, "res"), (1, 64, 64, 19, 36, "res"), (2, 96, 96, 9, 72, "res"), (1, 128, 128, 34, 60, "res"),
         (2, 192, 192, 17, 30, "res"), (1, 192, 192, 17, 30, "res"), (1, 64, 40, 11, 45, "conv"), (1, 96, 16, 5, 130, "pre"),
         (1, 64, 48, 17, 64, "res2"), (2, 128, 128, 7, 33, "res"), (2, 96, 96, 68, 120, "res"), (2, 64, 64, 136, 240, "res")]
for cfg in ks:
    for (nb, cin, cout, h, w, kind) in cases:
        x = torch.randn(nb, cin, h, w, generator=g) * 3.0
        wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        xd = x.double()
        xg = x.to(dev)
        if kind == "res":
            beta = torch.rand(1, cout, 1, 1, generator=g) + 0.5
            ref = F.leaky_relu(F.conv2d(xd, wt.double(), b.double(), padding=1) * beta.double() + xd, 0.2)
            got = ops.Conv3x3(wt, b, 1, True, beta, device=dev, cfg=cfg)(xg, residual=xg)
        elif kind == "res2":
            r1, r2 = torch.randn(nb, cout, h, w, generator=g), torch.randn(nb, cout, h, w, generator=g)
            ref = F.conv2d(xd, wt.double(), b.double(), padding=1) + r1.double() + r2.double()
            got = ops.Conv3x3(wt, b, 1, None, None, device=dev, cfg=cfg)(xg, residual=r1.to(dev), residual2=r2.to(dev))
        elif kind == "pre":
            ref = F.conv2d(F.prelu(xd, torch.tensor([0.25], dtype=torch.float64)), wt.double(), b.double(), padding=1)
            got = ops.Conv3x3(wt, b, 1, None, None, device=dev, cfg=cfg, pre_slope=0.25)(xg)
        else:
            ref = F.leaky_relu(F.conv2d(xd, wt.double(), b.double(), padding=1), 0.2)
            got = ops.Conv3x3(wt, b, 1, True, None, device=dev, cfg=cfg)(xg)
        torch.cuda.synchronize()
        d = (got.cpu().double() - ref).abs()
        scale = float(ref.abs().max())
        err = float(d.max())
        ok = err <= 5e-6 * max(1.0, scale)
        bad += 0 if ok else 1
        msg = ""
        if not ok:
            idx = torch.nonzero(d > 5e-6 * max(1.0, scale))
            msg = f" n_bad={idx.shape[0]} first={idx[0].tolist()} last={idx[-1].tolist()}"
        print(f"{'ok ' if ok else 'BAD'} cfg{cfg} {kind:5s} [{nb}x{cin}->{cout} {h}x{w}] err={err:.3e} |ref|={scale:.2f}{msg}", flush=True)
print(f"{bad} failing", flush=True)
if "--no-time" in sys.argv:
    sys.exit(1 if bad else 0)


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


layers = [("b3.res 1080p N2", 2, 64, 136, 240), ("b2.res 1080p N2", 2, 96, 68, 120), ("b1.res 1080p N2", 2, 128, 34, 60),
          ("b0.res 1080p N2", 2, 192, 17, 30), ("b0.res 1080p N1", 1, 192, 17, 30), ("b3.res 4K N2", 2, 64, 272, 480),
          ("b2.res 4K N2", 2, 96, 136, 240), ("b1.res 4K N2", 2, 128, 68, 120), ("b0.res 4K N2", 2, 192, 34, 60),
          ("grid 64 half", 1, 64, 576, 960), ("grid 96 quarter", 1, 96, 288, 480)]
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
        try:
            res.append((timeit(lambda: layer(x, residual=x, out=out)), cfg))
        except _lib.DrbaHipError:
            pass
    other = min(r for r in res if r[1] not in ks)
    line = f"{name:18s} {flop / 1e9:6.2f} GF | best other cfg{other[1]:2d} {other[0]:6.1f} us {flop / other[0] / 1e6:6.1f} TF/s | ks: "
    line += "  ".join(f"cfg{cfg} {us:6.1f} us {flop / us / 1e6:6.1f} TF/s" for us, cfg in res if cfg in ks)
    print(line, flush=True)
sys.exit(1 if bad else 0)
