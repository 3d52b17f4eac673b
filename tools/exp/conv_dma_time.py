#!/usr/bin/env python3
"""Times the conv_dma configuration(s) of every library under tools/exp/build/ (experiment builds of conv_dma.hip, see
conv_dma_variants.sh) on the 1080p ResConv shapes, one subprocess per library.  python tools/exp/conv_dma_time.py [reps]"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, torch
sys.path.insert(0, %r)
from drba_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from drba_amd import ops
lib = _lib.load()
dev = torch.device("cuda:0")
reps = int(sys.argv[2])
g = torch.Generator().manual_seed(0)
out = []
only = os.environ.get("SHAPES")
warm = int(os.environ.get("WARM", "3"))
for name, n, c, h, w in (("b4", 2, 32, 272, 480), ("b3", 2, 64, 136, 240), ("b2", 2, 96, 68, 120), ("b1", 2, 128, 34, 60), ("g64", 1, 64, 576, 960)):
    if only and name not in only.split(","):
        continue
    x = torch.randn(n, c, h, w, generator=g).to(dev)
    wt = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    b = torch.randn(c, generator=g) * 0.1
    beta = torch.rand(1, c, 1, 1, generator=g) + 0.5
    o = torch.empty_like(x)
    for cfg in range(lib.drba_conv3x3_num_cfgs()):
        if lib.drba_conv3x3_cfg_family(cfg) != 2:
            continue
        layer = ops.Conv3x3(wt, b, 1, True, beta, device=dev, cfg=cfg)
        for _ in range(warm):
            layer(x, residual=x, out=o)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            layer(x, residual=x, out=o)
        e1.record()
        torch.cuda.synchronize()
        out.append(f"{name}/cfg{cfg} {e0.elapsed_time(e1) / reps * 1e3:6.1f}")
print("  ".join(out))
''' % ROOT
reps = sys.argv[1] if len(sys.argv) > 1 else "30"
for lib in sorted(glob.glob(os.path.join(ROOT, "tools", "exp", "build", "libdrba_hip_*.so"))):
    r = subprocess.run([sys.executable, "-c", CHILD, lib, reps], capture_output=True, text=True, timeout=300)
    tag = os.path.basename(lib)[len("libdrba_hip_"):-3]
    print(f"{tag:28s} {r.stdout.strip() or r.stderr[-400:]}", flush=True)
