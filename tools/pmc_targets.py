#!/usr/bin/env python3
"""Launch the roofline-target kernels of bench.py three times each (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)
and write the launch manifest tools/pmc_traffic.py needs.  A plain elementwise kernel with a known byte count
(drba_affine on 64 Mi floats: 256 MiB read + 256 MiB written, 4 B per lane like the targets) calibrates the counters,
as MI355X_MICROARCH.md 'HBM' prescribes.  Names are the ones bench.py prints in `roofline.kernel`."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drba_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
H, W = 1088, 1920
REPS = 3
manifest = []  # (name, substring of the kernel symbol, launches), in launch order


marker = torch.zeros(256, device=dev)


def target(name, sym, fn):
    """A 256-element affine launch marks the start of each target's segment in the dispatch order; the target's
    counted launches are the last REPS dispatches of its symbol in the segment (autotune launches come before them)."""
    ops.affine(marker, 1.0, 0.0)
    fn()  # warm / autotune
    torch.cuda.synchronize()
    for _ in range(REPS):
        fn()
    manifest.append({"name": name, "symbol": sym, "launches": REPS})


a = torch.randn(64 << 20, generator=g).to(dev)
target("calibration affine 64Mi floats (256 MiB read, 256 MiB written)", "affine_kernel", lambda: ops.affine(a, 1.5, 0.25))
img0, img1 = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 3, H, W, generator=g).to(dev)
f0, f1 = torch.randn(1, 16, H, W, generator=g).to(dev), torch.randn(1, 16, H, W, generator=g).to(dev)
flow = torch.nn.functional.interpolate(torch.randn(1, 4, H // 32, W // 32, generator=g) * 6, size=(H, W), mode="bilinear").to(dev).contiguous()
tmap = torch.rand(1, 1, H, W, generator=g).to(dev)
for s in (1.0, 2.0):
    tprev = torch.randn(1, 13, int(H / (2 * s)), int(W / (2 * s)), generator=g).to(dev)
    h, w = int(H / s), int(W / s)
    target(f"ifblock_input_kernel<true> 52ch {H}x{W} -> {h}x{w}", "ifblock_input",
           lambda: ops.ifblock_input(img0, img1, f0, f1, tmap, flow, tprev, 2 * s, s))
convs = []
for (c, h, w, n) in ((64, 136, 240, 2), (32, 272, 480, 2)):
    x = torch.randn(n, c, h, w, generator=g).to(dev)
    layer = ops.Conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.05, torch.zeros(c), 1, True, torch.ones(1, c, 1, 1), device=dev)
    out = torch.empty_like(x)
    layer(x, residual=x, out=out)  # autotune (before any conv segment starts: its launches carry the same symbols)
    kern = "conv_split_mfma" if ops._tuned[("conv3x3", n, c, c, h, w, 1)] >= 14 else "conv_mfma"
    convs.append((f"{kern} {c}->{c}ch {h}x{w} s1 N{n} (ResConv)", kern, layer, x, out))
for name, kern, layer, x, out in convs:
    target(name, kern, lambda: layer(x, residual=x, out=out))
torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(manifest, open(os.path.join(ROOT, "gpurun_out", "pmc_manifest.json"), "w"), indent=1)
