#!/usr/bin/env python3
"""tools/pmc_targets.py for the 4K scale-0.5 geometry (BASELINE.json configs[4]'s per-GPU work; `bench.py --config 4k`): the kernels
that lead the 4K step -- the scale-2 stage fused with block 4's conv0[0] (52 -> 16), the scale-4 stage-input gather, the encoder,
the final blend and block 4's 32-channel ResConv / transposed convolution at 544 x 960 -- 8 samples over six frames as the
pipeline's groups launch them, so that `roofline.traffic` is not null on the 4K line and on config 5's entry.
Same protocol as pmc_targets.py (marker launch, REPS counted launches, a calibration copy); writes gpurun_out/pmc_manifest_4k.json."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drba_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
H, W = 2176, 3840
REPS = 3
manifest = []
marker = torch.zeros(256, device=dev)


def target(name, fn, pick=-1):
    fn()
    torch.cuda.synchronize()
    ops.trace_begin()
    fn()
    rec = ops.trace_end()[pick]
    ops.affine(marker, 1.0, 0.0)
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()
    manifest.append({"name": name, "symbol": rec["name"], "label": rec["label"], "launches": REPS, "algorithmic": rec["work"],
                     "unit": rec["unit"]})


a = torch.randn(64 << 20, generator=g).to(dev)
target("calibration affine 64Mi floats (256 MiB read, 256 MiB written)", lambda: ops.affine(a, 1.5, 0.25))
del a
tmap = torch.rand(1, 1, H, W, generator=g).to(dev)
frames6 = []
for _ in range(6):
    im, ft = torch.rand(1, 3, H, W, generator=g).to(dev), torch.randn(1, 16, H, W, generator=g).to(dev)
    ops.pair_interleaved(ft)
    ops.rgbx(im)
    frames6.append((im, ft))
items8 = []
for j in range(4):
    (a0, fa), (b0, fb), (c0, fc) = frames6[j], frames6[j + 1], frames6[j + 2]
    items8 += [(b0, a0, tmap, fb, fa), (b0, c0, tmap, fb, fc)]
n = 8


def head(st, amp):  # a head output whose flow channels are smooth (low-resolution noise, bicubic), as a trained stage's are
    hh, ww = int(H / st), int(W / st)
    t = torch.randn(n, 13, hh, ww, generator=g)
    lo = torch.randn(n, 4, max(hh // 8, 2), max(ww // 8, 2), generator=g) * amp
    t[:, :4] = torch.nn.functional.interpolate(lo, size=(hh, ww), mode="bicubic", align_corners=False)
    return t.to(dev)


pyr = {32.0: head(32.0, 2.0), 16.0: head(16.0, 0.6), 8.0: head(8.0, 0.6), 4.0: head(4.0, 0.6)}
conv0s2 = ops.Conv3x3(torch.randn(16, 52, 3, 3, generator=g) * 0.05, torch.zeros(16), 2, True, None, device=dev)
target("4K stage input s=2 + conv0[0] fused (block 4, 52 -> 16), flow as 3 terms, 8 samples",
       lambda: ops.stage_conv0(items8, None, pyr[4.0], 4.0, conv0s2, terms=[(pyr[32.0], 32.0), (pyr[16.0], 16.0), (pyr[8.0], 8.0)], scale=2))
xin4 = torch.empty(n, 52, H // 4, W // 4, device=dev)
target("4K stage input s=4, flow as 2 terms, 8 samples",
       lambda: ops.stage_inputs(items8, None, pyr[8.0], 8.0, 4.0, xin4, terms=[(pyr[32.0], 32.0), (pyr[16.0], 16.0)]))
del xin4
wb_terms = [(pyr[32.0], 32.0), (pyr[16.0], 16.0), (pyr[8.0], 8.0), (pyr[4.0], 4.0)]
wb_last = head(2.0, 0.6)
target("4K warp_blend_lazy, 4 terms, last stage at scale 2, 8 samples",
       lambda: ops.warp_blend_lazy([(it[0], it[1]) for it in items8], wb_terms, wb_last, 2.0))
del wb_terms, wb_last, pyr
from drba_amd.models.rife_426_heavy.IFNet_HDv3 import Head  # noqa: E402
hsd = {"encode.cnn0.weight": torch.randn(16, 3, 3, 3, generator=g) / 27 ** 0.5, "encode.cnn0.bias": torch.zeros(16),
       "encode.cnn1.weight": torch.randn(16, 16, 3, 3, generator=g) / 12, "encode.cnn1.bias": torch.zeros(16),
       "encode.cnn2.weight": torch.randn(16, 16, 3, 3, generator=g) / 12, "encode.cnn2.bias": torch.zeros(16),
       "encode.cnn3.weight": torch.randn(16, 16, 4, 4, generator=g) / 8, "encode.cnn3.bias": torch.zeros(16)}
head_net = Head(hsd, "encode.", dev)
target("4K head_fused 2176x3840, pair layout only", lambda: head_net(frames6[0][0], planar=False))
x = torch.randn(n, 32, 272, 480, generator=g).to(dev)
layer = ops.Conv3x3(torch.randn(32, 32, 3, 3, generator=g) * 0.05, torch.zeros(32), 1, True, torch.ones(1, 32, 1, 1), device=dev)
out = torch.empty_like(x)
target("block 4 ResConv 32->32ch 272x480 N8 (4K scale 0.5: the same map as 1080p)", lambda: layer(x, residual=x, out=out))
last = ops.Deconv4x4(torch.randn(32, 20, 4, 4, generator=g) * 0.05, torch.zeros(20), pixel_shuffle=True, device=dev)
target("block 4 lastconv 32->20 deconv + PixelShuffle 272x480 N8", lambda: last(x))
torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(manifest, open(os.path.join(ROOT, "gpurun_out", "pmc_manifest_4k.json"), "w"), indent=1)
print(json.dumps(manifest, indent=1))
