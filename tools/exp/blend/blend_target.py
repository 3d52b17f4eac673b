#!/usr/bin/env python3
"""Three launches of drba_warp_blend_lazy_batch as the loop launches it (8 samples over six frames with their [H,W,4] copies, four
terms), at 1080p (last stage at scale 1) or 4K scale 0.5 (last stage at scale 2), for timing and rocprofv3 --pmc passes.
    python tools/exp/blend/blend_target.py [1080p|4k] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from drba_amd import ops  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "1080p"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
(H, W), last = ((1088, 1920), 1.0) if cfg == "1080p" else ((2176, 3840), 2.0)
B = 8
fr = []
for _ in range(B // 2 + 2):
    im = torch.rand(1, 3, H, W, generator=g).to(dev)
    ops.rgbx(im)
    fr.append(im)
items = []
for j in range(B // 2):
    a, b, c = fr[j], fr[j + 1], fr[j + 2]
    items += [(b, a), (b, c)]


def head(st, amp):
    hh, ww = int(H / st), int(W / st)
    t = torch.randn(B, 13, hh, ww, generator=g)
    lo = torch.randn(B, 4, max(hh // 8, 2), max(ww // 8, 2), generator=g) * amp
    t[:, :4] = torch.nn.functional.interpolate(lo, size=(hh, ww), mode="bicubic", align_corners=False)
    return t.to(dev)


terms = [(head(16 * last, 1.0), 16.0 * last), (head(8 * last, 0.4), 8.0 * last), (head(4 * last, 0.3), 4.0 * last), (head(2 * last, 0.3), 2.0 * last)]
tl = head(last, 0.3)
out = ops.warp_blend_lazy(items, terms, tl, last)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = ops.warp_blend_lazy(items, terms, tl, last)
e1.record()
torch.cuda.synchronize()
P = H * W
print(f"warp_blend_lazy {cfg} 8 samples: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per launch, "
      f"{B * 4.0 * 13 * P / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12:.2f} TB/s of algorithmic bytes; checksum {float(out[3].double().sum()):.6f}")
