#!/usr/bin/env python3
"""One gather kernel of the pipeline, launched `reps` times as the loop launches it (8 samples over six frames with their [H,W,4]
copies and pair-interleaved features, the flow as terms), for timing and for tools/exp/kernel_pmc.sh.
    python tools/exp/gather_target.py <blend|s8|s4|s2|s2conv|s1conv|drm|rev|head> <1080p|4k> [reps]
1080p: scale list 16, 8, 4, 2, 1; 4k (scale 0.5): 32, 16, 8, 4, 2 -- `s4` is the stage-input gather that feeds a block's conv0[0]
at (frame / 4 x last scale) resolution, etc.: the name is the stage's scale relative to the LAST stage's."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd import ops  # noqa: E402

what, cfg = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
(H, W), last = ((1088, 1920), 1.0) if cfg == "1080p" else ((2176, 3840), 2.0)
B = 8
tmap = torch.rand(1, 1, H, W, generator=g).to(dev)
fr = []
for _ in range(B // 2 + 2):
    im, ft = torch.rand(1, 3, H, W, generator=g).to(dev), torch.randn(1, 16, H, W, generator=g).to(dev)
    ops.rgbx(im)
    ops.pair_interleaved(ft)
    fr.append((im, ft))
items = []
for j in range(B // 2):
    (a, fa), (b, fb), (c, fc) = fr[j], fr[j + 1], fr[j + 2]
    items += [(b, a, tmap, fb, fa), (b, c, tmap, fb, fc)]


def head(st, amp):
    hh, ww = int(H / st), int(W / st)
    t = torch.randn(B, 13, hh, ww, generator=g)
    lo = torch.randn(B, 4, max(hh // 8, 2), max(ww // 8, 2), generator=g) * amp
    t[:, :4] = torch.nn.functional.interpolate(lo, size=(hh, ww), mode="bicubic", align_corners=False)
    return t.to(dev)


S = {k: k * last for k in (16, 8, 4, 2, 1)}
pyr = {k: head(S[k], 1.0 if k == 16 else 0.3) for k in (16, 8, 4, 2, 1)}
def smooth_flow(n, amp):
    lo = torch.randn(n, 2, H // 64, W // 64, generator=g) * amp
    return torch.nn.functional.interpolate(lo, size=(H, W), mode="bicubic", align_corners=False).to(dev).contiguous()


if what in ("drm", "rev"):
    fa, fb = smooth_flow(8, 3.0), smooth_flow(8, 3.0)
    if what == "drm":
        jobs = [(fa[k:k + 1], fb[k:k + 1], 0.25 + 0.05 * k) for k in range(8)]
        fn = lambda: ops.drm_rife_linear_many(jobs, 1e-4)  # noqa: E731
        nbytes = 8 * 20.0 * H * W
    else:
        fn = lambda: ops.flow_reverse(fa)  # noqa: E731
        nbytes = 8 * 16.0 * H * W
elif what in ("s2c52", "s2c32", "s2c48", "s2c39", "s2c64", "s2c16"):  # the stride-2 convolutions of the IFBlocks (conv0[0] / conv0[1]), 8 samples
    cin, cout, h, w = {"s2c52": (52, 48, 272, 480), "s2c32": (32, 64, 272, 480), "s2c48": (48, 96, 136, 240), "s2c39": (39, 96, 68, 120),
                       "s2c64": (64, 128, 68, 120), "s2c16": (16, 32, 544, 960)}[what]
    x = torch.randn(8, cin, h, w, generator=g).to(dev)
    layer = ops.Conv3x3(torch.randn(cout, cin, 3, 3, generator=g) * 0.05, torch.zeros(cout), 2, True, None, device=dev)
    fn = lambda: layer(x)  # noqa: E731
    nbytes = None
elif what in ("conv64g", "conv64r", "conv32r", "deconv4", "conv96r", "conv128r", "conv192r", "conv96g"):
    if what == "deconv4":
        x = torch.randn(8, 32, 272, 480, generator=g).to(dev)
        layer = ops.Deconv4x4(torch.randn(32, 20, 4, 4, generator=g) * 0.05, torch.zeros(20), pixel_shuffle=True, device=dev)
        fn = lambda: layer(x)  # noqa: E731
    else:
        n, c, h, w, pre = {"conv64g": (1, 64, 576, 960, 0.25), "conv64r": (8, 64, 136, 240, None), "conv32r": (8, 32, 272, 480, None), "conv96r": (8, 96, 68, 120, None),
                               "conv128r": (8, 128, 34, 60, None), "conv192r": (8, 192, 17, 30, None), "conv96g": (1, 96, 288, 480, 0.25)}[what]
        x = torch.randn(n, c, h, w, generator=g).to(dev)
        if pre is None:
            layer = ops.Conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.05, torch.zeros(c), 1, True, torch.ones(1, c, 1, 1), device=dev)
            out = torch.empty_like(x)
            fn = lambda: layer(x, residual=x, out=out)  # noqa: E731
        else:
            layer = ops.Conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.05, torch.zeros(c), 1, None, None, device=dev, pre_slope=pre)
            out = torch.empty_like(x)
            fn = lambda: layer(x, residual=x, out=out)  # noqa: E731
    nbytes = None
elif what in ("attn", "mlp", "merge"):
    if what == "attn":  # GMFlow's fine-scale window attention at 1080p: 2 x 34560 tokens, 8 x 8 windows (18 x 30 tokens each), shifted
        qkv = torch.randn(2, 144 * 240, 384, generator=g).to(dev)
        fn = lambda: ops.window_attention(qkv[..., 0:128], qkv[..., 128:256], qkv[..., 256:384], 144, 240, 8, True, 128 ** -0.5)  # noqa: E731
    elif what == "mlp":
        lin = ops.LinearSplit(torch.randn(1024, 256, generator=g) * 0.05, torch.zeros(1024), gelu=True, device=dev)
        xt = torch.randn(2 * 144 * 240, 256, generator=g).to(dev)
        fn = lambda: lin(xt)  # noqa: E731
    else:
        lin = ops.LinearSplit(torch.randn(128, 128, generator=g) * 0.05, None, device=dev)
        xt = torch.randn(2 * 144 * 240, 128, generator=g).to(dev)
        lw, lb = torch.ones(128).to(dev), torch.zeros(128).to(dev)
        fn = lambda: lin.layernorm(xt, lw, lb, residual=xt)  # noqa: E731
    nbytes = None
elif what == "head":
    from drba_amd.models.rife_426_heavy.IFNet_HDv3 import Head
    hsd = {"encode.cnn0.weight": torch.randn(16, 3, 3, 3, generator=g) / 27 ** 0.5, "encode.cnn0.bias": torch.zeros(16),
           "encode.cnn1.weight": torch.randn(16, 16, 3, 3, generator=g) / 12, "encode.cnn1.bias": torch.zeros(16),
           "encode.cnn2.weight": torch.randn(16, 16, 3, 3, generator=g) / 12, "encode.cnn2.bias": torch.zeros(16),
           "encode.cnn3.weight": torch.randn(16, 16, 4, 4, generator=g) / 8, "encode.cnn3.bias": torch.zeros(16)}
    head_net = Head(hsd, "encode.", dev)
    fn = lambda: head_net(fr[0][0], planar=False)  # noqa: E731
    nbytes = 4.0 * 19 * H * W
elif what == "blend":
    terms = [(pyr[k], S[k]) for k in (16, 8, 4, 2)]
    fn = lambda: ops.warp_blend_lazy([(it[0], it[1]) for it in items], terms, pyr[1], S[1])  # noqa: E731
    nbytes = B * 4.0 * 13 * H * W
elif what in ("s8", "s4", "s2"):
    k = int(what[1:])
    xin = torch.empty(B, 52, int(H / S[k]), int(W / S[k]), device=dev)
    terms = [(pyr[j], S[j]) for j in (16, 8, 4, 2) if j > 2 * k]
    fn = lambda: ops.stage_inputs(items, None, pyr[2 * k], S[2 * k], S[k], xin, terms=terms)  # noqa: E731
    nbytes = None
else:
    k = 2 if what == "s2conv" else 1
    if S[k] not in (1.0, 2.0):
        raise SystemExit(f"{what} at {cfg}: the fused stage kernels take the stage at frame scale 1 or 2")
    cout = 32 if (cfg == "1080p" and k == 2) else 16
    conv = ops.Conv3x3(torch.randn(cout, 52, 3, 3, generator=g) * 0.05, torch.zeros(cout), 2, True, None, device=dev)
    terms = [(pyr[j], S[j]) for j in (16, 8, 4, 2) if j > 2 * k]
    fn = lambda: ops.stage_conv0(items, None, pyr[2 * k], S[2 * k], conv, terms=terms, scale=int(S[k]))  # noqa: E731
    nbytes = None
fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"{what} {cfg} 8 samples: {us:.1f} us per launch" + (f", {nbytes / us / 1e6:.2f} TB/s of algorithmic bytes" if nbytes else ""))
