#!/usr/bin/env python3
"""Launch the roofline-target kernels of bench.py three times each (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)
and write the launch manifest tools/pmc_traffic.py needs.  A plain elementwise kernel with a known byte count
(drba_affine on 64 Mi floats: 256 MiB read + 256 MiB written, 4 B per lane like the targets) calibrates the counters,
as MI355X_MICROARCH.md 'HBM' prescribes.  Every target is first run once under the library's own kernel trace, which
yields the exact kernel symbol and the launch label bench.py prints in `roofline.kernel` / `by_geometry[].launch`:
profiles/pmc_traffic.json is keyed {symbol: {label: bytes per launch}}."""
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
manifest = []  # {name, symbol, label, launches, algorithmic}, in launch order
marker = torch.zeros(256, device=dev)


def target(name, fn, pick=-1):
    """A 256-element affine launch marks the start of each target's segment in the dispatch order; the target's
    counted launches are the FIRST REPS dispatches of its symbol after the marker (the next target's warm-up and autotune
    launches follow them in the same segment)."""
    fn()  # warm / autotune, outside any segment that matters (its launches precede the marker)
    torch.cuda.synchronize()
    ops.trace_begin()
    fn()
    rec = ops.trace_end()[pick]  # the kernel of interest is the last launch of the call unless told otherwise
    ops.affine(marker, 1.0, 0.0)
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()
    manifest.append({"name": name, "symbol": rec["name"], "label": rec["label"], "launches": REPS, "algorithmic": rec["work"],
                     "unit": rec["unit"]})


a = torch.randn(64 << 20, generator=g).to(dev)
target("calibration affine 64Mi floats (256 MiB read, 256 MiB written)", lambda: ops.affine(a, 1.5, 0.25))
img0, img1 = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 3, H, W, generator=g).to(dev)
f0, f1 = torch.randn(1, 16, H, W, generator=g).to(dev), torch.randn(1, 16, H, W, generator=g).to(dev)
flow = torch.nn.functional.interpolate(torch.randn(1, 4, H // 32, W // 32, generator=g) * 6, size=(H, W), mode="bilinear").to(dev).contiguous()
tmap = torch.rand(1, 1, H, W, generator=g).to(dev)
ops.pair_interleaved(f0), ops.pair_interleaved(f1)  # made once, as in the pipeline (calc_flow)
# the pipeline launches a stage's input kernel once for BOTH frames of a `-t 2` step (blockIdx.y = item): same here
img2 = torch.rand(1, 3, H, W, generator=g).to(dev)
f2 = torch.randn(1, 16, H, W, generator=g).to(dev)
ops.pair_interleaved(f2)
for _im in (img0, img1, img2):
    ops.rgbx(_im)  # the [H,W,4] copies the pipeline's frames carry (ops.to_inp writes them): the gathers read their image taps from those
items2 = [(img1, img0, tmap, f1, f0), (img1, img2, tmap, f1, f2)]
flows2 = [flow, (flow * 0.9).contiguous()]
# N = 2: the samples of one `-t 2` step (what a step-by-step driver launches); N = 8: a group of 4 steps (RIFE.GROUP, what
# bench.py's loop launches) -- the launch label ends in N, so both geometries get their own row
# a group of 4 steps (8 items) reads SIX distinct frames the way the pipeline's items do -- step j: (I[j+1], I[j]) and (I[j+1], I[j+2]) --
# so that what the items share in L2 (round 4: the items of a tile run back to back on one XCD) is what the pipeline shares
frames6 = [(img0, f0), (img1, f1), (img2, f2)]
for _ in range(3):
    im, ft = torch.rand(1, 3, H, W, generator=g).to(dev), torch.randn(1, 16, H, W, generator=g).to(dev)
    ops.pair_interleaved(ft)
    ops.rgbx(im)
    frames6.append((im, ft))
items8 = []
for j in range(4):
    (a, fa), (b, fb), (c, fc) = frames6[j], frames6[j + 1], frames6[j + 2]
    items8 += [(b, a, tmap, fb, fa), (b, c, tmap, fb, fc)]
flows8 = [flow, (flow * 0.9).contiguous()] * 4
for n in (2, 8):
    items, flows = (items2, flows2) if n == 2 else (items8, flows8)
    for s in (1.0, 2.0):
        tprev = torch.randn(n, 13, int(H / (2 * s)), int(W / (2 * s)), generator=g).to(dev)
        xin = torch.empty(n, 52, int(H / s), int(W / s), device=dev)
        target(f"stage input s={s:.0f} with the folded flow update, {n} samples",
               lambda: ops.stage_inputs(items, flows, tprev, 2 * s, s, xin, fold=True))
    tprev = torch.randn(n, 13, H // 2, W // 2, generator=g).to(dev)
    conv00f = ops.Conv3x3(torch.randn(16, 52, 3, 3, generator=g) * 0.05, torch.zeros(16), 2, True, None, device=dev)
    target(f"stage input s=1 + conv0[0] fused, folded flow update, {n} samples",
           lambda: ops.stage_conv0(items, flows, tprev, 2.0, conv00f, fold=True))
    tprev = torch.randn(n, 13, H // 8, W // 8, generator=g).to(dev)
    xin4 = torch.empty(n, 52, H // 4, W // 4, device=dev)
    target(f"stage input s=4, {n} samples", lambda: ops.stage_inputs(items, flows, tprev, 8.0, 4.0, xin4))

    # what the pipeline launches since the running flow is a list of terms (head outputs of the earlier stages: a refinement
    # of a few pixels per stage on top of the coarse flow) -- the launch labels end in "+lazy"
    def head(st, amp):  # a head output whose flow channels are SMOOTH (low-resolution noise, bicubic), as a trained stage's are
        hh, ww = int(H / st), int(W / st)
        t = torch.randn(n, 13, hh, ww, generator=g)
        lo = torch.randn(n, 4, max(hh // 8, 2), max(ww // 8, 2), generator=g) * amp
        t[:, :4] = torch.nn.functional.interpolate(lo, size=(hh, ww), mode="bicubic", align_corners=False)
        return t.to(dev)
    pyr = {16.0: head(16.0, 1.0), 8.0: head(8.0, 0.3), 4.0: head(4.0, 0.3), 2.0: head(2.0, 0.3)}
    target(f"stage input s=1 + conv0[0] fused, flow as 3 terms, {n} samples",
           lambda: ops.stage_conv0(items, None, pyr[2.0], 2.0, conv00f, terms=[(pyr[16.0], 16.0), (pyr[8.0], 8.0), (pyr[4.0], 4.0)]))
    xl2 = torch.empty(n, 52, H // 2, W // 2, device=dev)
    target(f"stage input s=2, flow as 2 terms, {n} samples",
           lambda: ops.stage_inputs(items, None, pyr[4.0], 4.0, 2.0, xl2, terms=[(pyr[16.0], 16.0), (pyr[8.0], 8.0)]))
    target(f"stage input s=4, flow as 1 term, {n} samples",
           lambda: ops.stage_inputs(items, None, pyr[8.0], 8.0, 4.0, xin4, terms=[(pyr[16.0], 16.0)]))
    # the scale-2 stage fused with block 3's conv0[0] (52 -> 32, stride 2), what the pipeline launches since round 5
    conv0s2 = ops.Conv3x3(torch.randn(32, 52, 3, 3, generator=g) * 0.05, torch.zeros(32), 2, True, None, device=dev)
    target(f"stage input s=2 + conv0[0] fused, flow as 2 terms, {n} samples",
           lambda: ops.stage_conv0(items, None, pyr[4.0], 4.0, conv0s2, terms=[(pyr[16.0], 16.0), (pyr[8.0], 8.0)], scale=2))
    del xl2, pyr
    for (c, h, w) in ((64, 136, 240), (32, 272, 480), (96, 68, 120), (128, 34, 60), (192, 17, 30)) + (((32, 544, 960),) if n == 2 else ()):
        x = torch.randn(n, c, h, w, generator=g).to(dev)
        layer = ops.Conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.05, torch.zeros(c), 1, True, torch.ones(1, c, 1, 1), device=dev)
        out = torch.empty_like(x)
        target(f"ResConv {c}->{c}ch {h}x{w} N{n}", lambda: layer(x, residual=x, out=out))
    x52 = torch.randn(n, 52, H, W, generator=g).to(dev)
    conv00 = ops.Conv3x3(torch.randn(16, 52, 3, 3, generator=g) * 0.05, torch.zeros(16), 2, True, None, device=dev)
    target(f"block4 conv0.0 52->16 s2 1088x1920 N{n}", lambda: conv00(x52))
    x32 = torch.randn(n, 32, 272, 480, generator=g).to(dev)
    last = ops.Deconv4x4(torch.randn(32, 52, 4, 4, generator=g) * 0.05, torch.zeros(52), pixel_shuffle=True, device=dev)
    target(f"block4 lastconv 32->52 deconv + PixelShuffle 272x480 N{n}", lambda: last(x32))
    del x52, xin, x32
# the context encoder as the pipeline runs it (pair-interleaved features only), and the final synthesis of a group of 4 steps
from drba_amd.models.rife_426_heavy.IFNet_HDv3 import Head  # noqa: E402
hsd = {"encode.cnn0.weight": torch.randn(16, 3, 3, 3, generator=g) / 27 ** 0.5, "encode.cnn0.bias": torch.zeros(16),
       "encode.cnn1.weight": torch.randn(16, 16, 3, 3, generator=g) / 12, "encode.cnn1.bias": torch.zeros(16),
       "encode.cnn2.weight": torch.randn(16, 16, 3, 3, generator=g) / 12, "encode.cnn2.bias": torch.zeros(16),
       "encode.cnn3.weight": torch.randn(16, 16, 4, 4, generator=g) / 8, "encode.cnn3.bias": torch.zeros(16)}
head_net = Head(hsd, "encode.", dev)
target("head_fused 1088x1920, pair layout only", lambda: head_net(img0, planar=False))


def head8(st, amp):
    hh, ww = int(H / st), int(W / st)
    t = torch.randn(8, 13, hh, ww, generator=g)
    lo = torch.randn(8, 4, max(hh // 8, 2), max(ww // 8, 2), generator=g) * amp
    t[:, :4] = torch.nn.functional.interpolate(lo, size=(hh, ww), mode="bicubic", align_corners=False)
    return t.to(dev)


wb_terms = [(head8(16.0, 1.0), 16.0), (head8(8.0, 0.3), 8.0), (head8(4.0, 0.3), 4.0), (head8(2.0, 0.3), 2.0)]
wb_last = head8(1.0, 0.3)[:, :5].contiguous() if False else head8(1.0, 0.3)
target("warp_blend_lazy, 4 terms, 8 samples", lambda: ops.warp_blend_lazy([(it[0], it[1]) for it in items8], wb_terms, wb_last, 1.0))
del wb_terms, wb_last
# GMFSS_UNION's matrix-core kernels at 1080p (1152x1920 -> 576x960 working resolution, GMFlow at 1/8: 72x120 = 8640 tokens,
# fine scale 144x240 = 34560 tokens x 2 directions): fused window attention, the MLP's 256 -> 1024 linear, GridNet's
# full-resolution 32-channel layer
if os.environ.get("DRBA_PMC_GMFSS", "1") == "1":
    qkv = torch.randn(2, 144 * 240, 384, generator=g).to(dev)
    target("GMFlow window attention, fine scale (2 x 34560 tokens, 8x8 windows, C = 128)",
           lambda: ops.window_attention(qkv[..., 0:128], qkv[..., 128:256], qkv[..., 256:384], 144, 240, 8, True, 128 ** -0.5), pick=0)
    lin = ops.LinearSplit(torch.randn(1024, 256, generator=g) * 0.05, torch.zeros(1024), gelu=True, device=dev)
    xt = torch.randn(2 * 144 * 240, 256, generator=g).to(dev)
    target("transformer MLP 256 -> 1024 + GELU, 69120 tokens", lambda: lin(xt))
    xg = torch.randn(1, 32, 1152, 1920, generator=g).to(dev)
    grid = ops.Conv3x3(torch.randn(32, 32, 3, 3, generator=g) * 0.05, torch.zeros(32), 1, "prelu", None, device=dev, pre_slope=0.25, post_slope=0.25)
    og = torch.empty_like(xg)
    target("GridNet 32 -> 32 at 1152x1920 (PReLU pre-activation)", lambda: grid(xg, out=og))
torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(manifest, open(os.path.join(ROOT, "gpurun_out", "pmc_manifest.json"), "w"), indent=1)
print(json.dumps(manifest, indent=1))
