#!/usr/bin/env python3
"""stage_conv.hip (scale-1 stage input fused with conv0[0]) against the unfused pair it replaces
(ops.stage_inputs(scale=1) + ops.Conv3x3(stride 2)) on the same inputs: with / without the folded flow update, timestep
map / scalar, ragged sizes (tiles cut by the border, odd H / W), several items per launch; then timing of both paths at
1080p for 2 and 8 samples.    python tools/stage_conv_check.py [reps] [--no-time]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
g = torch.Generator().manual_seed(5)
bad = 0


def make(B, H, W, tmap, with_flow):
    items, flows = [], []
    for _ in range(B):
        i0, i1 = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 3, H, W, generator=g).to(dev)
        f0, f1 = torch.randn(1, 16, H, W, generator=g).to(dev), torch.randn(1, 16, H, W, generator=g).to(dev)
        t = torch.rand(1, 1, H, W, generator=g).to(dev) if tmap else 0.37
        items.append((i0, i1, t, f0, f1))
        lo = torch.randn(1, 4, max(H // 16, 2), max(W // 16, 2), generator=g) * 5
        flows.append(torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear").to(dev).contiguous() if with_flow else None)
    tprev = torch.randn(B, 13, H // 2, W // 2, generator=g).to(dev)
    return items, flows, tprev


def both(conv, items, flows, tprev, fold):
    B = len(items)
    _, _, H, W = items[0][0].shape
    xin = torch.empty(B, 52, H, W, device=dev)
    fl_ref = ops.stage_inputs(items, flows, tprev, 2.0, 1.0, xin, fold=fold)
    y_ref = conv(xin)
    y, fl = ops.stage_conv0(items, flows, tprev, 2.0, conv, fold=fold)
    torch.cuda.synchronize()
    return y_ref, fl_ref, y, fl


wt = torch.randn(16, 52, 3, 3, generator=g) / (52 * 9) ** 0.5
bs = torch.randn(16, generator=g) * 0.1
conv = ops.Conv3x3(wt, bs, 2, True, None, device=dev)
cases = [(1, 64, 128, True, True, True), (2, 64, 128, False, True, False), (1, 70, 90, True, True, True), (2, 35, 67, True, False, True),
         (1, 33, 34, False, True, True), (3, 96, 160, True, True, True), (1, 256, 448, True, True, False), (2, 1088, 1920, True, True, True)]
for (B, H, W, tmap, with_flow, fold) in cases:
    if not fold and not with_flow:
        continue
    items, flows, tprev = make(B, H, W, tmap, with_flow)
    assert ops.stage_conv0_ok(conv, H, W, 1.0, 2.0)
    y_ref, fl_ref, y, fl = both(conv, items, flows, tprev, fold)
    err = float((y - y_ref).abs().max())
    scale = float(y_ref.abs().max())
    ferr = 0.0
    if fold:
        ferr = max(float((a - b).abs().max()) for a, b in zip(fl, fl_ref))
    ok = err <= 2e-5 * max(1.0, scale) and ferr == 0.0 and bool(torch.isfinite(y).all())
    bad += not ok
    print(f"B{B} {H}x{W} tmap={int(tmap)} flow={int(with_flow)} fold={int(fold)}: max err {err:.3e} (|y| <= {scale:.2f}), flow err {ferr:.1e} "
          f"{'ok' if ok else 'FAIL'}", flush=True)
print("FAILED" if bad else "all ok", flush=True)

# ---- the two paths of the kernel: source boxes in LDS (16-byte aligned frames, smooth flows) and per-lane gathers (frames
# whose planes are not 16-byte aligned run it on every tile; so do tiles whose taps do not fit the boxes: the random flows)
def misaligned(t):
    buf = torch.empty(t.numel() + 1, device=t.device, dtype=t.dtype)
    v = buf[1:].view(t.shape)
    v.copy_(t)
    return v


def smooth_heads(B, H, W, amp16=1.0):
    def head(st, amp):
        t = torch.randn(B, 13, H // st, W // st, generator=g)
        lo = torch.randn(B, 4, max(H // st // 8, 2), max(W // st // 8, 2), generator=g) * amp
        t[:, :4] = torch.nn.functional.interpolate(lo, size=(H // st, W // st), mode="bicubic", align_corners=False)
        return t.to(dev)
    return [(head(16, amp16), 16.0), (head(8, 0.4), 8.0), (head(4, 0.3), 4.0)], head(2, 0.3)


for (B, H, W) in ((2, 128, 256), (3, 96, 160), (2, 1088, 1920)):
    items, _, _ = make(B, H, W, True, False)
    for amp, tag in ((1.0, "smooth flows"), (30.0, "rough flows")):
        terms, tprev = smooth_heads(B, H, W, amp)
        y_box, _ = ops.stage_conv0(items, None, tprev, 2.0, conv, terms=terms)
        items_g = [(misaligned(a), b, c, d, e) for a, b, c, d, e in items]
        y_gat, _ = ops.stage_conv0(items_g, None, tprev, 2.0, conv, terms=terms)
        xin = torch.empty(B, 52, H, W, device=dev)
        ops.stage_inputs(items, None, tprev, 2.0, 1.0, xin, terms=terms)
        y_ref = conv(xin)
        torch.cuda.synchronize()
        e1, e2 = float((y_box - y_ref).abs().max()), float((y_gat - y_ref).abs().max())
        ok = max(e1, e2) <= 2e-5 * max(1.0, float(y_ref.abs().max())) and bool(torch.isfinite(y_box).all())
        bad += not ok
        print(f"lazy B{B} {H}x{W} {tag}: aligned frames (box path where the taps fit) err {e1:.3e}, misaligned frames (gather path) err {e2:.3e} "
              f"{'ok' if ok else 'FAIL'}", flush=True)
print("FAILED" if bad else "all ok (lazy / box vs gather)", flush=True)

if "--no-time" not in sys.argv:
    H, W = 1088, 1920
    for B in (8,):
        items, _, _ = make(B, H, W, True, False)
        items_g = [(misaligned(a), b, c, d, e) for a, b, c, d, e in items]
        for amp, tag in ((0.0, "zero"), (0.1, "gentle"), (1.0, "smooth"), (30.0, "rough")):
            terms, tprev = smooth_heads(B, H, W, amp)
            if amp <= 0.1:  # flows that certainly fit the boxes: every tile of the aligned run takes the box path
                terms = [(t * (1.0 if amp else 0.0), sc) for t, sc in terms]
                tprev = tprev.clone()
                tprev[:, :4] *= amp
                if amp:
                    terms = [(torch.cat((t[:, :4] * 0.1, t[:, 4:]), 1).contiguous(), sc) for t, sc in terms]
            for name, its in (("box path   ", items), ("gather path", items_g)):
                for _ in range(3):
                    ops.stage_conv0(its, None, tprev, 2.0, conv, terms=terms)
                torch.cuda.synchronize()
                ops.trace_begin()
                for _ in range(reps):
                    ops.stage_conv0(its, None, tprev, 2.0, conv, terms=terms)
                recs = [r for r in ops.trace_end() if "stage_conv0" in r["name"]]
                us = sum(r["ms"] for r in recs) / len(recs) * 1e3
                alg = B * 4.0 * (39.0 * H * W + 16 * (H // 2) * (W // 2))
                print(f"1080p B{B} lazy {tag} flows, {name}: {us:8.1f} us per launch = {us / B:6.1f} us per sample ({alg / us / 1e3:7.1f} GB/s algorithmic, "
                      f"{alg / us / 1e3 / 8000:.3f} of HBM)", flush=True)
if "--no-time" not in sys.argv:
    H, W = 1088, 1920
    for B in (2, 8):
        items, flows, tprev = make(B, H, W, True, True)
        xin = torch.empty(B, 52, H, W, device=dev)

        def unfused():
            ops.stage_inputs(items, flows, tprev, 2.0, 1.0, xin, fold=True)
            return conv(xin)

        def fused():
            return ops.stage_conv0(items, flows, tprev, 2.0, conv, fold=True)

        for name, fn in (("unfused", unfused), ("fused", fused)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            alg = B * 4.0 * (43.0 * H * W + 16 * (H // 2) * (W // 2) + 4.0 * H * W)
            print(f"1080p B{B} {name}: {us:8.1f} us per call  ({alg / us / 1e3:7.1f} GB/s of the fused kernel's algorithmic bytes)", flush=True)
sys.exit(1 if bad else 0)
