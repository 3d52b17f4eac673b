#!/usr/bin/env python3
"""stage_conv16.hip (scale-1 stage input fused with conv0[0], two fp16 terms per operand) against
  * an fp64 convolution of the unfused stage input (ops.stage_inputs(scale=1)) -- the bound of the split family, 5e-6 max|y|,
  * stage_conv.hip (exact fp32 products) -- and the folded flows of both against ifblock_input_lds', bit for bit,
on ragged sizes, with / without the fold, the flow as terms; then the launch time of both forms at
1080p (8 items, the flow as three terms: the loop's launch).    python tools/stage_conv16_check.py [reps] [--no-time | --time-only]"""
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


def heads(B, H, W, amp16=1.0):
    def head(st, amp):
        t = torch.randn(B, 13, max(H // st, 1), max(W // st, 1), generator=g)
        lo = torch.randn(B, 4, max(H // st // 8, 2), max(W // st // 8, 2), generator=g) * amp
        t[:, :4] = torch.nn.functional.interpolate(lo, size=t.shape[2:], mode="bicubic", align_corners=False)
        return t.to(dev)
    return [(head(16, amp16), 16.0), (head(8, 0.4), 8.0), (head(4, 0.3), 4.0)], head(2, 0.3)


def conv64(xin, w, b):
    y = torch.nn.functional.conv2d(xin.double(), w.double().to(dev), b.double().to(dev), stride=2, padding=1)
    return torch.nn.functional.leaky_relu(y, 0.2)


def run(conv, two, *a, **k):
    ops.STAGE_CONV_TWO_TERM = two
    try:
        return ops.stage_conv0(*a, **k)
    finally:
        ops.STAGE_CONV_TWO_TERM = None


for cout in (() if "--time-only" in sys.argv else (16,)):
    wt = torch.randn(cout, 52, 3, 3, generator=g) / (52 * 9) ** 0.5
    bs = torch.randn(cout, generator=g) * 0.1
    conv = ops.Conv3x3(wt, bs, 2, True, None, device=dev)
    cases = [(1, 64, 128, True, True, True), (2, 64, 128, False, True, False), (1, 70, 90, True, True, True), (2, 35, 67, True, False, True),
             (1, 33, 34, False, True, True), (3, 96, 160, True, True, True), (1, 256, 448, True, True, False), (2, 1088, 1920, True, True, True)]
    for (B, H, W, tmap, with_flow, fold) in cases:
        items, flows, tprev = make(B, H, W, tmap, with_flow)
        xin = torch.empty(B, 52, H, W, device=dev)
        fl_ref = ops.stage_inputs(items, flows, tprev, 2.0, 1.0, xin, fold=fold)
        ref = conv64(xin, wt, bs)
        y, fl = run(conv, True, items, flows, tprev, 2.0, conv, fold=fold)
        torch.cuda.synchronize()
        mag = float(ref.abs().max())
        err = float((y.double() - ref).abs().max())
        e32 = float("nan")
        if cout == 16:
            y32, _ = run(conv, False, items, flows, tprev, 2.0, conv, fold=fold)
            e32 = float((y32.double() - ref).abs().max())
        ferr = max(float((a - b).abs().max()) for a, b in zip(fl, fl_ref)) if fold else 0.0
        ok = err <= 5e-6 * max(1.0, mag) and ferr == 0.0 and bool(torch.isfinite(y).all())
        bad += not ok
        print(f"cout {cout} B{B} {H}x{W} tmap={int(tmap)} flow={int(with_flow)} fold={int(fold)}: two-term {err:.2e} / fp32 form {e32:.2e} vs fp64 "
              f"(|y| <= {mag:.2f}), flow err {ferr:.1e} {'ok' if ok else 'FAIL'}", flush=True)
    for (B, H, W) in ((2, 128, 256), (3, 96, 160), (1, 70, 90), (2, 1088, 1920)):
        items, _, _ = make(B, H, W, True, False)
        for amp, tag in ((1.0, "smooth flows"), (30.0, "rough flows")):
            terms, tprev = heads(B, H, W, amp)
            xin = torch.empty(B, 52, H, W, device=dev)
            ops.stage_inputs(items, None, tprev, 2.0, 1.0, xin, terms=terms)
            ref = conv64(xin, wt, bs)
            y, _ = run(conv, True, items, None, tprev, 2.0, conv, terms=terms)
            torch.cuda.synchronize()
            mag = float(ref.abs().max())
            err = float((y.double() - ref).abs().max())
            ok = err <= 5e-6 * max(1.0, mag) and bool(torch.isfinite(y).all())
            bad += not ok
            print(f"cout {cout} lazy B{B} {H}x{W} {tag}: {err:.2e} vs fp64 (|y| <= {mag:.2f}) {'ok' if ok else 'FAIL'}", flush=True)
# ---- scale 2 (stage_conv16_s2): the stage input at half resolution (the mean of 2 x 2 warped sample points) fused with conv0[0];
# reference = the unfused gather (ops.stage_inputs at scale 2, flow as terms) + an fp64 convolution; 16 and 32 output channels
def heads2(B, H, W, amp=1.0):
    def head(st, a):
        t = torch.randn(B, 13, max(H // st, 1), max(W // st, 1), generator=g)
        lo = torch.randn(B, 4, max(H // st // 8, 2), max(W // st // 8, 2), generator=g) * a
        t[:, :4] = torch.nn.functional.interpolate(lo, size=t.shape[2:], mode="bicubic", align_corners=False)
        return t.to(dev)
    return [(head(16, amp), 16.0), (head(8, 0.4), 8.0)], head(4, 0.3)


for cout in (() if "--time-only" in sys.argv else (16, 32)):
    wt = torch.randn(cout, 52, 3, 3, generator=g) / (52 * 9) ** 0.5
    bs = torch.randn(cout, generator=g) * 0.1
    conv = ops.Conv3x3(wt, bs, 2, True, None, device=dev)
    for (B, H, W) in ((2, 128, 256), (3, 96, 160), (1, 72, 104), (1, 64, 64), (2, 1088, 1920)):
        items, _, _ = make(B, H, W, True, False)
        for it in items:
            ops.rgbx(it[0]), ops.rgbx(it[1])
        for amp, tag in ((1.0, "smooth flows"), (20.0, "rough flows")):
            terms, tprev = heads2(B, H, W, amp)
            xin = torch.empty(B, 52, H // 2, W // 2, device=dev)
            ops.stage_inputs(items, None, tprev, 4.0, 2.0, xin, terms=terms)
            ref = conv64(xin, wt, bs)
            assert ops.stage_conv0_ok(conv, H, W, 2.0, 4.0, items=items)
            y, _ = run(conv, True, items, None, tprev, 4.0, conv, terms=terms, scale=2)
            torch.cuda.synchronize()
            mag = float(ref.abs().max())
            err = float((y.double() - ref).abs().max())
            ok = err <= 5e-6 * max(1.0, mag) and bool(torch.isfinite(y).all()) and tuple(y.shape) == tuple(ref.shape)
            bad += not ok
            print(f"scale 2 cout {cout} B{B} {H}x{W} {tag}: {err:.2e} vs fp64 (|y| <= {mag:.2f}) {'ok' if ok else 'FAIL'}", flush=True)
print("FAILED" if bad else "all ok", flush=True)

if "--no-time" not in sys.argv and "--scale1-only" not in sys.argv:
    for (H, W, cout, tag) in ((1088, 1920, 32, "1080p block 3"), (2176, 3840, 16, "4K scale 0.5 block 4")):
        B = 8
        items, _, _ = make(B, H, W, True, False)
        fr = [(items[j][0], items[j][3]) for j in range(B // 2 + 2)]   # six frames for eight items, as the pipeline's groups
        items = [it for j in range(B // 2) for it in ((fr[j + 1][0], fr[j][0], items[2 * j][2], fr[j + 1][1], fr[j][1]),
                                                      (fr[j + 1][0], fr[j + 2][0], items[2 * j + 1][2], fr[j + 1][1], fr[j + 2][1]))]
        for im, _ in fr:
            ops.rgbx(im)
        wt = torch.randn(cout, 52, 3, 3, generator=g) / (52 * 9) ** 0.5
        conv = ops.Conv3x3(wt, torch.zeros(cout), 2, True, None, device=dev)
        terms, tprev = heads2(B, H, W, 1.0)
        xin = torch.empty(B, 52, H // 2, W // 2, device=dev)

        def unfused():
            ops.stage_inputs(items, None, tprev, 4.0, 2.0, xin, terms=terms)
            return conv(xin)

        def fused():
            return run(conv, True, items, None, tprev, 4.0, conv, terms=terms, scale=2)

        for name, fn in (("gather + conv0[0]", unfused), ("fused (stage_conv16_s2)", fused)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ops.trace_begin()
            for _ in range(reps):
                fn()
            recs = ops.trace_end()
            us = sum(r["ms"] for r in recs) / reps * 1e3
            print(f"scale 2, {tag} ({H}x{W}, cout {cout}, B{B}, shared frames): {name}: {us:8.1f} us per call ({', '.join(sorted({r['name'].split('<')[0].split('::')[-1] for r in recs}))})", flush=True)

if "--no-time" not in sys.argv and "--scale2-only" not in sys.argv:
    H, W, B = 1088, 1920, 8
    items, _, _ = make(B, H, W, True, False)
    if "--planar-frames" not in sys.argv:  # the pipeline's frames carry their [H,W,4] copies (ops.to_inp): the launch the loop makes
        for it in items:
            ops.rgbx(it[0]), ops.rgbx(it[1])
    if "--same-frames" in sys.argv:   # every item reads the same two frames: 7 of 8 items find their sources in L2
        items = [items[0]] * B
    elif "--shared-frames" in sys.argv:  # the items of a group share their frames as the pipeline's do (6 frames for 8 items)
        fr = [(items[j][0], items[j][3]) for j in range(B // 2 + 2)]
        items = [it for j in range(B // 2) for it in ((fr[j + 1][0], fr[j][0], items[2 * j][2], fr[j + 1][1], fr[j][1]),
                                                      (fr[j + 1][0], fr[j + 2][0], items[2 * j + 1][2], fr[j + 1][1], fr[j + 2][1]))]
    for cout in (16,):
        wt = torch.randn(cout, 52, 3, 3, generator=g) / (52 * 9) ** 0.5
        conv = ops.Conv3x3(wt, torch.zeros(cout), 2, True, None, device=dev)
        for amp, tag in ((0.0, "zero"), (1.0, "smooth"), (30.0, "rough")):
            terms, tprev = heads(B, H, W, max(amp, 1.0))
            if amp == 0.0:
                terms = [(t * 0.0, sc) for t, sc in terms]
                tprev = tprev.clone()
                tprev[:, :4] = 0.0
            for two in ((True, False) if cout == 16 else (True,)):
                for _ in range(3):
                    run(conv, two, items, None, tprev, 2.0, conv, terms=terms)
                torch.cuda.synchronize()
                ops.trace_begin()
                for _ in range(reps):
                    run(conv, two, items, None, tprev, 2.0, conv, terms=terms)
                recs = [r for r in ops.trace_end() if "stage_conv" in r["name"]]
                us = sum(r["ms"] for r in recs) / len(recs) * 1e3
                alg = B * 4.0 * (39.0 * H * W + cout * (H // 2) * (W // 2))
                print(f"1080p B{B} cout {cout} lazy {tag} flows, {'two-term fp16' if two else 'fp32 MFMA    '}: {us:8.1f} us per launch = {us / B:6.1f} us per sample "
                      f"({alg / us / 1e3:7.1f} GB/s algorithmic, {alg / us / 1e3 / 8000:.3f} of HBM)", flush=True)
sys.exit(1 if bad else 0)
