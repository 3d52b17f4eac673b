#!/usr/bin/env python3
"""Experiment: the full-resolution stage's input kernel fused with conv0[0] (stage_head.hip next to this file).
    python tools/exp/stage_head/run.py [--clocks] [H W]
Builds the kernel into a private copy of libdrba_hip.so (/tmp), checks it against the product's unfused pair
(drba_ifblock_input_lds + drba_conv3x3) on the same inputs, and times both per sample on an otherwise idle GPU, for a
smooth flow (taps inside the LDS-staged windows: fast path) and a rough one (slow path: per-lane gathers).
--clocks: experiment build with -DDRBA_SH_CLOCKS, clocks between the kernel's barriers."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
argv = [a for a in sys.argv[1:] if a != "--clocks"]
clocks = "--clocks" in sys.argv
csrc = os.path.join(ROOT, "drba_amd", "csrc")
exp = "/tmp/drba_sh_exp"
os.makedirs(exp, exist_ok=True)
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics"] + (["-DDRBA_SH_CLOCKS"] if clocks else [])
subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(HERE, "stage_head.hip"), "-o", os.path.join(exp, "stage_head.o")], check=True)
objs = [os.path.join(csrc, o) for o in sorted(os.listdir(csrc)) if o.endswith(".o")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(exp, "stage_head.o")] + objs +
               ["-o", os.path.join(exp, "libdrba_hip.so")], check=True)
from drba_amd import _lib, ops  # noqa: E402
_lib.LIB_PATH = os.path.join(exp, "libdrba_hip.so")  # the private build, for this process only
L = _lib.load()
_p, _i, _f = C.c_void_p, C.c_int, C.c_float
L.drba_stage_head_packed_floats.restype, L.drba_stage_head_packed_floats.argtypes = C.c_size_t, [_i]
L.drba_stage_head_pack.restype, L.drba_stage_head_pack.argtypes = _i, [_p, _p, _i]
L.drba_stage_head.restype = _i
L.drba_stage_head.argtypes = [_p, _p, _p, _p, _p, _f, _p, _p, _i, _i, _f, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]

dev = torch.device("cuda:0")
H, W = (int(argv[0]), int(argv[1])) if len(argv) > 1 else (1088, 1920)
g = torch.Generator().manual_seed(0)
img0, img1 = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 3, H, W, generator=g).to(dev)
f0, f1 = torch.randn(1, 16, H, W, generator=g).to(dev), torch.randn(1, 16, H, W, generator=g).to(dev)
tmap = torch.rand(1, 1, H, W, generator=g).to(dev)
tprev = (torch.randn(1, 13, H // 2, W // 2, generator=g) * 0.05).to(dev)
f0p, f1p = ops.pair_interleaved(f0), ops.pair_interleaved(f1)
conv = ops.Conv3x3(torch.randn(16, 52, 3, 3, generator=g) * 0.05, torch.randn(16, generator=g) * 0.1, 2, True, None, device=dev)
packed = torch.empty(L.drba_stage_head_packed_floats(16), dtype=torch.float32)
assert L.drba_stage_head_pack(conv.w_host.data_ptr(), packed.data_ptr(), 16) == 0
packed = packed.to(dev)
ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1


def stage_head(flow, fold):
    out = torch.empty(1, 16, ho, wo, device=dev)
    fo = torch.empty(1, 4, H, W, device=dev) if fold else None
    rc = L.drba_stage_head(img0.data_ptr(), img1.data_ptr(), f0p.data_ptr(), f1p.data_ptr(), tmap.data_ptr(), 0.0, flow.data_ptr(),
                           tprev.data_ptr(), H // 2, W // 2, 2.0, None if fo is None else fo.data_ptr(), packed.data_ptr(),
                           conv.bias.data_ptr(), out.data_ptr(), H, W, H, W, 16, 1.0, ops._stream())
    assert rc == 0, rc
    return out, fo


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, amp in (("smooth (varies by < 1 px inside a tile, the bench workload's regime: LDS-staged windows)", 0.5),
                  ("rough (varies by several px inside a tile: slow path, per-lane gathers)", 6.0)):
    flow = torch.nn.functional.interpolate(torch.randn(1, 4, H // 32, W // 32, generator=g) * amp, size=(H, W), mode="bilinear").to(dev).contiguous()
    xin = torch.empty(1, 52, H, W, device=dev)
    _, fo_ref = ops.ifblock_input_lds(img0, img1, f0, f1, tmap, flow, tprev, 2.0, 1.0, out=xin, fold=True)
    want = conv(xin)
    got, fo = stage_head(flow, True)
    torch.cuda.synchronize()
    err = float((got - want).abs().max()) / max(1.0, float(want.abs().max()))
    print(f"flow: {name}\n  fused vs pair: max rel diff {err:.2e}, folded flow max diff {float((fo - fo_ref).abs().max()):.2e}")
    t_in = timed(lambda: ops.ifblock_input_lds(img0, img1, f0, f1, tmap, flow, tprev, 2.0, 1.0, out=xin, fold=True))
    t_conv = timed(lambda: conv(xin))
    t_f = timed(lambda: stage_head(flow, True))
    t_nf = timed(lambda: stage_head(flow, False))
    print(f"  {H}x{W}: input+fold {t_in:.1f} us + conv0[0] {t_conv:.1f} us = {t_in + t_conv:.1f} us | fused+fold {t_f:.1f} us, fused without the fold {t_nf:.1f} us")
    if clocks:
        L.drba_stage_head_clocks.argtypes = [C.c_void_p, C.c_int]
        L.drba_stage_head_clocks(None, 1)
        for _ in range(10):
            stage_head(flow, True)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 32)()
        L.drba_stage_head_clocks(buf, 0)
        tot = sum(buf)
        print("  share of a workgroup's clocks between marks (odd: work before barrier k = (i + 1) / 2, even: the barrier's wait):")
        print("   " + " ".join(f"[{i}] {100.0 * buf[i] / tot:.1f}%" for i in range(1, 30)))
