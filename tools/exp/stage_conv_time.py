#!/usr/bin/env python3
"""Times drba_stage_conv0_batch of every library under tools/exp/build/ (experiment builds of stage_conv.hip, see
stage_conv_variants.sh) at 1080p, 2 and 8 samples per launch, one subprocess per library; kernel time from the library's own
launch trace.    python tools/exp/stage_conv_time.py [reps]"""
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
dev = torch.device("cuda:0")
reps = int(sys.argv[2])
g = torch.Generator().manual_seed(0)
H, W = 1088, 1920
conv = ops.Conv3x3(torch.randn(16, 52, 3, 3, generator=g) * 0.05, torch.zeros(16), 2, True, None, device=dev)
res = []
for B in (2, 8):
    items, flows = [], []
    for _ in range(B):
        i0, i1 = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 3, H, W, generator=g).to(dev)
        f0, f1 = torch.randn(1, 16, H, W, generator=g).to(dev), torch.randn(1, 16, H, W, generator=g).to(dev)
        items.append((i0, i1, torch.rand(1, 1, H, W, generator=g).to(dev), f0, f1))
        lo = torch.randn(1, 4, H // 16, W // 16, generator=g) * 5
        flows.append(torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear").to(dev).contiguous())
    tprev = torch.randn(B, 13, H // 2, W // 2, generator=g).to(dev)
    terms = [(torch.randn(B, 13, H // st, W // st, generator=g).to(dev), float(st)) for st in (16, 8, 4)]  # the 1080p pyramid
    for _ in range(3):
        ops.stage_conv0(items, None, tprev, 2.0, conv, terms=terms)
    torch.cuda.synchronize()
    ops.trace_begin()
    for _ in range(reps):
        ops.stage_conv0(items, None, tprev, 2.0, conv, terms=terms)
    recs = [r for r in ops.trace_end() if "stage_conv0" in r["name"]]
    us = sum(r["ms"] for r in recs) / len(recs) * 1e3
    res.append("B%%d %%7.1f us (%%5.1f us/sample)" %% (B, us, us / B))
print("  ".join(res))
''' % ROOT
reps = sys.argv[1] if len(sys.argv) > 1 else "10"
combos = [(t, r) for t in (8, 6, 4) for r in (1, 0)]  # DRBA_SC_TOH x DRBA_SC_WRES (read by libraries built with TUNING switches)
for lib in sorted(glob.glob(os.path.join(ROOT, "tools", "exp", "build", "libdrba_hip_*.so"))):
    for toh, wres in combos:
        env = dict(os.environ, DRBA_SC_TOH=str(toh), DRBA_SC_WRES=str(wres))
        out = subprocess.run([sys.executable, "-c", CHILD, lib, reps], capture_output=True, text=True, env=env)
        print(f"{os.path.basename(lib)[12:-3]:24s} TOH={toh} WRES={wres}  {out.stdout.strip() or out.stderr.strip()[-300:]}", flush=True)
