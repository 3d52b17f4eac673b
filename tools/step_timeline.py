#!/usr/bin/env python3
"""Timeline of traced steady-state bench steps from the library's own kernel trace (event pairs on the dispatch packets):
per stream, every launch with its start, duration and the idle gap before it.  Traced steps run slower than plain ones
(profiled dispatches), so read the structure -- which stream waits for which -- not the absolute step time.
    python tools/step_timeline.py [--steps 2] [--config 1080p] [--brief]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from drba_amd.models.rife import RIFE  # noqa: E402
from drba_amd.utils import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--config", default="1080p")
ap.add_argument("--brief", action="store_true")
ap.add_argument("--one-stream", action="store_true", help="side / prefetch work on the main stream: every duration is the kernel's own")
ap.add_argument("--table", action="store_true", help="per (symbol, launch label) totals instead of the timeline")
ap.add_argument("--no-img-x4", action="store_true", help="A/B: the gathers read the planar frames (ops.IMG_X4 = False)")
a = ap.parse_args()
if a.no_img_x4:
    from drba_amd import ops as _ops
    _ops.IMG_X4 = False
if a.one_stream:
    from drba_amd.models import lookahead as _la
    _la.ONE_STREAM = True
(H, W), scale, _ = bench.CONFIGS[a.config]
dev = torch.device("cuda:0")
m = RIFE(weights=synth.ifnet_state_dict(0), scale=scale, device=dev)
clip = bench.DeviceClip(12, H, W, 1234, dev)
frames = [clip[k] for k in range(len(clip))]
# bench.py's own loop (groups of RIFE.GROUP steps, three streams), its instrumented steps traced launch by launch
args = argparse.Namespace(steps=8, warmup=3, no_lookahead=False)
_, _, recs, traced, _ = bench.step_loop(m, frames, 8, args, 1, trace=True)
a.steps = traced
t0 = min(r["start_ms"] for r in recs)
for r in recs:
    r["start_ms"] -= t0
streams = sorted({r["stream"] for r in recs})
names = {s: f"s{i}" for i, s in enumerate(streams)}
t_end = max(r["start_ms"] + r["ms"] for r in recs)
print(f"{a.steps} traced steps: {len(recs)} launches, span {t_end:.3f} ms = {t_end / a.steps:.3f} ms/step (traced)")
if a.table:
    agg = {}
    for r in recs:
        k = (r["name"].replace("drba_conv_split::", "").replace("drba_conv::", "")[:70], r["label"] or "")
        v = agg.setdefault(k, [0, 0.0])
        v[0] += 1
        v[1] += r["ms"]
    tot = sum(v[1] for v in agg.values())
    print(f"{a.steps} traced steps, {len(recs)} launches, kernel time {tot / a.steps:.3f} ms per step, span {t_end / a.steps:.3f} ms per step")
    for (nm, lab), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{ms / a.steps * 1e3:8.1f} us/step {100 * ms / tot:5.1f} %  {n / a.steps:5.2f} x {ms / n * 1e3:8.1f} us  {nm} {lab}")
    sys.exit(0)
busy = {s: sum(r["ms"] for r in recs if r["stream"] == s) for s in streams}
for s in streams:
    print(f"  stream {names[s]}: {sum(1 for r in recs if r['stream'] == s)} launches, {busy[s]:.3f} ms of kernels")
last = {}
for r in sorted(recs, key=lambda r: r["start_ms"]):
    s = r["stream"]
    gap = r["start_ms"] - last.get(s, r["start_ms"])
    last[s] = r["start_ms"] + r["ms"]
    nm = r["name"].replace("drba_conv_split::", "").replace("drba_conv::", "")
    if a.brief and gap < 0.02 and r["ms"] < 0.06:
        continue
    print(f"{names[s]} t={r['start_ms'] * 1e3:8.1f} dur={r['ms'] * 1e3:7.1f} gap={gap * 1e3:7.1f}  {nm[:58]} {r['label'] or ''}")
