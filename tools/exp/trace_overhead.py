#!/usr/bin/env python3
"""Host and GPU cost of a bench step with the library's kernel trace on vs off (diagnostic for bench.py's roofline leg)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from drba_amd import ops
from drba_amd.models.rife import RIFE
from drba_amd.utils import synth
dev = torch.device("cuda:0")
m = RIFE(weights=synth.ifnet_state_dict(0), scale=1.0, device=dev)
clip = bench.DeviceClip(12, 1080, 1920, 1234, dev)
fr = [ops.to_inp(clip[k], (1088, 1920)) for k in range(12)]
TS = np.array([0.75, 1.25])
reuse = None
def step(k, look=True):
    global reuse
    out, reuse = m.inference_ts_drba(fr[k % 10], fr[(k + 1) % 10], fr[(k + 2) % 10], TS, reuse, True,
                                     lookahead=(fr[(k + 3) % 10], TS) if look else None)
    return out
for k in range(5):
    step(k)
torch.cuda.synchronize()
ops.trace_begin(); ops.trace_pause()
for label, traced in (("plain", False), ("traced#1", True), ("plain", False), ("traced#2", True), ("traced#3", True), ("plain", False)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if traced: ops.trace_resume()
    step(7)
    if traced: ops.trace_pause()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label:10s} host {1e3*(t1-t0):7.2f} ms   total {1e3*(t2-t0):7.2f} ms", flush=True)
t0 = time.perf_counter()
recs = ops.trace_end()
print("trace_end (name lookup + event reads)", round(1e3 * (time.perf_counter() - t0), 2), "ms for", len(recs), "records")
