#!/usr/bin/env python3
"""Where does the one-off ~40 ms host stall of a traced bench run sit?  Runs the bench step with a given schedule
(string of p = plain step, t = traced step) and prints the host time of every step.
    python tools/exp/trace_stall.py ppptpppppppppppppptpppppppp [--sync-before-traced]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from drba_amd import ops
from drba_amd.models.rife import RIFE
from drba_amd.utils import synth
sched = sys.argv[1]
sync_before = "--sync-before-traced" in sys.argv
dev = torch.device("cuda:0")
m = RIFE(weights=synth.ifnet_state_dict(0), scale=1.0, device=dev)
clip = bench.DeviceClip(12, 1080, 1920, 1234, dev)
fr = [ops.to_inp(clip[k], (1088, 1920)) for k in range(12)]
TS = np.array([0.75, 1.25])
reuse = None
def step(k):
    global reuse
    out, reuse = m.inference_ts_drba(fr[k % 10], fr[(k + 1) % 10], fr[(k + 2) % 10], TS, reuse, True,
                                     lookahead=(fr[(k + 3) % 10], TS))
    return [ops.to_out(x, (1080, 1920)) for x in out]
for k in range(4):
    step(k)
torch.cuda.synchronize()
mode = os.environ.get("STALL_MODE", "begin")
if mode == "begin":
    ops.trace_begin(); ops.trace_pause()
elif mode == "begin_sleep":
    ops.trace_begin(); ops.trace_pause(); time.sleep(0.3)
elif mode == "begin_c_only":  # events created, python-side TRACE stays None
    from drba_amd import _lib
    _lib.load().drba_trace_begin(); _lib.load().drba_trace_end()
elif mode == "begin_gcfreeze":
    import gc
    ops.trace_begin(); ops.trace_pause(); gc.collect(); gc.freeze()
elif mode == "begin_gcoff":
    import gc
    ops.trace_begin(); ops.trace_pause(); gc.collect(); gc.disable()
elif mode == "none":
    assert "t" not in sched
elif mode == "sleep_only":
    time.sleep(0.3)
print("mode", mode)
times = []
T0 = time.perf_counter()
for k, c in enumerate(sched):
    if c == "t" and sync_before:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    if c == "t": ops.trace_resume()
    step(4 + k)
    if c == "t": ops.trace_pause()
    times.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
total = (time.perf_counter() - T0) * 1e3
print(sched)
print(" ".join(f"{c}{t:.1f}" for c, t in zip(sched, times)))
print(f"total {total:.1f} ms for {len(sched)} steps = {total / len(sched):.2f} ms/step")
ops.trace_end()
