#!/usr/bin/env python3
"""The CPU oracle's 1080p warm DRBA step (bench.py's cpu_baseline workload) against the thread count, on the GPU box's
host: which `--cpu-threads` is the fair (fastest) CPU baseline.  Every setting runs in its own process with a time limit
(oneDNN / OpenMP oversubscription makes the 256-thread setting ~100x slower than the best one).
    python tools/cpu_threads.py [limit_seconds] > profiles/rNN_cpu_threads.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, numpy as np, torch
sys.path.insert(0, %r)
import oracle
from drba_amd.utils import synth
from drba_amd.models.utils.tools import get_valid_net_inp_size
n = int(sys.argv[1]); torch.set_num_threads(n)
H, W = 1080, 1920
dst = get_valid_net_inp_size(np.zeros((H, W, 3), np.uint8), 1.0, div=64)["dst_size"]
ora = oracle.rife.RifeOracle(synth.ifnet_state_dict(seed=0), 1.0)
fr = [oracle.ops.resize(torch.from_numpy(f.transpose(2, 0, 1)).unsqueeze(0).float() / 255.0, dst) for f in synth.make_clip(4, H, W, seed=1234)]
ts = np.array([0.75, 1.25])
with torch.no_grad():
    f12, f21, a, b = ora.calc_flow(fr[0], fr[1]); reuse = (f21, f12, b, a)
    oracle.ifnet.ifnet(ora.sd, torch.cat((fr[0], fr[1]), 1), 0.5, ora.scale_list, f0=a, f1=b)   # oneDNN primitives exist
    t0 = time.perf_counter()
    out, _ = ora.inference_ts_drba(fr[0], fr[1], fr[2], ts, reuse, True)
    dt = time.perf_counter() - t0
print(f"{n:4d} threads: {len(out) / dt:8.4f} frames/s ({dt:6.1f} s per step)")
''' % ROOT
limit = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
avail = len(os.sched_getaffinity(0))
print(f"host: {avail} hardware threads available; torch CPU fp32 oracle, one warm inference_ts_drba step at 1088x1920 (2 frames)")
for n in [t for t in (1, 8, 16, 32, 64, 128, 256) if t <= avail]:
    try:
        r = subprocess.run([sys.executable, "-c", CHILD, str(n)], capture_output=True, text=True, timeout=limit)
        print(r.stdout.strip() or f"{n:4d} threads: failed ({r.stderr.strip().splitlines()[-1] if r.stderr.strip() else 'no output'})", flush=True)
    except subprocess.TimeoutExpired:
        print(f"{n:4d} threads: > {limit:.0f} s (stopped; < {2 / limit:.4f} frames/s)", flush=True)
