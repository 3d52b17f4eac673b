#!/usr/bin/env python3
"""Does the library's per-launch event trace perturb the step it measures?  bench.py's loop with the trace switched on for
runs of consecutive steps; average duration of two ResConv geometries per position inside a run, next to rocprofv3's."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from drba_amd import ops  # noqa: E402
from drba_amd.models.rife import RIFE  # noqa: E402
from drba_amd.models.utils import tools  # noqa: E402
from drba_amd.utils import synth  # noqa: E402

dev = torch.device("cuda:0")
model = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=1.0, device=dev)
clip = bench.DeviceClip(12, 1080, 1920, 1234, dev)
frames = [clip[k] for k in range(12)]
size = tools.get_valid_net_inp_size(np.zeros((1080, 1920, 3), np.uint8), 1.0, div=64)
src_size, dst_size = size["src_size"], size["dst_size"]
st = {"I0": ops.to_inp(frames[0], dst_size), "I1": ops.to_inp(frames[1], dst_size), "reuse": None, "k": 2}
TS = bench.TS


def step():
    I2 = st.pop("next", None)
    if I2 is None:
        I2 = ops.to_inp(frames[st["k"] % 12], dst_size)
    nxt = st.pop("next2", None)
    if nxt is None:
        nxt = ops.to_inp(frames[(st["k"] + 1) % 12], dst_size)
    st["next2"] = ops.to_inp(frames[(st["k"] + 2) % 12], dst_size)
    model.prefetch_frame(st["next2"])
    model.prefetch_pair(nxt, st["next2"])
    out, st["reuse"] = model.inference_ts_drba(st["I0"], st["I1"], I2, TS, st["reuse"], linear=True, lookahead=(nxt, TS))
    [ops.to_out(x, src_size) for x in out]
    st["I0"], st["I1"], st["next"] = st["I1"], I2, nxt
    st["k"] += 1


for _ in range(6):
    step()
RUN = 5
pos = {i: {} for i in range(RUN)}
for rep in range(4):
    for _ in range(6):
        step()
    for i in range(RUN):
        ops.trace_begin()
        step()
        torch.cuda.synchronize()
        for r in ops.trace_end():
            if r["label"] and "conv3x3 (14," in r["label"]:
                pos[i].setdefault(r["label"], []).append(r["ms"] * 1e3)
for i in range(RUN):
    print(f"traced step {i} of a run: " + "  ".join(f"{lab}: {np.mean(v):.1f} us (n={len(v)})" for lab, v in sorted(pos[i].items())))
