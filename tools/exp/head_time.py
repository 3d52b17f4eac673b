#!/usr/bin/env python3
"""IFNet's encoder alone: one fused kernel (head_fused.hip) against the four layers + the pair-interleave copy.
    python tools/exp/head_time.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd import ops  # noqa: E402
from drba_amd.models.rife_426_heavy.IFNet_HDv3 import Head  # noqa: E402
from drba_amd.utils import synth  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
head = Head(synth.ifnet_state_dict(0), "encode.", dev)
for (H, W) in ((1088, 1920), (2176, 3840)):
    x = torch.rand(1, 3, H, W).to(dev)
    for fused, two, planar in ((True, True, False), (True, False, False), (True, True, True), (True, False, True), (False, False, True)):
        ops.HEAD_FUSED, ops.HEAD_TWO_TERM = fused, two  # two: head_fused16.hip (two fp16 terms) / head_fused.hip (fp32 MFMA)
        for _ in range(3):
            ops.pair_interleaved(head(x, planar=planar))
        torch.cuda.synchronize()
        ops.trace_begin()
        for _ in range(reps):
            ops.pair_interleaved(head(x, planar=planar))
        recs = ops.trace_end()
        per = {}
        for r in recs:
            per[r["name"][:60]] = per.get(r["name"][:60], 0.0) + r["ms"] * 1e3 / reps
        print(f"{H}x{W} fused={fused} two_term={two} planar={planar}: {sum(per.values()):7.1f} us per frame  " + "  ".join(f"{k.split('<')[0].split('::')[-1]} {v:.0f}" for k, v in per.items()), flush=True)
