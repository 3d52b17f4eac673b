#!/usr/bin/env python3
"""How coherent are the warps of the bench workload?  For the per-stage running flows of one 1080p interpolation
(seeded weights, synthetic clip): |flow| percentiles, and the fraction of pixels whose bilinear tap origin (x0, y0) is
exactly one column to the right of the left neighbour's (same row) -- the case in which a lane could take its left taps
from the neighbouring lane instead of the texture-address path."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from drba_amd.models.rife import RIFE  # noqa: E402
from drba_amd.utils import synth  # noqa: E402

dev = torch.device("cuda:0")
H, W = 1088, 1920
m = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=1.0, device=dev)
clip = synth.make_clip(3, 1080, 1920, seed=1234)
from drba_amd import ops  # noqa: E402
fr = [ops.to_inp(torch.from_numpy(f).to(dev), (H, W)) for f in clip]
net = m.ifnet
_, flows = net.forward_pair(fr[0], fr[1], 0.5, m.scale_list, want_flows=True)
xs = torch.arange(W, device=dev).float()[None, :]
ys = torch.arange(H, device=dev).float()[:, None]
for i, fl in enumerate(flows):
    fl = fl[0]
    mag = torch.sqrt(fl[0] ** 2 + fl[1] ** 2).flatten()
    q = torch.quantile(mag[:: 97].float(), torch.tensor([0.5, 0.9, 0.99], device=dev)).tolist()
    out = [f"stage {i}: |flow01| p50 {q[0]:.2f} p90 {q[1]:.2f} p99 {q[2]:.2f} max {float(mag.max()):.1f}"]
    for k in (0, 2):
        x0 = torch.floor((xs + fl[k]).clamp(0, W - 1))
        y0 = torch.floor((ys + fl[k + 1]).clamp(0, H - 1))
        same = ((x0[:, 1:] == x0[:, :-1] + 1) & (y0[:, 1:] == y0[:, :-1])).float().mean().item()
        samey = ((y0[1:, :] == y0[:-1, :] + 1) & (x0[1:, :] == x0[:-1, :])).float().mean().item()
        dx = (fl[k][:, 1:] - fl[k][:, :-1]).abs().mean().item()
        # spread of the tap origins inside a 32 x 8 tile, relative to the tile's mean displacement
        t = fl[k: k + 2].reshape(2, H // 8, 8, W // 32, 32)
        dev_ = (t - t.mean(dim=(2, 4), keepdim=True)).abs().amax(dim=(0, 2, 4)).flatten()
        qd = torch.quantile(dev_, torch.tensor([0.5, 0.9, 0.99], device=dev)).tolist()
        out.append(f"  flow[{k}:{k+2}]: right-neighbour tap shared {same:.3f}, lower-neighbour {samey:.3f}, mean |d flow/dx| {dx:.3f}, "
                   f"max deviation from the 32x8 tile mean p50 {qd[0]:.2f} p90 {qd[1]:.2f} p99 {qd[2]:.2f}")
    print("\n".join(out))
