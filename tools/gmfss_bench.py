#!/usr/bin/env python3
"""GMFSS_UNION / GMFSS steady-state throughput on one MI355X (not the headline metric; DESIGN.md quotes it).

    python tools/gmfss_bench.py [--model gmfss_union|gmfss] [--size 4k|1080p|720p|480p] [--scale S] [--steps K] [--warmup W]

Step = to_inp + warm inference_ts_drba(I0, I1, I2, ts=[0.75, 1.25], reuse, linear=True) + to_out, as bench.py."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import ops  # noqa: E402
from drba_amd.models.utils import tools  # noqa: E402
from drba_amd.utils import synth  # noqa: E402

SIZES = {"4k": (2160, 3840), "1080p": (1080, 1920), "720p": (720, 1280), "480p": (480, 854)}


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--model", default="gmfss_union", choices=("gmfss_union", "gmfss"))
    p.add_argument("--size", default="1080p", choices=sorted(SIZES))
    p.add_argument("--scale", type=float, default=1.0)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--no-lookahead", action="store_true")
    p.add_argument("--table", action="store_true",
                   help="after the timed steps: 2 more steps on ONE stream with every launch traced -> per-symbol kernel time per step")
    p.add_argument("--batch-fusion", action="store_true", help="A/B: one GridNet pass over the frames of a step instead of one per frame")
    p.add_argument("--swap-copies", action="store_true", help="A/B: splats into temporaries + a full swap_select pass (Model.SWAP_IN_PLACE = False)")
    a = p.parse_args()
    if a.batch_fusion:
        from drba_amd.models.model_gmfss_union.GMFSS import Model
        Model.BATCH_FUSION = True
    if a.swap_copies:
        from drba_amd.models.model_gmfss_union.GMFSS import Model
        Model.SWAP_IN_PLACE = False
    dev = torch.device("cuda", 0)
    sds = synth.gmfss_union_state_dicts(0)
    if a.model == "gmfss_union":
        from drba_amd.models.gmfss_union import GMFSS_UNION
        model = GMFSS_UNION(weights=sds, scale=a.scale, device=dev)
    else:
        from drba_amd.models.gmfss import GMFSS
        sds["fusion"] = synth.seeded_state_dict(synth.gridnet_shapes(12, "head"), 0, "grid12.")
        model = GMFSS(weights=sds, scale=a.scale, device=dev)
    H, W = SIZES[a.size]
    size = tools.get_valid_net_inp_size(np.zeros((H, W, 3), np.uint8), model.scale, div=model.pad_size)
    src_size, dst_size = size["src_size"], size["dst_size"]
    frames = [torch.from_numpy(f).to(dev) for f in synth.make_clip(6, H, W, seed=7)]
    to_inp = lambda k: ops.to_inp(frames[k % 6], dst_size)  # noqa: E731
    I0, I1, reuse, k = to_inp(0), to_inp(1), None, 2
    ts = np.array([0.75, 1.25])
    sink = []
    t0 = nxt = None
    for it in range(a.warmup + a.steps):
        if it == a.warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        I2 = nxt if nxt is not None else to_inp(k)
        nxt = None if a.no_lookahead else to_inp(k + 1)  # the driver reads one frame ahead (drba_amd/infer.py)
        out, reuse = model.inference_ts_drba(I0, I1, I2, ts, reuse, True, lookahead=nxt)
        sink = [ops.to_out(x, src_size) for x in out]
        I0, I1, k = I1, I2, k + 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if a.table:
        from drba_amd.models import lookahead as la
        la.ONE_STREAM = True
        model._look = None
        for traced in (False, True):  # one untraced step settles the one-stream state
            if traced:
                ops.trace_begin()
            for _ in range(2):
                I2 = to_inp(k)
                out, reuse = model.inference_ts_drba(I0, I1, I2, ts, reuse, True, lookahead=None)
                sink = [ops.to_out(x, src_size) for x in out]
                I0, I1, k = I1, I2, k + 1
            torch.cuda.synchronize()
        recs = ops.trace_end()
        agg = {}
        for r in recs:
            v = agg.setdefault(r["name"][:90], [0, 0.0])
            v[0] += 1
            v[1] += r["ms"]
        tot = sum(v[1] for v in agg.values())
        print(f"one stream, 2 traced steps: {len(recs) / 2:.0f} launches and {tot / 2:.2f} ms of kernel time per step", file=sys.stderr)
        for nm, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            print(f"{ms / 2:8.3f} ms/step {100 * ms / tot:5.1f} %  {n / 2:6.1f} x {ms / n * 1e3:8.1f} us  {nm}", file=sys.stderr)
    print(json.dumps({"model": a.model, "size": a.size, "net_size": list(dst_size), "scale": a.scale,
                      "frames_per_s": round(2 * a.steps / dt, 3), "ms_per_step": round(dt / a.steps * 1e3, 2),
                      "steps": a.steps, "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
