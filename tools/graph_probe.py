#!/usr/bin/env python3
"""What would replaying a group of DRBA steps as ONE HIP graph buy?  (round 6)

    python tools/graph_probe.py [--config 1080p|4k] [--reps 20] [--group 4]

One group of RIFE.GROUP warm steps at the bench's geometry -- to_inp of the 4 new frames, the stacked IFNet pass over the
group's 8 frames (encoders, batched coarse flows + reversal, the group's DRM maps, 5 stages, blend), to_out of the 8 frames --
on ONE stream, timed three ways on the same process:
  eager     the calls as the library issues them (Python + ctypes per launch)
  graph     the same launch sequence captured once (torch.cuda.CUDAGraph over hipGraph) and replayed
  3-stream  what bench.py measures (printed by bench.py itself: run it next to this tool)
The graph reads the group's frames from fixed uint8 buffers and its entering `reuse` from fixed tensors (a timing probe:
every replay computes the same group), so the numbers are the launch path's, not a product feature."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--config", default="1080p")
    p.add_argument("--reps", type=int, default=20)
    p.add_argument("--group", type=int, default=4)
    a = p.parse_args()
    import bench
    from drba_amd import ops
    from drba_amd.models import lookahead
    from drba_amd.models.rife import RIFE
    from drba_amd.models.utils import tools
    from drba_amd.utils import synth
    lookahead.ONE_STREAM = True
    (sh, sw), scale, _ = bench.CONFIGS[a.config]
    dev = torch.device("cuda", 0)
    RIFE.GROUP = a.group
    g = a.group
    model = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=scale, device=dev)
    size = tools.get_valid_net_inp_size(np.zeros((sh, sw, 3), np.uint8), scale, div=model.pad_size)
    dst = size["dst_size"]
    u8 = [torch.from_numpy(f).to(dev) for f in bench.make_frames_u8(g + 2, sh, sw, seed=1234)]
    ts_list = [bench.TS.copy() for _ in range(g)]
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        F01 = [ops.to_inp(u8[0], dst), ops.to_inp(u8[1], dst)]
        reuse = model.warm_reuse(F01[0], F01[1])

        def group():
            F = F01 + [ops.to_inp(u8[2 + j], dst) for j in range(g)]
            model._group_out = ()
            out, _ = model._drba_group(F, ts_list, reuse, None)
            frames = list(out)
            for po in model._group_out:
                frames += po[5]
            model._group_out = ()
            return [ops.to_out(x, (sh, sw)) for x in frames]

        for _ in range(3):  # autotune, workspaces, the stream's conv counters
            outs = group()
        stream.synchronize()
        n_frames = len(outs)

        def timed(fn, reps):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            host = time.perf_counter() - t0
            e1.synchronize()
            return e0.elapsed_time(e1) / reps, host * 1e3 / reps

        eager_ms, eager_host = timed(group, a.reps)
        ref = [o.clone() for o in group()]
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        ops.conv_state_reset()
        stream.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            ops.conv_state_reset()
            gouts = group()
        stream.synchronize()
        for _ in range(3):
            graph.replay()
        stream.synchronize()
        same = all(torch.equal(x, y) for x, y in zip(gouts, ref))
        max_lsb = max(int((x.int() - y.int()).abs().max()) for x, y in zip(gouts, ref))
        graph_ms, graph_host = timed(graph.replay, a.reps)
    steps = g
    print(f"config {a.config} net {dst} group of {g} steps = {n_frames} frames, one stream")
    print(f"  eager : {eager_ms:8.3f} ms per group = {eager_ms / steps:6.3f} ms/step  {1e3 * n_frames / eager_ms:8.1f} frames/s   host {eager_host:6.3f} ms per group")
    print(f"  graph : {graph_ms:8.3f} ms per group = {graph_ms / steps:6.3f} ms/step  {1e3 * n_frames / graph_ms:8.1f} frames/s   host {graph_host:6.3f} ms per replay")
    print(f"  graph outputs == eager outputs: {same} (max difference {max_lsb} LSB)")


if __name__ == "__main__":
    main()
