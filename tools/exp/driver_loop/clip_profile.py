#!/usr/bin/env python3
"""Where does the host time of a driver-loop leg go?  (round 6: config 3 read 6 ms of host per step in some runs)
    python tools/exp/driver_loop/clip_profile.py [--pingpong 0|1] [--profile]
Runs bench.clip_leg for config 3 (1080p, -fps 60, scdet, one planted cut) and prints value / host time / path; with --profile
the timed iterations run under cProfile and the top entries by cumulative time are printed."""
import argparse
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--pingpong", type=int, default=1)
    p.add_argument("--profile", action="store_true")
    p.add_argument("--profile-rep", type=int, default=1)
    p.add_argument("--steps", type=int, default=40)
    p.add_argument("--size", default="1080p")
    p.add_argument("--fresh", action="store_true", help="a new model instance per repetition (the situation of bench.py's extra configs: the tuner's table is warm, the instance is not)")
    a = p.parse_args()
    import bench
    from drba_amd.models.rife import RIFE
    from drba_amd.utils import synth
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(warmup=bench.kClipWarmup, steps=a.steps, no_lookahead=False)
    n = args.warmup + args.steps + 3 + bench.kTraceWarm + bench.kTraceKeep + 1
    cut = args.warmup + args.steps // 2 + 2
    (h, w), scale = ((1080, 1920), 1.0) if a.size == "1080p" else ((2160, 3840), 0.5)
    m = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=scale, device=dev)
    for rep in range(2):
        if a.fresh and rep:
            m = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=scale, device=dev)
        clip = bench.DeviceClip(n, h, w, 1234, dev, cut_at=cut, pingpong=bool(a.pingpong))
        prof = cProfile.Profile() if (a.profile and rep == a.profile_rep) else None
        if prof:
            prof.enable()
        r = bench.clip_leg(m, clip, 60.0, -1, True, args, "probe")
        if prof:
            prof.disable()
        print(f"rep {rep}: {r['value']:.1f} frames/s  {r['ms_per_step']:.3f} ms/step  host {r['host_ms_per_step']:.3f}  generated {r['frames_generated']}  path {r['path']}")
        if prof:
            pstats.Stats(prof).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
