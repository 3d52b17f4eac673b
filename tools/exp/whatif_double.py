#!/usr/bin/env python3
"""Sensitivity probe (wrong work on purpose, timing only): run ONE part of the step twice and see how much the step grows.
A part on the critical path costs its whole duration (or more, under contention); a part with slack costs little.
    python tools/exp/whatif_double.py <base|block0|block1|block2|block3|block4|encode|gather3|gather4> [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from drba_amd import ops  # noqa: E402
from drba_amd.models.rife_426_heavy import IFNet_HDv3 as M  # noqa: E402

what = sys.argv[1]
sys.argv = [sys.argv[0]] + sys.argv[2:]
if what.startswith("block"):
    idx = int(what[5:])
    orig_load = M.IFNet.load_state_dict

    def load(self, sd, strict=False):
        r = orig_load(self, sd, strict)
        blk = self.block[idx]
        core = blk.core

        def twice(x):
            core(x)
            return core(x)
        blk.core = twice
        return r
    M.IFNet.load_state_dict = load
elif what == "encode":
    orig = M.Head.__call__

    def twice(self, x, feat=False):
        orig(self, x, feat)
        return orig(self, x, feat)
    M.Head.__call__ = twice
    M.Head.forward = twice
elif what in ("gather3", "gather4"):
    s_target = 2.0 if what == "gather3" else 1.0
    orig = ops.ifblock_input_lds

    def twice(img0, img1, f0, f1, timestep, flow, tmp_prev, prev_scale, scale, out=None, fold=False):
        if scale == s_target:
            orig(img0, img1, f0, f1, timestep, flow, tmp_prev, prev_scale, scale, out=out, fold=fold)
        return orig(img0, img1, f0, f1, timestep, flow, tmp_prev, prev_scale, scale, out=out, fold=fold)
    ops.ifblock_input_lds = twice
import bench  # noqa: E402
bench.main()
