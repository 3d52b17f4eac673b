import os, sys
sys.path.insert(0, "/root/repo")
import torch
from drba_amd.models.rife import RIFE
from drba_amd.models.rife_426_heavy import IFNet_HDv3 as M
mode = sys.argv[1]; sys.argv = [sys.argv[0]] + sys.argv[2:]
if mode == "nohead":
    # WHAT-IF (wrong results, timing only): the encoder of the new frame costs nothing (a cached feature tensor is
    # returned, with its pair-interleaved copy already attached)
    cache = {}
    orig = M.Head.__call__
    def fake(self, x, feat=False):
        k = tuple(x.shape)
        if k not in cache:
            cache[k] = orig(self, x, feat)
            from drba_amd import ops
            ops.pair_interleaved(cache[k])
        return cache[k]
    M.Head.__call__ = fake
    M.Head.forward = fake
import bench
bench.main()
