#!/usr/bin/env python3
"""What would stacking the frames of TWO consecutive steps (4 interpolations) in one IFNet pass buy?  Times
IFNet.forward_pairs with 2 and with 4 items at 1088x1920 on one stream (every kernel's own time, no overlap), whole pass and
the low-resolution stages 0-2 / the full-resolution stages 3-4 separately.   python tools/exp/batch4_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd.models.rife import RIFE  # noqa: E402
from drba_amd.utils import synth  # noqa: E402

dev = torch.device("cuda:0")
m = RIFE(weights=synth.ifnet_state_dict(seed=0), scale=1.0, device=dev)
H, W = 1088, 1920
g = torch.Generator().manual_seed(0)
I = [torch.rand(1, 3, H, W, generator=g).to(dev) for _ in range(6)]
f = [m.ifnet.encode(x) for x in I]
tm = [torch.rand(1, 1, H, W, generator=g).to(dev) for _ in range(8)]
items4 = [(I[1], I[0], tm[0], f[1], f[0]), (I[1], I[2], tm[1], f[1], f[2]), (I[2], I[1], tm[2], f[2], f[1]), (I[2], I[3], tm[3], f[2], f[3])]
items8 = items4 + [(I[3], I[2], tm[4], f[3], f[2]), (I[3], I[4], tm[5], f[3], f[4]), (I[4], I[3], tm[6], f[4], f[3]), (I[4], I[5], tm[7], f[4], f[5])]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, first, last in (("all stages", 0, 5), ("stages 0-2", 0, 3)):
    t2 = timeit(lambda: m.ifnet.forward_pairs(items4[:2], m.scale_list, first, last))
    t4 = timeit(lambda: m.ifnet.forward_pairs(items4, m.scale_list, first, last))
    t8 = timeit(lambda: m.ifnet.forward_pairs(items8, m.scale_list, first, last))
    print(f"{name}: 2 items {t2:.3f} ms ({t2 / 2:.3f} per frame), 4 items {t4:.3f} ms ({t4 / 4:.3f} per frame): x{t4 / t2:.2f}, 8 items {t8:.3f} ms ({t8 / 8:.3f} per frame)")
st2 = m.ifnet.forward_pairs(items4[:2], m.scale_list, 0, 3)
st4 = m.ifnet.forward_pairs(items4, m.scale_list, 0, 3)
t2 = timeit(lambda: m.ifnet.forward_pairs(items4[:2], m.scale_list, 3, 5, st2))
t4 = timeit(lambda: m.ifnet.forward_pairs(items4, m.scale_list, 3, 5, st4))
st8 = m.ifnet.forward_pairs(items8, m.scale_list, 0, 3)
t8 = timeit(lambda: m.ifnet.forward_pairs(items8, m.scale_list, 3, 5, st8))
print(f"stages 3-4: 2 items {t2:.3f} ms, 4 items {t4:.3f} ms: x{t4 / t2:.2f}, 8 items {t8:.3f} ms ({t8 / 8:.3f} per frame vs {t2 / 2:.3f})")
