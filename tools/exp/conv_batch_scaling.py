"""ResConv throughput vs batch size: separates per-launch fill/tail effects from per-tile efficiency."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (c, h, w) in ((64, 136, 240), (32, 272, 480), (96, 68, 120)):
    wt = torch.randn(c, c, 3, 3, generator=g) * 0.05
    for n in (1, 2, 4, 8, 16):
        x = torch.randn(n, c, h, w, generator=g).to(dev)
        layer = ops.Conv3x3(wt, torch.zeros(c), 1, True, torch.ones(1, c, 1, 1), device=dev)
        out = torch.empty_like(x)
        layer(x, residual=x, out=out)
        us = timeit(lambda: layer(x, residual=x, out=out))
        fl = 2.0 * c * c * 9 * h * w * n
        print(f"c={c} {h}x{w} N={n:2d}: {us:7.1f} us {fl / us / 1e6:6.1f} TF/s  cfg={ops._tuned.get(('conv3x3', n, c, c, h, w, 1))}")
