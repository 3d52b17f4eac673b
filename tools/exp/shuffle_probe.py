"""GridNet's tail at 1080p: conv 64 -> 256 at 576 x 960 + PixelShuffle(2) as two kernels against drba_conv3x3_shuffle, per accepting
configuration (time per call)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator().manual_seed(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


n, cin, cout, h, w = 1, 64, 256, 576, 960
x = torch.randn(n, cin, h, w, generator=g).to(dev)
wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
b = torch.randn(cout, generator=g) * 0.1
layer = ops.Conv3x3(wt, b, 1, None, None, device=dev)
out = torch.empty((n, cout // 4, 2 * h, 2 * w), device=dev)
plain = torch.empty((n, cout, h, w), device=dev)
print(f"pixel_shuffle2 alone: {timeit(lambda: ops.pixel_shuffle2(plain)):.1f} us")
for cfg in range(lib.drba_conv3x3_num_cfgs()):
    if lib.drba_conv3x3_cfg_family(cfg) != 4 or lib.drba_conv3x3_cfg_stride(cfg) != 1 or lib.drba_conv3x3_packed_floats(cin, cout, cfg) == 0:
        continue
    wp = layer._pack(cfg)
    conv = lambda: lib.drba_conv3x3(ops._p(x), ops._p(wp), ops._p(layer.bias), None, None, None, ops._p(plain), n, cin, h, w, cout, 1, 0, 0.0, 0, 0.0, cfg, ops._stream())  # noqa: E731
    shuf = lambda: lib.drba_conv3x3_shuffle(ops._p(x), ops._p(wp), ops._p(layer.bias), ops._p(out), n, cin, h, w, cout, 0, 0.0, cfg, ops._stream())  # noqa: E731
    if conv() != 0:
        continue
    t_conv = timeit(conv)
    t_shuf = timeit(shuf) if shuf() == 0 else float("nan")
    print(f"cfg{cfg}: conv {t_conv:.1f} us, conv + shuffle store {t_shuf:.1f} us", flush=True)
