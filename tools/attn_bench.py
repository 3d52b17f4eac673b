"""Fused window attention vs the unfused formulation (BLAS QK^T + masked softmax kernel + BLAS PV + roll/split copies)
at the two GMFSS_UNION 1080p shapes.  python tools/attn_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drba_amd import ops  # noqa: E402
from drba_amd.models.gmflow.gmflow import GMFlow  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    net = GMFlow.__new__(GMFlow)
    net._mask, net.device = {}, dev
    for b, h, w, splits in ((2, 72, 120, 2), (2, 144, 240, 8)):
        q, k, v = [torch.randn(b, h * w, 128, device=dev) for _ in range(3)]
        L = (h // splits) * (w // splits)
        flops = 4.0 * b * splits * splits * L * L * 128
        for shift in (False, True):
            f = timeit(lambda: ops.window_attention(q, k, v, h, w, splits, shift, 128 ** 0.5, terms=3))
            f2 = timeit(lambda: ops.window_attention(q, k, v, h, w, splits, shift, 128 ** 0.5, terms=2))
            d2 = (ops.window_attention(q, k, v, h, w, splits, shift, 128 ** 0.5, terms=2) - ops.window_attention(q, k, v, h, w, splits, shift, 128 ** 0.5, terms=3)).abs().max()
            print(f"b{b} {h}x{w} splits{splits} shift{int(shift)} L={L}: fp32 MFMA {f * 1e3:.0f} us ({flops / f / 1e9:.1f} TFLOP/s)  "
                  f"two-term fp16 {f2 * 1e3:.0f} us ({flops / f2 / 1e9:.1f} TFLOP/s)  max|two-term - fp32| {float(d2):.2e}", flush=True)

if __name__ == "__main__":
    main()
