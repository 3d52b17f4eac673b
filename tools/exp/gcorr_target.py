"""drba_global_expect2 at GMFSS_UNION's 1080p coarse-scale shape (L = 72 x 120 tokens): time per call, checksum."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
L = 72 * 120
q, k = torch.randn(L, 128, device=dev) * 0.6, torch.randn(L, 128, device=dev) * 0.6
for _ in range(3):
    out = ops.global_expect2(q, k, None, 120, 128 ** 0.5)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = ops.global_expect2(q, k, None, 120, 128 ** 0.5)
e1.record()
torch.cuda.synchronize()
print(f"global_expect2 L={L}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call, checksum {float(out.double().abs().sum()):.4f}")
