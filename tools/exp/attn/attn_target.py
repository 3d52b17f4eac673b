"""One GMFSS_UNION 1080p window-attention shape, a few launches (PMC target / timing): attn_target.py <8|4> <shift 0|1> [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from drba_amd import ops  # noqa: E402

which, shift = sys.argv[1], bool(int(sys.argv[2]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
b, h, w, splits = {"8": (2, 72, 120, 2), "4": (2, 144, 240, 8)}[which]
dev = torch.device("cuda:0")
torch.manual_seed(0)
q, k, v = [torch.randn(b, h * w, 128, device=dev) for _ in range(3)]
for _ in range(3):
    ops.window_attention(q, k, v, h, w, splits, shift, 128 ** 0.5, terms=2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    out = ops.window_attention(q, k, v, h, w, splits, shift, 128 ** 0.5, terms=2)
e1.record()
torch.cuda.synchronize()
print(f"attn 1/{which} shift{int(shift)}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per call, checksum {float(out.double().sum()):.6f}")
