"""drba_local_corr_flow at GMFSS_UNION's 1080p fine-scale shape (and a ragged one): time per call and error against the fp64 oracle."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd import ops  # noqa: E402
from oracle import gmflow as ogm  # noqa: E402

dev = torch.device("cuda:0")
for (h, w) in ((13, 45), (37, 70), (144, 240)):
    torch.manual_seed(h)
    a, b = torch.randn(1, 128, h, w) * 0.6, torch.randn(1, 128, h, w) * 0.6
    ref = ogm.local_correlation_softmax(a.double(), b.double(), 4)
    ga, gb = a.to(dev), b.to(dev)
    got = ops.local_corr_flow(ga, gb, 4)
    err = float((got.cpu().double() - ref).abs().max())
    floor = float((ogm.local_correlation_softmax(a, b, 4).double() - ref).abs().max())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.local_corr_flow(ga, gb, 4)
    e1.record()
    torch.cuda.synchronize()
    print(f"local_corr_flow {h}x{w}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call, max|err| vs fp64 {err:.2e} (fp32 oracle {floor:.2e})")
