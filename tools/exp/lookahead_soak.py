"""Soak: N warm steps at 1080p with and without the side-stream lookahead must give the same frames (races would show)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd import ops
from drba_amd.models.rife import RIFE
from drba_amd.utils import synth
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sd = synth.ifnet_state_dict(0)
raw = [torch.from_numpy(f).to(dev) for f in synth.make_clip(8, 1080, 1920, seed=5)]
ts = np.array([0.75, 1.25])
def run(look):
    m = RIFE(weights=sd, scale=1.0, device=dev)
    fr = lambda k: ops.resize_bilinear(ops.u8hwc_to_f32nchw(torch.roll(raw[k % 8], (k // 8) * 5, 1)), (1088, 1920))
    I0, I1, nxt, reuse, sums = fr(0), fr(1), None, None, []
    for k in range(N):
        I2 = nxt if nxt is not None else fr(k + 2)
        nxt = fr(k + 3) if look else None
        out, reuse = m.inference_ts_drba(I0, I1, I2, ts, reuse, True, lookahead=None if nxt is None else (nxt, ts))
        sums += [o.double().sum().item() for o in out] + [o[0, :, ::97, ::89].clone() for o in out]
        I0, I1 = I1, I2
    torch.cuda.synchronize()
    return sums
a, b = run(True), run(False)
bad = 0
for x, y in zip(a, b):
    d = abs(x - y) if isinstance(x, float) else float((x - y).abs().max())
    tol = 1e-2 if isinstance(x, float) else 1e-6
    if d > tol:
        bad += 1
        print("  mismatch", "sum" if isinstance(x, float) else "pixels", "diff %.3e" % d)
print("steps", N, "mismatches", bad, "of", len(a))
c = run(False)  # run-to-run variation of the inline path itself (atomic arrival order in the splats)
print("inline vs inline:", sum((abs(x - y) if isinstance(x, float) else float((x - y).abs().max())) > (1e-2 if isinstance(x, float) else 1e-6) for x, y in zip(b, c)))
