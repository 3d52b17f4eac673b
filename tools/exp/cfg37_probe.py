import os, sys, torch
sys.path.insert(0, "/root/repo")
from drba_amd import _lib, ops
dev = torch.device("cuda:0"); lib = _lib.load()
g = torch.Generator().manual_seed(0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, c, h, w) in ((1, 96, 288, 480), (1, 96, 288, 480), (2, 96, 288, 480), (1, 96, 144, 240), (1, 96, 288, 512)):
    x = torch.randn(n, c, h, w, generator=g).to(dev)
    wt = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    b = torch.randn(c, generator=g) * 0.1
    beta = torch.rand(1, c, 1, 1, generator=g) + 0.5
    out = torch.empty_like(x)
    for cfg in (37, 33):
        layer = ops.Conv3x3(wt, b, 1, True, beta, device=dev, cfg=cfg)
        t = timeit(lambda: layer(x, residual=x, out=out))
        ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x.double(), wt.double().to(dev), b.double().to(dev), padding=1) * beta.double().to(dev) + x.double(), 0.2)
        print(f"cfg{cfg} [{n},{c},{h},{w}]: {t:.1f} us, max err {float((out.double() - ref).abs().max()):.2e}", flush=True)
