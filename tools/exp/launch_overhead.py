import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from drba_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
ops.AUTOTUNE = False
for (c, h, w, n, cfg) in ((192, 17, 30, 2, 7), (192, 17, 30, 2, 5), (192, 17, 30, 1, 7), (128, 34, 60, 2, 5), (192, 17, 30, 2, 2), (64, 136, 240, 2, 6)):
    x = torch.randn(n, c, h, w, generator=g).to(dev)
    layer = ops.Conv3x3(torch.randn(c, c, 3, 3, generator=g) * 0.05, torch.zeros(c), 1, True, torch.ones(1, c, 1, 1), device=dev, cfg=cfg)
    out = torch.empty_like(x)
    for _ in range(10): layer(x, residual=x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): layer(x, residual=x, out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"c={c} {h}x{w} N={n} cfg{cfg}: host {1e6*(t1-t0)/500:.1f} us/launch, total {1e6*(t2-t0)/500:.1f} us/launch")
