import sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from drba_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
cfg = [c for c in range(lib.drba_conv3x3_num_cfgs()) if lib.drba_conv3x3_cfg_family(c) == 3][0]
g = torch.Generator().manual_seed(1)
for (h, w) in ((2, 16), (2, 18)):
    # impulse tests: x = 1 at one pixel of channel 0, identity-like weights: only centre tap of (co=0, ci=0) = 1
    for tap in (4, 3, 5):
        x = torch.zeros(1, 64, h, w)
        for px in range(w):
            x[0, 0, 0, px] = px + 1
        wt = torch.zeros(64, 64, 3, 3)
        wt[0, 0, tap // 3, tap % 3] = 1.0
        b = torch.zeros(64)
        ref = F.conv2d(x, wt, b, padding=1)
        got = ops.Conv3x3(wt, b, 1, None, None, device=dev, cfg=cfg)(x.to(dev)).cpu()
        print(h, w, "tap", tap, "ref", [int(v) for v in ref[0, 0, 0]], "got", [round(float(v), 1) for v in got[0, 0, 0]])
