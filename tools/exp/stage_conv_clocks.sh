#!/bin/bash
# In-kernel clock profile of the round-4 EXPERIMENT kernel tools/exp/stage_conv_box/stage_conv_box.hip (stage_conv0 with the
# gathers' source boxes staged in LDS by LDS-DMA; not the product kernel): where a wave's time goes -- prologue, flow + bounding
# boxes, DMA issue, MFMAs, sampling, waits, barriers.  Builds it as stage_conv.o with -DDRBA_SC_CLOCKS (+ the tuning switches),
# relinks the library IN the box's copy of the tree (nothing persists) and launches the 8-sample 1080p geometry once per path.
#   tools/exp/stage_conv_clocks.sh > gpurun_out/<tag>/clocks.txt        (result of round 4: profiles/r04_stage_conv_box_clocks.txt)
cd $(dirname $0)/../../drba_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I. -DDRBA_TUNING_SWITCHES -DDRBA_SC_CLOCKS -c ../../tools/exp/stage_conv_box/stage_conv_box.hip -o stage_conv.o && make > /dev/null 2>&1
cd ../..
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from drba_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
H, W, B = 1088, 1920, 8
items = []
for _ in range(B):
    items.append((torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 1, H, W, generator=g).to(dev),
                  torch.randn(1, 16, H, W, generator=g).to(dev), torch.randn(1, 16, H, W, generator=g).to(dev)))
def mis(t):
    buf = torch.empty(t.numel() + 1, device=t.device); v = buf[1:].view(t.shape); v.copy_(t); return v
items_g = [(mis(a), b, c, d, e) for a, b, c, d, e in items]
terms = [(torch.zeros(B, 13, H // s, W // s, device=dev), float(s)) for s in (16, 8, 4)]
tprev = torch.randn(B, 13, H // 2, W // 2, generator=g).to(dev)
tprev[:, :4] = 0
conv = ops.Conv3x3(torch.randn(16, 52, 3, 3, generator=g) * 0.05, torch.zeros(16), 2, True, None, device=dev)
for name, its in (("box", items), ("gather", items_g)):
    for _ in range(2):
        ops.stage_conv0(its, None, tprev, 2.0, conv, terms=terms)
    torch.cuda.synchronize()
    print("----", name, flush=True)
PY
