#!/bin/bash
# In-kernel clock profile of stage_conv16 (drba_amd/csrc/stage_conv16.hip): where a wave's time goes -- prologue, flow + taps, and per
# channel group: gather-finish + split + park, barrier, MFMAs, barrier.  Rebuilds stage_conv16.o with -DDRBA_SC16_CLOCKS IN the box's
# copy of the tree (nothing persists) and launches the 8-sample 1080p geometry (lazy flow, three terms) twice.
#   tools/exp/stage_conv16_clocks.sh [extra hipcc flags] > gpurun_out/<tag>/clocks.txt
cd $(dirname $0)/../../drba_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wno-unused-function -DDRBA_SC16_CLOCKS "$@" -c stage_conv16.hip -o stage_conv16.o && make > /dev/null 2>&1
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
def head(st, amp):
    t = torch.randn(B, 13, H // st, W // st, generator=g)
    lo = torch.randn(B, 4, max(H // st // 8, 2), max(W // st // 8, 2), generator=g) * amp
    t[:, :4] = torch.nn.functional.interpolate(lo, size=t.shape[2:], mode="bicubic", align_corners=False)
    return t.to(dev)
terms = [(head(16, 1.0), 16.0), (head(8, 0.4), 8.0), (head(4, 0.3), 4.0)]
tprev = head(2, 0.3)
conv = ops.Conv3x3(torch.randn(16, 52, 3, 3, generator=g) * 0.05, torch.zeros(16), 2, True, None, device=dev)
for _ in range(2):
    ops.stage_conv0(items, None, tprev, 2.0, conv, terms=terms)
    torch.cuda.synchronize()
    print("----", flush=True)
PY
