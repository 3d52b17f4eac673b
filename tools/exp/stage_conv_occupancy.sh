#!/bin/bash
# Occupancy probe of the product stage_conv0: how much does the kernel lose with ONE workgroup per CU instead of two?
# (TUNING build of stage_conv.o in the GPU box's copy of the tree; DRBA_SC_LDS_PAD pads the LDS request.)
cd $(dirname $0)/../../drba_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DDRBA_TUNING_SWITCHES -c stage_conv.hip -o stage_conv.o && make > /dev/null 2>&1
cd ../..
for pad in 0 30000; do
  echo "== DRBA_SC_LDS_PAD=$pad"
  DRBA_SC_LDS_PAD=$pad python tools/stage_conv_check.py 20 2>&1 | grep -E "1080p B8 lazy (zero|smooth) flows, box|1080p B8 fused"
done
