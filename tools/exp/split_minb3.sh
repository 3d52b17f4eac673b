#!/bin/bash
# Same-box A/B: the small two-term tiles (4 x 32 x 32 conv, the 32-cout transposed tile) capped at 168 registers for three
# workgroups per CU (DRBA_SPLIT_MINB3_F16=1; 40 / 80 bytes of scratch) against two per CU.
for rep in 1 2; do
  for v in minb2 minb3; do
    echo "##### $v"
    cp tools/exp/build/lib_$v.so drba_amd/csrc/libdrba_hip.so
    python tools/exp/split_per_cu.py 2>&1 | grep -v amdgpu.ids | grep -E "cfg21|cfg8"
  done
done
