#!/bin/bash
# Same-box A/B: the next tile's first chunk requested from the epilogue (DRBA_SPLIT_DEFER=1) against under the last chunk's
# MFMAs (0).  Libraries: tools/exp/build_tuning.sh [-DDRBA_SPLIT_DEFER=0] -> tools/exp/build/lib_defer{0,1}.so; phase clocks:
# tools/exp/build/csp (defer 0) / csp_defer.
for rep in 1 2; do
  for v in 0 1; do
    echo "##### defer $v"
    cp tools/exp/build/lib_defer$v.so drba_amd/csrc/libdrba_hip.so
    python tools/exp/split_per_cu.py 2>&1 | grep -v amdgpu.ids
  done
done
for b in csp csp_defer; do
  echo "== $b"
  tools/exp/build/$b 6 8 64 136 240
  tools/exp/build/$b 9 8 96 68 120
  tools/exp/build/$b 5 8 32 272 480
  tools/exp/build/$b 0 8 32 272 480 20
  tools/exp/build/$b 1 8 64 136 240 52
done
