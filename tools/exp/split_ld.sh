#!/bin/bash
# Same-box A/B of the loader-wave form of the 4 x 32 x 64 two-term tile (DRBA_SPLIT_LD=1, conv_split.hip) against the
# four-wave form: phase clocks (tools/exp/build/csp / csp_ld), every two-term config per layer on the two tuning libraries
# (tools/exp/build/lib_ld0.so / lib_ld1.so), then the parity tests of the convolutions on the library in the tree.
for b in csp csp_ld; do
  echo "== $b"
  DRBA_PHASE_DIST=1 tools/exp/build/$b 6 8 64 136 240
  tools/exp/build/$b 6 1 64 576 960
  tools/exp/build/$b 6 1 128 288 480
done
cp drba_amd/csrc/libdrba_hip.so /tmp/lib_tree.so
for rep in 1 2; do
  for v in ld0 ld1; do
    echo "##### $v"
    cp tools/exp/build/lib_$v.so drba_amd/csrc/libdrba_hip.so
    python tools/exp/split_per_cu.py 2>&1 | grep -v amdgpu.ids | grep cfg22 | sed 's/cfg2[3-9] *[0-9.]*//g'
  done
done
cp /tmp/lib_tree.so drba_amd/csrc/libdrba_hip.so
python -m pytest tests/test_gpu_parity.py -q -x -k "conv_layers or families_agree" 2>&1 | tail -3
python -m pytest tests/test_gpu_fullsize.py -q -x -k "split_conv_configs" 2>&1 | tail -3
