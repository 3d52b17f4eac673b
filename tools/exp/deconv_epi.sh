#!/bin/bash
# Same-box A/B of the transposed convolution's epilogue (buffer stores, no branches, PixelShuffle selected once) against
# the build before it (tools/exp/build/lib_base.so): parity, per-layer timings, one-stream table rows.
python -m pytest tests/test_gpu_parity.py -q -x -k "conv_layers or families_agree" 2>&1 | tail -2
python -m pytest tests/test_gpu_fullsize.py -q -x -k "split_conv_configs" 2>&1 | tail -2
cp drba_amd/csrc/libdrba_hip.so /tmp/lib_tree.so
for rep in 1 2; do
  for v in base tree; do
    echo "##### $v"
    if [ $v = base ]; then cp tools/exp/build/lib_base.so drba_amd/csrc/libdrba_hip.so; else cp /tmp/lib_tree.so drba_amd/csrc/libdrba_hip.so; fi
    python tools/exp/split_per_cu.py 2>&1 | grep lastconv | cut -c1-110
  done
done
cp /tmp/lib_tree.so drba_amd/csrc/libdrba_hip.so
tools/exp/ab_table.sh tools/exp/build/lib_base.so "deconv"
