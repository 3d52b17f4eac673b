#!/bin/bash
# Same-box A/B of the epilogue specialised on the layer's operand kind (conv_split.hip) against the build before it
# (tools/exp/build/lib_base.so): phase clocks, per-layer timings of the two-term configs, parity, bench.
for b in csp csp_kind; do
  echo "== $b"
  tools/exp/build/$b 6 8 64 136 240
  tools/exp/build/$b 9 8 96 68 120
  tools/exp/build/$b 6 1 128 288 480
done
python -m pytest tests/test_gpu_parity.py -q -x -k "conv_layers or families_agree" 2>&1 | tail -2
python -m pytest tests/test_gpu_fullsize.py -q -x -k "split_conv_configs" 2>&1 | tail -2
cp drba_amd/csrc/libdrba_hip.so /tmp/lib_tree.so
for rep in 1 2; do
  for v in base tree; do
    echo "##### $v"
    if [ $v = base ]; then cp tools/exp/build/lib_base.so drba_amd/csrc/libdrba_hip.so; else cp /tmp/lib_tree.so drba_amd/csrc/libdrba_hip.so; fi
    python tools/exp/split_per_cu.py 2>&1 | grep -v amdgpu.ids | cut -c1-110
  done
done
cp /tmp/lib_tree.so drba_amd/csrc/libdrba_hip.so
tools/exp/ab_bench_libs.sh tools/exp/build/lib_base.so -- --no-cpu-baseline --no-roofline --no-extra
python tools/gmfss_bench.py 2>/dev/null | tail -1
cp tools/exp/build/lib_base.so drba_amd/csrc/libdrba_hip.so; python tools/gmfss_bench.py 2>/dev/null | tail -1
cp /tmp/lib_tree.so drba_amd/csrc/libdrba_hip.so; python tools/gmfss_bench.py 2>/dev/null | tail -1
