#!/bin/bash
# Same-box A/B of the double-buffered staging (one barrier per chunk) of the two-term 4 x 32 tiles against the build before it
# (tools/exp/build/lib_base.so): phase clocks (csp = DRBA_SPLIT_DB=0, csp_db), parity, per-layer timings, bench, GMFSS_UNION.
for b in csp csp_db; do
  echo "== $b"
  tools/exp/build/$b 6 8 64 136 240
  tools/exp/build/$b 6 1 128 288 480
  tools/exp/build/$b 0 8 32 272 480 20
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
for v in tree base tree base; do
  if [ $v = base ]; then cp tools/exp/build/lib_base.so drba_amd/csrc/libdrba_hip.so; else cp /tmp/lib_tree.so drba_amd/csrc/libdrba_hip.so; fi
  echo -n "gmfss $v: "; python tools/gmfss_bench.py 2>/dev/null | tail -1 | cut -c1-140
done
cp /tmp/lib_tree.so drba_amd/csrc/libdrba_hip.so
