#!/bin/bash
# same-box A/B of two library builds on the per-layer convolution table and the 1080p loop:
#   tools/exp/ab_lib_conv.sh <old .so under tools/exp/build> (the new one is the tree's)
cd $(dirname $0)/../..
cp drba_amd/csrc/libdrba_hip.so /tmp/lib_new.so
for round in 1 2; do for v in old new; do
  if [ $v = old ]; then cp tools/exp/build/$1 drba_amd/csrc/libdrba_hip.so; else cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so; fi
  echo "== $v"
  [ $round = 1 ] && python tools/exp/split_per_cu.py 2>/dev/null | grep -v "^DRBA" | cut -c1-200
  python bench.py --no-cpu-baseline --no-extra --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('1080p', d['value'], d['ms_per_step'])"
done; done
cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so
