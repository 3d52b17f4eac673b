#!/bin/bash
# Same-box A/B of two builds of the library on the one-stream per-kernel table (every kernel's own duration at the loop's launch
# geometry): tools/exp/ab_table.sh <other libdrba_hip.so> [grep pattern]      (nothing persists on the GPU box)
cd $(dirname $0)/../..
OTHER=${1:-tools/exp/build/lib_base.so}; PAT=${2:-.}
cp drba_amd/csrc/libdrba_hip.so /tmp/lib_new.so
for r in 1 2; do
 for w in new base; do
  if [ $w = new ]; then cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so; else cp $OTHER drba_amd/csrc/libdrba_hip.so; fi
  echo "== $w $r"; python tools/step_timeline.py --one-stream --table 2>/dev/null | grep -E "kernel time|$PAT" | cut -c1-160
 done
done
cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so
