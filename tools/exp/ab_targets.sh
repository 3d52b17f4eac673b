#!/bin/bash
# Same-box A/B of two library builds on single-kernel targets: tools/exp/ab_targets.sh <other .so> "<target args>" ["<target args>" ...]
# each target = arguments of tools/exp/gather_target.py; alternates new / base twice.
OTHER=$1; shift
cp drba_amd/csrc/libdrba_hip.so /tmp/lib_new.so
for t in "$@"; do
  for round in 1 2; do
    for which in new base; do
      if [ $which = new ]; then cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so; else cp $OTHER drba_amd/csrc/libdrba_hip.so; fi
      echo -n "$which: "; python tools/exp/gather_target.py $t 2>/dev/null | tail -1
    done
  done
done
cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so
