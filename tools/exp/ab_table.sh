#!/bin/bash
# same-box A/B of two libraries on the one-stream per-kernel table
cd /root/repo
cp drba_amd/csrc/libdrba_hip.so /tmp/lib_new.so
for r in 1 2; do
 for w in new loop; do
  if [ $w = new ]; then cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so; else cp tools/exp/build/lib_loop_prologue.so drba_amd/csrc/libdrba_hip.so; fi
  echo "== $w $r"; python tools/step_timeline.py --one-stream --table 2>/dev/null | grep -E 'kernel time|warp_blend|ifblock_input_lds|stage_conv16' | cut -c1-150
 done
done
cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so
