#!/bin/bash
# same-box A/B: the gathers reading the frames' [H,W,4] copies against the planar frames, one-stream per-kernel table
cd $(dirname $0)/../..
for r in 1 2; do
 for v in "" "--no-img-x4"; do
  echo "== x4 ${v:-on} $r"; python tools/step_timeline.py --one-stream --table $v 2>/dev/null | grep -E 'kernel time|warp_blend|ifblock_input_lds|stage_conv16|conv3x3 \(10, 52, 32|to_inp' | cut -c1-150
 done
done
