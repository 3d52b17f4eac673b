#!/bin/bash
# same-box A/B of window-attention builds: tools/exp/attn/ab_attn.sh <lib A> <lib B> ... (paths under tools/exp/build), two rounds
cd $(dirname $0)/../../..
cp drba_amd/csrc/libdrba_hip.so /tmp/lib_keep.so
for round in 1 2; do for lib in "$@"; do
  cp tools/exp/build/$lib drba_amd/csrc/libdrba_hip.so
  for which in 8 4; do echo -n "$lib: "; python tools/exp/attn/attn_target.py $which 1 30; done
done; done
cp /tmp/lib_keep.so drba_amd/csrc/libdrba_hip.so
