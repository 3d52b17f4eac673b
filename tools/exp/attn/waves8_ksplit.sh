#!/bin/bash
# the 8-wave window attention at the coarse 1080p shape for key splits 1..10 (the A/B library), twice
cd $(dirname $0)/../../..
cp tools/exp/build/libdrba_hip_tuning.so drba_amd/csrc/libdrba_hip.so
for round in 1 2; do for ks in 1 3 4 5 6 7 8 9 10; do
  echo -n "waves 8 ksplit $ks: "; DRBA_ATTN_WAVES=8 DRBA_ATTN_KSPLIT=$ks python tools/exp/attn/attn_target.py 8 1 30 2>/dev/null
done; done
