#!/bin/bash
# the two-term window attention at 4 / 6 / 8 waves per workgroup (the A/B library: DRBA_ATTN_WAVES, DRBA_ATTN_KSPLIT)
cd $(dirname $0)/../../..
cp tools/exp/build/libdrba_hip_tuning.so drba_amd/csrc/libdrba_hip.so
for which in 8 4; do for wv in 4 6 8; do for ks in ${KS:-0}; do
  echo -n "waves $wv ksplit $ks: "; DRBA_ATTN_WAVES=$wv DRBA_ATTN_KSPLIT=$ks python tools/exp/attn/attn_target.py $which 1 20
done; done; done
