#!/bin/bash
# window attention at the two GMFSS_UNION 1080p shapes for every key split 1..9 (the A/B library: DRBA_ATTN_KSPLIT)
cd $(dirname $0)/../../..
cp tools/exp/build/libdrba_hip_tuning.so drba_amd/csrc/libdrba_hip.so
for which in 8 4; do for ks in 1 2 3 4 5 6 7 8 9 10 12; do
  echo -n "ksplit $ks: "; DRBA_ATTN_KSPLIT=$ks python tools/exp/attn/attn_target.py $which 1 20
done; done
