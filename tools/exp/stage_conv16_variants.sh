#!/bin/bash
# stage_conv16's workgroup shapes on the 8-sample 1080p launch (tools/stage_conv16_check.py --time-only) with the A/B library
# (tools/exp/build_tuning.sh) copied over the product library ON THE BOX:  tools/exp/stage_conv16_variants.sh > gpurun_out/<tag>/variants.txt
cd $(dirname $0)/../..
cp tools/exp/build/libdrba_hip_tuning.so drba_amd/csrc/libdrba_hip.so || exit 1
for toh in 7 5 3 7; do
  echo "==== DRBA_SC16_TOH=$toh"
  DRBA_SC16_TOH=$toh python tools/stage_conv16_check.py 20 --time-only 2>&1 | grep -E 'smooth|zero'
done
