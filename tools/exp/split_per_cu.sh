#!/bin/bash
# Same-box A/B of conv_split_mfma's persistent grid (3 workgroups per CU by LDS against the resident count) on the tuning
# library, plus the phase clocks of the 64-channel tile per workgroup.  tools/exp/build_tuning.sh and the csp build first.
cp tools/exp/build/libdrba_hip_tuning.so drba_amd/csrc/libdrba_hip.so
for v in 3 2 3 2; do DRBA_SPLIT_PER_CU=$v python tools/exp/split_per_cu.py; done
python tools/exp/split_per_cu.py
for v in 3 2; do
  echo "== phases, 64 ch 136x240 N8, cfg 6, per CU $v"; DRBA_PHASE_DIST=1 DRBA_SPLIT_PER_CU=$v tools/exp/build/csp 6 8 64 136 240
  echo "== phases, 96 ch 68x120 N8, cfg 9, per CU $v"; DRBA_PHASE_DIST=1 DRBA_SPLIT_PER_CU=$v tools/exp/build/csp 9 8 96 68 120
done
