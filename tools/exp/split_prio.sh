#!/bin/bash
# conv_split_mfma: does a fixed priority for one of the two co-resident workgroups of a CU (by wave slot parity) break
# the convoy (both in their MFMA phases at half rate, then both in their epilogues)?  csp / csp_prio1 (whole kernel) /
# csp_prio2 (MFMA phase only): tools/exp/conv_split_phases.hip built with -DDRBA_SPLIT_PRIO=0/1/2.
for rep in 1 2; do
for b in csp csp_prio1 csp_prio2; do
  echo "== $b"
  DRBA_PHASE_DIST=1 tools/exp/build/$b 6 8 64 136 240
  DRBA_PHASE_DIST=1 tools/exp/build/$b 9 8 96 68 120
  DRBA_PHASE_DIST=1 tools/exp/build/$b 5 8 32 272 480
  tools/exp/build/$b 6 1 64 576 960
  tools/exp/build/$b 0 8 32 272 480 20
done
done
