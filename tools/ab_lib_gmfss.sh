#!/bin/bash
# Same-box A/B of two builds of the library on the GMFSS_UNION 1080p step: tools/ab_lib_gmfss.sh <other libdrba_hip.so> <out dir>
OTHER=$1; OUT=$2
mkdir -p $OUT
cp drba_amd/csrc/libdrba_hip.so /tmp/lib_new.so
for round in 1 2; do
  for which in new base; do
    if [ $which = new ]; then cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so; else cp $OTHER drba_amd/csrc/libdrba_hip.so; fi
    echo -n "$which $round: "; python tools/gmfss_bench.py --steps 8 --warmup 4 2> $OUT/gmfss_${which}_$round.err | tee $OUT/gmfss_${which}_$round.json
  done
done
cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so
