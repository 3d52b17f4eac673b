#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_lib.sh <other libdrba_hip.so> <out dir> [bench flags]
# Runs bench.py (1080p, no extras) alternately with the tree's library and with <other> copied in its place, twice each
# (boxes differ by +-3 %: only same-box pairs say anything about a kernel change).  Nothing persists on the GPU box.
OTHER=$1; OUT=$2; shift 2
mkdir -p $OUT
cp drba_amd/csrc/libdrba_hip.so /tmp/lib_new.so
for round in 1 2; do
  for which in new base; do
    if [ $which = new ]; then cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so; else cp $OTHER drba_amd/csrc/libdrba_hip.so; fi
    timeout 600 python bench.py --no-extra --no-cpu-baseline "$@" > $OUT/bench_${which}_$round.json 2> $OUT/bench_${which}_$round.err
    python - <<PY
import json
d = json.load(open("$OUT/bench_${which}_$round.json"))
print("$which $round:", d["value"], "frames/s", d["ms_per_step"], "ms/step")
PY
  done
done
cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so
