#!/bin/bash
# Experiment builds of stage_conv.hip: one library per set of -D switches under tools/exp/build/ (git-ignored, travels to
# the GPU box).  tools/exp/stage_conv_variants.sh "BASE DEPTH=2 DEPTH=2+WPE=5 EXP_NOMFMA ..." ; then on the GPU box:
# python tools/exp/stage_conv_time.py
cd $(dirname $0)/../../drba_amd/csrc
mkdir -p ../../tools/exp/build
rm -f ../../tools/exp/build/libdrba_hip_*.so
OBJS=$(ls *.o | grep -v stage_conv.o)
for v in ${1:-BASE}; do
  D=""
  for f in $(echo $v | tr '+' ' '); do [ "$f" != "BASE" ] && D="$D -DDRBA_SC_$f"; done
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $D $EXTRA -c stage_conv.hip -o /tmp/stage_conv_$v.o 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/stage_conv_$v.o -o ../../tools/exp/build/libdrba_hip_$v.so && echo built $v
done
