#!/bin/bash
# Experiment builds of conv_dma.hip (wrong results on purpose, timing only): one library per -D switch under
# tools/exp/build/ (git-ignored, travels to the GPU box).  tools/exp/conv_dma_variants.sh "NOMFMA NOSPLIT NODMA NOSTORE ..."
# then on the GPU box: python tools/exp/conv_dma_time.py
cd $(dirname $0)/../../drba_amd/csrc
OBJS=$(ls *.o | grep -v conv_dma.o)
for v in ${1:-BASE NOMFMA NOSPLIT NODMA NOSTORE}; do
  D=""
  for f in $(echo $v | tr '+' ' '); do [ "$f" != "BASE" ] && D="$D -DDRBA_EXP_$f"; done
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics $D $EXTRA -c conv_dma.hip -o /tmp/conv_dma_$v.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/conv_dma_$v.o -o ../../tools/exp/build/libdrba_hip_$v.so && echo built $v
done
