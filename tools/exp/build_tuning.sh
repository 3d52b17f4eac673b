#!/bin/bash
# The A/B library: the tree's sources built with -DDRBA_TUNING_SWITCHES (common.hpp env_int reads the DRBA_* variables) into
# tools/exp/build/libdrba_hip_tuning.so (git-ignored, travels to the GPU box).  A tools/exp script copies it over
# drba_amd/csrc/libdrba_hip.so ON THE BOX (nothing persists there); the product library never reads the environment.
#   tools/exp/build_tuning.sh [extra hipcc flags]
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
rm -rf /tmp/drba_tuning && mkdir -p /tmp/drba_tuning/drba_amd /tmp/drba_tuning/include
cp -r $ROOT/drba_amd/csrc /tmp/drba_tuning/drba_amd/ && cp $ROOT/include/drba_hip.h /tmp/drba_tuning/include/
cd /tmp/drba_tuning/drba_amd/csrc && rm -f *.o *.so
make -j8 TUNING=1 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -DDRBA_TUNING_SWITCHES $*" > /tmp/drba_tuning/build.log 2>&1 || { tail -20 /tmp/drba_tuning/build.log; exit 1; }
mkdir -p $ROOT/tools/exp/build && cp libdrba_hip.so $ROOT/tools/exp/build/libdrba_hip_tuning.so
ls -la $ROOT/tools/exp/build/libdrba_hip_tuning.so
