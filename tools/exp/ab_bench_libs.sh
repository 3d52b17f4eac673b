#!/bin/bash
# Same-box A/B of two builds of the library on bench.py's headline loop: tools/exp/ab_bench_libs.sh <other .so> [bench args]
cd $(dirname $0)/../..
OTHER=${1:-tools/exp/build/lib_base.so}; shift
cp drba_amd/csrc/libdrba_hip.so /tmp/lib_new.so
for r in 1 2 3; do
 for w in new base; do
  if [ $w = new ]; then cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so; else cp $OTHER drba_amd/csrc/libdrba_hip.so; fi
  echo -n "$w $r: "; python tools/ab_bench.py "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:(v.get('value') if isinstance(v,dict) else v) for k,v in (d.get('extra_configs') or {}).items()})"
 done
done
cp /tmp/lib_new.so drba_amd/csrc/libdrba_hip.so
