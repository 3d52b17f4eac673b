#!/bin/bash
# tools/exp/cu_mask/sweep.sh <out dir>: the CU-partition sweep of round 6 (one GPU box, same process settings per line)
OUT=${1:-gpurun_out/cu_mask}
mkdir -p $OUT
cd $(dirname $0)/../../..
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/exp/cu_mask/census.hip -o /tmp/census 2> /dev/null && /tmp/census > $OUT/census.txt 2>&1
cat $OUT/census.txt
timeout 600 python tools/graph_probe.py --config 1080p > $OUT/graph_probe_1080p.txt 2> $OUT/graph_probe_1080p.err; cat $OUT/graph_probe_1080p.txt; tail -3 $OUT/graph_probe_1080p.err
B="--no-extra --no-cpu-baseline --no-roofline"
run() {  # name, ab_bench flags
  n=$1; shift
  timeout 400 python tools/ab_bench.py "$@" -- $B > $OUT/$n.json 2> $OUT/$n.err
  python - "$OUT/$n.json" "$n" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:44s} {r['value']:8.1f} frames/s  {r['ms_per_step']:.3f} ms/step  host {r.get('host_ms_per_step')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run base
run side4_pref4_shared --cu-mask side=28:4,prefetch=28:4
run side4_pref4_main28 --cu-mask side=28:4,prefetch=28:4,main=0:28
run side8_pref8_shared --cu-mask side=24:8,prefetch=24:8
run side8_pref8_main24 --cu-mask side=24:8,prefetch=24:8,main=0:24
run side4_pref4_main24_3way --cu-mask side=28:4,prefetch=24:4,main=0:24
run side2_pref2_main28 --cu-mask side=30:2,prefetch=28:2,main=0:28
run side8_shared_only --cu-mask side=24:8
run base_again
run one_stream --no-lookahead
