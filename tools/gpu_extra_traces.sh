#!/bin/bash
# rocprofv3 kernel traces of the workloads behind bench.py's extra_configs (the per-GPU work of config 5: 4K at scale
# 0.5; config 4: GMFSS_UNION 1080p) -> steady-state per-step kernel tables.  tools/gpu_extra_traces.sh <tag>
TAG=${1:-r04}
cd $(dirname $0)/..
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $OUT/trace_4k -o rife4k -- python $REPO/bench.py --config 4k --no-extra --no-cpu-baseline --no-roofline > $OUT/bench_4k_under_rocprof.json 2> $OUT/rocprof_4k.err; echo "rocprof 4k exit $?")
DB=$(ls $OUT/trace_4k/*_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_steady.py $DB 12 $OUT/steady_state_4k_s0.5.csv --rows 14
[ -n "$DB" ] && python tools/rocpd_steady.py $DB 12 $OUT/steady_state_4k_s0.5_by_grid.csv --by-grid --by-queue --rows 0 > /dev/null
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $OUT/trace_gmfss -o gmfss -- python $REPO/tools/gmfss_bench.py --steps 6 --warmup 3 > $OUT/gmfss_union_1080p_under_rocprof.json 2> $OUT/rocprof_gmfss.err; echo "rocprof gmfss exit $?")
DB=$(ls $OUT/trace_gmfss/*_results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_steady.py $DB 4 $OUT/steady_state_gmfss_union_1080p.csv --rows 14
rm -rf $OUT/trace_4k $OUT/trace_gmfss   # the databases are large; the tables are what is kept
du -sh $OUT
