#!/bin/bash
# One GPU-box pass producing everything profiles/ holds for a round: tools/gpu_round.sh <tag> [skip-tests]
#   gpurun_out/<tag>/{pytest_gpu.txt, bench_*.json, kernel trace (rocpd db + csv), steady-state tables, PMC passes, timeline}
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
[ -z "$GRAFT_REPO_ROOT" ] && OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
cd $(dirname $0)/..
if [ "$2" != "skip-tests" ]; then
  rm -f $OUT/parity_report.txt
  DRBA_PARITY_REPORT=$OUT/parity_report.txt timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
  echo "pytest exit $?" >> $OUT/pytest_gpu.txt
  tail -3 $OUT/pytest_gpu.txt
fi
# PMC passes first: bench.py reads profiles/pmc_traffic.json (written here) for roofline.traffic of the same build
export TMPDIR=/tmp
REPO=$(pwd)
python tools/pmc_targets.py > /dev/null 2>&1   # autotune / warm outside the counter passes
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python $REPO/tools/pmc_targets.py > /dev/null 2> $OUT/pmc_fetch.err; echo "pmc fetch exit $?")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python $REPO/tools/pmc_targets.py > /dev/null 2> $OUT/pmc_write.err; echo "pmc write exit $?")
cp gpurun_out/pmc_manifest.json $OUT/ 2>/dev/null
# matrix-core utilisation of the same targets: its own pass (SQ counters only, no trace domains beside the kernel trace)
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_mfma -o m -- python $REPO/tools/pmc_targets.py > /dev/null 2> $OUT/pmc_mfma.err; echo "pmc mfma exit $?")
cp gpurun_out/pmc_manifest.json $OUT/pmc_manifest_mfma.json 2>/dev/null
M=$(ls $OUT/pmc_mfma/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$M" ] && (cd tools && python pmc_mfma.py $M $OUT/pmc_manifest_mfma.json $OUT/mfma_util.md > /dev/null)
rm -rf $OUT/pmc_mfma/*kernel_trace.csv
F=$(ls $OUT/pmc_fetch/*counter_collection.csv 2>/dev/null | head -1); Wc=$(ls $OUT/pmc_write/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$Wc" ]; then
  python tools/pmc_traffic.py $F $Wc $OUT/pmc_manifest.json $OUT/pmc_traffic.md && cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
fi
rm -rf $OUT/pmc_fetch/*kernel_trace.csv $OUT/pmc_write/*kernel_trace.csv
# the 4K scale-0.5 geometry (round 6): its own target set, merged into the same table
python tools/pmc_targets_4k.py > /dev/null 2>&1
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc4k_fetch -o f -- python $REPO/tools/pmc_targets_4k.py > /dev/null 2> $OUT/pmc4k_fetch.err; echo "pmc 4k fetch exit $?")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc4k_write -o w -- python $REPO/tools/pmc_targets_4k.py > /dev/null 2> $OUT/pmc4k_write.err; echo "pmc 4k write exit $?")
cp gpurun_out/pmc_manifest_4k.json $OUT/ 2>/dev/null
F4=$(ls $OUT/pmc4k_fetch/*counter_collection.csv 2>/dev/null | head -1); W4=$(ls $OUT/pmc4k_write/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$F4" ] && [ -n "$W4" ]; then
  DRBA_PMC_MERGE=1 python tools/pmc_traffic.py $F4 $W4 $OUT/pmc_manifest_4k.json $OUT/pmc_traffic_4k.md > /dev/null && cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
fi
rm -rf $OUT/pmc4k_fetch/*kernel_trace.csv $OUT/pmc4k_write/*kernel_trace.csv
timeout 900 python bench.py > $OUT/bench_1080p.json 2> $OUT/bench_1080p.err; echo "bench exit $?"
for c in 4k 480p 4k_s1; do
  timeout 600 python bench.py --config $c --no-extra --no-cpu-baseline > $OUT/bench_$c.json 2>> $OUT/bench_other.err
done
python tools/step_timeline.py --steps 2 > $OUT/timeline_1080p.txt 2>/dev/null
export TMPDIR=/tmp
REPO=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d $OUT/trace -o rife1080 -- python $REPO/bench.py --no-extra --no-cpu-baseline --no-roofline > $OUT/bench_1080p_under_rocprof.json 2> $OUT/rocprof.err; echo "rocprof exit $?")
ls $OUT/trace | head
DB=$(ls $OUT/trace/*_results.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then
  python tools/rocpd_steady.py $DB 12 $OUT/steady_state.csv --rows 14
  python tools/rocpd_steady.py $DB 12 $OUT/steady_state_by_grid.csv --by-grid --by-queue --rows 0 > /dev/null
  python tools/rocpd_stats.py $DB $OUT/kernel_stats_whole_run.csv > /dev/null
fi
du -sh $OUT
