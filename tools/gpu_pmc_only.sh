#!/bin/bash
# The PMC passes of tools/gpu_round.sh alone (FETCH_SIZE / WRITE_SIZE over tools/pmc_targets.py -> pmc_traffic.{md,json}): tools/gpu_pmc_only.sh <tag>
TAG=${1:-r04}
cd $(dirname $0)/..
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python tools/pmc_targets.py > /dev/null 2>&1
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python $REPO/tools/pmc_targets.py > /dev/null 2> $OUT/pmc_fetch.err; echo "pmc fetch exit $?")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python $REPO/tools/pmc_targets.py > /dev/null 2> $OUT/pmc_write.err; echo "pmc write exit $?")
cp gpurun_out/pmc_manifest.json $OUT/ 2>/dev/null
F=$(ls $OUT/pmc_fetch/*counter_collection.csv 2>/dev/null | head -1); Wc=$(ls $OUT/pmc_write/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$Wc" ]; then
  python tools/pmc_traffic.py $F $Wc $OUT/pmc_manifest.json $OUT/pmc_traffic.md && cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
fi
rm -rf $OUT/pmc_fetch/*kernel_trace.csv $OUT/pmc_write/*kernel_trace.csv
grep -E "8 samples|head_fused" $OUT/pmc_traffic.md | cut -c1-70,160-270
