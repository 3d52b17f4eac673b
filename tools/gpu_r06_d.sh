#!/bin/bash
OUT=gpurun_out/r06_d
mkdir -p $OUT
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_1080p.json 2> $OUT/bench_1080p.err; echo bench exit $?
python - <<PY
import json
r=json.loads(open("$OUT/bench_1080p.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["host_ms_per_step"])
for k,v in r.get("extra_configs",{}).items():
    print(k, v["value"], v["ms_per_step"], v.get("host_ms_per_step"), v.get("frames_generated"), v.get("path"))
PY
B="--no-extra --no-cpu-baseline --no-roofline"
for n in base split100 base2 split100b; do
  f=""; case $n in split*) f="--chain-split 100";; esac
  timeout 400 python tools/ab_bench.py $f -- $B > $OUT/ab_$n.json 2> $OUT/ab_$n.err
  python -c "
import json,sys
r=json.loads(open('$OUT/ab_$n.json').read().strip().splitlines()[-1]); print('$n', r['value'], r['ms_per_step'])"
done
timeout 600 python bench.py --config 4k --no-extra --no-cpu-baseline > $OUT/bench_4k.json 2> $OUT/bench_4k.err
python -c "
import json
r=json.loads(open('$OUT/bench_4k.json').read().strip().splitlines()[-1]); print('4k', r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['frac'])"
# 4K PMC passes
export TMPDIR=/tmp
REPO=$(pwd)
python tools/pmc_targets_4k.py > /dev/null 2> $OUT/pmc4k_warm.err
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/pmc4k_fetch -o f -- python $REPO/tools/pmc_targets_4k.py > /dev/null 2> $REPO/$OUT/pmc4k_fetch.err; echo "pmc fetch exit $?")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$OUT/pmc4k_write -o w -- python $REPO/tools/pmc_targets_4k.py > /dev/null 2> $REPO/$OUT/pmc4k_write.err; echo "pmc write exit $?")
cp gpurun_out/pmc_manifest_4k.json $OUT/
F=$(ls $OUT/pmc4k_fetch/*counter_collection.csv 2>/dev/null | head -1); Wc=$(ls $OUT/pmc4k_write/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$F" ] && [ -n "$Wc" ]; then
  DRBA_PMC_MERGE=1 python tools/pmc_traffic.py $F $Wc $OUT/pmc_manifest_4k.json $OUT/pmc_traffic_4k.md && cp profiles/pmc_traffic.json $OUT/pmc_traffic.json
fi
rm -rf $OUT/pmc4k_fetch/*kernel_trace.csv $OUT/pmc4k_write/*kernel_trace.csv
du -sh $OUT
