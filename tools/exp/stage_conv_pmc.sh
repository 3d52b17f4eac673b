#!/bin/bash
# PMC passes over the fused stage kernel -> gpurun_out/sc/pmc_*.txt (per-dispatch counter values of stage_conv0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sc
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAVES" \
           "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/scpmc$i
  timeout 300 rocprofv3 --pmc $set -d /tmp/scpmc$i -o p --output-format csv -- python $R/tools/exp/stage_conv_pmc.py > /tmp/scpmc$i.log 2>&1
  f=$(find /tmp/scpmc$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY' > $R/gpurun_out/sc/pmc_$i.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "stage_conv0" in r["Kernel_Name"]]
d = collections.OrderedDict()
for r in rows:
    d.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    d[r["Dispatch_Id"]]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    d[r["Dispatch_Id"]]["_vgpr"] = r["VGPR_Count"]; d[r["Dispatch_Id"]]["_lds"] = r["LDS_Block_Size"]
for k, v in d.items():
    print(k, {a: (round(b) if isinstance(b, float) else b) for a, b in v.items()})
PY
  tail -2 /tmp/scpmc$i.log | cut -c1-200
  cat $R/gpurun_out/sc/pmc_$i.txt | tail -1
done
