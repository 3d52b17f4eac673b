#!/bin/bash
# PMC passes over head_fused (tools/exp/head_time.py launches it) -> gpurun_out/hf/head_pmc_*.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/hf
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum"; do
  i=$((i+1))
  rm -rf /tmp/hpmc$i
  timeout 300 rocprofv3 --pmc $set -d /tmp/hpmc$i -o p --output-format csv -- python $R/tools/exp/head_time.py 3 > /tmp/hpmc$i.log 2>&1
  f=$(find /tmp/hpmc$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY' | tee $R/gpurun_out/hf/head_pmc_$i.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "head_fused" in r["Kernel_Name"]]
d = collections.OrderedDict()
for r in rows:
    e = d.setdefault(r["Dispatch_Id"], {"grid": r["Grid_Size"], "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "vgpr": r["VGPR_Count"], "lds": r["LDS_Block_Size"]})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in list(d.items())[2:4]:
    print(k, {a: (round(b) if isinstance(b, float) else b) for a, b in v.items()})
PY
done
