#!/bin/bash
# PMC passes over the fused stage kernels (stage_conv16 and stage_conv0, same launch) -> one line per kernel and counter set
#   tools/exp/stage_conv16_pmc.sh > gpurun_out/<tag>/sc16_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_RDRET_STALL_sum TCP_TOTAL_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH"; do
  i=$((i+1))
  rm -rf /tmp/scpmc$i
  timeout 120 rocprofv3 --pmc $set -d /tmp/scpmc$i -o p --output-format csv -- python $R/tools/exp/stage_conv16_pmc.py > /tmp/scpmc$i.log 2>&1
  f=$(find /tmp/scpmc$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
if not sys.argv[1]:
    print("no counter file"); sys.exit(0)
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "stage_conv" in r["Kernel_Name"]]
d = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"].split("<")[0].split("::")[-1], r["Dispatch_Id"])
    d.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    d[k]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    d[k]["_vgpr"] = r["VGPR_Count"]
last = {}
for (name, disp), v in d.items():
    last[name] = v
for name, v in last.items():
    print(name, {a: (round(b) if isinstance(b, float) else b) for a, b in v.items()})
PY
  tail -1 /tmp/scpmc$i.log | cut -c1-160
done
