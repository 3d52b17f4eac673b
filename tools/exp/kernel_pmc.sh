#!/bin/bash
# PMC passes over ONE kernel of a target script -> one line per counter set (the last dispatch of the kernel in each pass)
#   tools/exp/kernel_pmc.sh <kernel name substring> <python script> [script args...]
KERN=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rm -rf /tmp/kpmc$i
  timeout 300 rocprofv3 --pmc $set -d /tmp/kpmc$i -o p --output-format csv -- python "$@" > /tmp/kpmc$i.log 2>&1
  f=$(find /tmp/kpmc$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$KERN" <<'PY'
import csv, sys, collections
if not sys.argv[1]:
    print("no counter file"); sys.exit(0)
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
d = collections.OrderedDict()
for r in rows:
    k = r["Dispatch_Id"]
    d.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    d[k]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    d[k]["_vgpr"] = r["VGPR_Count"]
    d[k]["_lds"] = r.get("LDS_Block_Size", "")
    d[k]["_grid"] = r["Grid_Size"]
if d:
    v = list(d.values())[-1]
    print(sys.argv[2], {a: (round(b) if isinstance(b, float) else b) for a, b in v.items()})
PY
done
