#!/usr/bin/env python3
"""Registers / scratch (spills) / occupancy of every kernel in one .hip file, from hipcc's resource-usage remarks.
    python tools/kernel_resources.py drba_amd/csrc/conv_split.hip [substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("(anonymous namespace)::", "")
        cur = re.sub(r"\(.*", "", cur).replace("void ", "").replace("drba_conv_split::", "").replace("drba_conv::", "")
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" ")[0]] = v
for n, r in rows.items():
    if flt in n:
        print(f"{n:70s} VGPR {str(r.get('VGPRs')):>4} AGPR {str(r.get('AGPRs')):>3} scratch {str(r.get('ScratchSize')):>4} occ {r.get('Occupancy')} lds {r.get('LDS')}")
