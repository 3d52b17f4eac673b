#!/usr/bin/env python3
"""profiles/rocprof_frac.json from a steady-state kernel table (tools/rocpd_steady.py output of a rocprofv3 kernel trace of
`python bench.py`): {kernel symbol: {"avg_us", "calls_per_step"}} -- bench.py prices the symbol's algorithmic work per launch
against it and reports `roofline.frac_rocprof` next to the in-step `frac` and `frac_standalone`.
The table is only valid for the workload it was traced on: "__workload__" names bench.py's --config (default 1080p) and
bench.py emits frac_rocprof for that configuration only.
    python tools/rocprof_frac.py profiles/r04_v1_steady_state.csv [config]"""
import csv
import json
import os
import sys

src = sys.argv[1]
out = {}
with open(src) as f:
    for row in csv.DictReader(f):
        out[row["Name"]] = {"avg_us": float(row["AverageUs"]), "calls_per_step": float(row["CallsPerStep"]), "source": os.path.basename(src)}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (csrc_sha: the kernel sources this table was taken on; bench.py refuses a table of other sources)
out["__workload__"] = {"config": sys.argv[2] if len(sys.argv) > 2 else "1080p", "source": os.path.basename(src), "csrc_sha": bench.csrc_sha()}
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "rocprof_frac.json")
json.dump(out, open(dst, "w"), indent=1)
print(f"{len(out)} symbols -> {dst}")
