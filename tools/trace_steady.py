#!/usr/bin/env python3
"""Steady-state per-step kernel table from a rocprofv3 kernel_trace.csv of bench.py.
    python tools/trace_steady.py kernel_trace.csv K [out.csv [rows]] [--by-grid]
Uses only the dispatches of the last K bench steps (a step ends with its 2nd f32_to_u8_kernel),
so one-time work (autotuning, warm-up, weight packing) is excluded."""
import csv
import re
import sys

BY_GRID = "--by-grid" in sys.argv
if BY_GRID:
    sys.argv.remove("--by-grid")
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
ends = [i for i, n in enumerate(names) if "f32_to_u8_kernel" in n]
assert len(ends) >= 2 * K + 1, (len(ends), K)
first = ends[-(2 * K) - 1] + 1
sel = rows[first:ends[-1] + 1]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
agg = {}
for r in sel:
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    nm = re.sub(r"\((float|unsigned) const.*", "", nm).replace("void ", "")
    if BY_GRID:  # one row per launch geometry: separates the layers that share a kernel instantiation
        nm += f" grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}"
    a = agg.setdefault(nm, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(a[1] for a in agg.values())
print(f"steady state: {K} steps, {len(sel)} dispatches ({len(sel) / K:.0f}/step), wall {(t1 - t0) / 1e6 / K:.3f} ms/step, "
      f"kernel-busy {busy / 1e6 / K:.3f} ms/step")
lines = ["Name,CallsPerStep,AverageUs,MsPerStep,Percentage"]
for nm, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"\"{nm}\",{c / K:.1f},{ns / c / 1e3:.1f},{ns / 1e6 / K:.3f},{100.0 * ns / busy:.1f}")
print("\n".join(lines[:int(sys.argv[4]) if len(sys.argv) > 4 else 40]))
if len(sys.argv) > 3 and sys.argv[3] != "-":
    open(sys.argv[3], "w").write("\n".join(lines) + "\n")
