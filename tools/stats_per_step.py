#!/usr/bin/env python3
"""Per-step view of a rocprofv3 kernel_stats.csv:  python tools/stats_per_step.py stats.csv NSTEPS"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU kernel time: {tot / 1e6:.1f} ms over {n:.0f} steps => {tot / 1e6 / n:.3f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    nm = re.sub(r"\((float|unsigned) const.*", "", nm).replace("void ", "")
    print(f"{nm:58s} calls={int(r['Calls']):4d} avg={float(r['AverageNs']) / 1e3:8.1f}us /step={float(r['TotalDurationNs']) / 1e6 / n:6.3f}ms {float(r['Percentage']):5.1f}%")
