#!/usr/bin/env python3
"""Per HSA queue: kernel-busy time and span over the last K steps of a rocpd kernel trace (a step ends with its 2nd to_out kernel):
which stream of a step carries the critical path.  python tools/exp/queue_busy.py results.db K"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
K = int(sys.argv[2])
rows = db.execute("select name, start, end, queue_id from kernels order by start").fetchall()
ends = [i for i, r in enumerate(rows) if "to_out_" in r[0]]
first = ends[-(2 * K) - 1] + 1
sel = rows[first:ends[-1] + 1]
t0, t1 = sel[0][1], max(r[2] for r in sel)
print(f"{K} steps: span {(t1 - t0) / 1e6 / K:.3f} ms per step, {len(sel) / K:.0f} launches per step")
for q in sorted(set(r[3] for r in sel)):
    rs = sorted((r[1], r[2]) for r in sel if r[3] == q)
    busy, cur_s, cur_e = 0, rs[0][0], rs[0][1]
    for s, e in rs[1:]:  # union of intervals (a queue's kernels may overlap each other)
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e in rs)
    gaps = sorted(((b[0] - a[1]) for a, b in zip(rs, rs[1:]) if b[0] > a[1]), reverse=True)
    print(f"queue {q}: {len(rs) / K:.0f} launches/step, kernel time {ksum / 1e6 / K:.3f} ms/step, busy (union) {busy / 1e6 / K:.3f} ms/step, "
          f"idle gaps > 20 us: {sum(1 for g in gaps if g > 20000) / K:.1f}/step totalling {sum(g for g in gaps if g > 20000) / 1e6 / K:.3f} ms/step")
