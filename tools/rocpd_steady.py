#!/usr/bin/env python3
"""Steady-state per-step kernel table from a rocprofv3 rocpd database (kernel trace) of bench.py.
    python tools/rocpd_steady.py results.db K [out.csv] [--by-grid] [--by-queue] [--rows N]
Uses only the dispatches of the last K bench steps (a step ends with its 2nd to_out_kernel), so one-time work
(autotuning, warm-up, weight packing) is excluded.  --by-queue splits each symbol by HSA queue (the main stream vs the
lookahead stream); --by-grid by launch geometry (separates the layers that share an instantiation)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def main():
    argv = sys.argv[1:]
    by_grid = "--by-grid" in argv
    by_queue = "--by-queue" in argv
    nrows = 45
    if "--rows" in argv:
        nrows = int(argv[argv.index("--rows") + 1])
        del argv[argv.index("--rows"):argv.index("--rows") + 2]
    argv = [a for a in argv if not a.startswith("--")]
    db = sqlite3.connect(argv[0])
    K = int(argv[1])
    rows = db.execute("select name, start, end, queue_id, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
    ends = [i for i, r in enumerate(rows) if "to_out_" in r[0]]  # to_out_kernel / to_out_rows_kernel
    assert len(ends) >= 2 * K + 1, (len(ends), K)
    first = ends[-(2 * K) - 1] + 1
    sel = rows[first:ends[-1] + 1]
    t0, t1 = sel[0][1], max(r[2] for r in sel)
    agg, perq = {}, {}
    for n, s, e, q, gx, gy, gz, wx in sel:
        nm = short(n)
        if by_grid:
            nm += f" grid {gx // max(wx, 1)}x{gy}x{gz}"
        if by_queue:
            nm = f"q{q} " + nm
        a = agg.setdefault(nm, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += e - s
        a[2] = min(a[2], e - s)
        a[3] = max(a[3], e - s)
        pq = perq.setdefault(q, [0, 0])
        pq[0] += 1
        pq[1] += e - s
    busy = sum(a[1] for a in agg.values())
    # union of the busy intervals: time during which at least one kernel was executing
    iv = sorted((s, e) for _, s, e, *_ in sel)
    cover, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            cover += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    cover += ce - cs
    print(f"steady state: {K} steps, {len(sel)} dispatches ({len(sel) / K:.0f}/step), wall {(t1 - t0) / 1e6 / K:.3f} ms/step, "
          f"sum of kernel durations {busy / 1e6 / K:.3f} ms/step, GPU non-idle {cover / 1e6 / K:.3f} ms/step")
    for q, (c, ns) in sorted(perq.items()):
        print(f"  queue {q}: {c / K:.0f} launches/step, {ns / 1e6 / K:.3f} ms/step of kernel time")
    lines = ["Name,CallsPerStep,AverageUs,MsPerStep,Percentage,MinUs,MaxUs"]
    for nm, (c, ns, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"\"{nm}\",{c / K:.2f},{ns / c / 1e3:.1f},{ns / 1e6 / K:.3f},{100.0 * ns / busy:.1f},{mn / 1e3:.1f},{mx / 1e3:.1f}")
    print("\n".join(lines[:nrows]))
    if len(argv) > 2:
        open(argv[2], "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
