#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/pmc_targets.py -> profiles/pmc_traffic.json (+ a table).
    python tools/pmc_traffic.py FETCH.csv|FETCH.db WRITE.csv|WRITE.db gpurun_out/pmc_manifest.json [table.md]
Counter unit: KiB.  gfx950 corrections (MI355X_MICROARCH.md 'HBM', re-checked by the calibration kernel in the same
run): FETCH_SIZE reports half of the bytes read (x2), WRITE_SIZE is exact.  traffic = 2*FETCH + WRITE per launch.
Output: {kernel symbol: {launch label: bytes per launch}} -- the keys bench.py's roofline object uses."""
import csv
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def rows_of(path, counter):
    """[(dispatch id, kernel name, grid size, value)] for one counter, from a counter_collection CSV or a rocpd database."""
    if path.endswith(".db"):
        db = sqlite3.connect(path)
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        q = "select dispatch_id, kernel_name, grid_size_x, counter_name, value from counters_collection" if "grid_size_x" in cols else \
            "select dispatch_id, kernel_name, grid_size, counter_name, value from counters_collection"
        acc = {}
        for d, k, gsz, c, v in db.execute(q):
            if c == counter:
                key = (int(d), k, int(gsz))
                acc[key] = acc.get(key, 0.0) + float(v)
        return sorted((d, k, gsz, v) for (d, k, gsz), v in acc.items())
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    return sorted((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])) for r in rows)


def per_target(path, manifest, counter):
    segments, cur = [], None
    for _, kname, gsz, val in rows_of(path, counter):
        if "affine_kernel" in kname and gsz == 256:  # segment marker
            cur = []
            segments.append(cur)
        elif cur is not None:
            cur.append((kname, val))
    assert len(segments) == len(manifest), (len(segments), len(manifest))
    out = []
    for m, seg in zip(manifest, segments):
        vals = [v for k, v in seg if short(k) == m["symbol"]][:m["launches"]]  # the reps follow the marker directly; what comes
        # after them in the segment is the NEXT target's warm-up / autotune (possibly the same symbol on another layer)
        # (a latency-bound layer can get another configuration from the autotuner in the profiled run: no figure then)
        out.append(sum(vals) / len(vals) if vals else None)
    return out


def main():
    fetch_p, write_p, man = sys.argv[1:4]
    manifest = json.load(open(man))
    f, w = per_target(fetch_p, manifest, "FETCH_SIZE"), per_target(write_p, manifest, "WRITE_SIZE")
    lines = [f"calibration: FETCH_SIZE {f[0]:.0f} KiB for 262144 KiB read (x{262144 / f[0]:.3f}), WRITE_SIZE {w[0]:.0f} KiB for "
             f"262144 KiB written (x{262144 / w[0]:.3f})", "",
             "| target | kernel | launch | FETCH_SIZE KiB | WRITE_SIZE KiB | corrected traffic MB | algorithmic MB | ratio |", "|---|---|---|---|---|---|---|---|"]
    traffic = {}
    for m, fv, wv in zip(manifest[1:], f[1:], w[1:]):
        if fv is None or wv is None:
            lines.append(f"| {m['name']} | `{m['symbol']}` | {m['label']} | - | - | (another configuration ran in the profiled pass) | | |")
            continue
        t = int((2.0 * fv + wv) * 1024)
        traffic.setdefault(m["symbol"], {})[m["label"]] = t
        alg = m["algorithmic"] if m["unit"] == "byte" else None
        lines.append(f"| {m['name']} | `{m['symbol']}` | {m['label']} | {fv:.0f} | {wv:.0f} | {t / 1e6:.1f} | "
                     f"{'' if alg is None else f'{alg / 1e6:.1f}'} | {'' if alg is None else f'{t / alg:.2f}'} |")
    print("\n".join(lines))
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.environ.get("DRBA_PMC_MERGE") == "1" and os.path.exists(path):  # a second target set (tools/pmc_targets_4k.py): add to the table
        old = json.load(open(path))
        for sym, labels in traffic.items():
            old.setdefault(sym, {}).update(labels)
        traffic = old
    json.dump(traffic, open(path, "w"), indent=1)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
