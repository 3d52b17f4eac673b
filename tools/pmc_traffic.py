#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/pmc_targets.py -> profiles/pmc_traffic.json (+ a table).
    python tools/pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv gpurun_out/pmc_manifest.json
Counter unit: KiB.  gfx950 corrections (MI355X_MICROARCH.md 'HBM', re-checked by the calibration kernel in the same
run): FETCH_SIZE reports half of the bytes read (x2), WRITE_SIZE is exact.  traffic = 2*FETCH + WRITE per launch."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_target(path, manifest, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    segments, cur = [], None
    for r in rows:
        if "affine_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 256:  # segment marker
            cur = []
            segments.append(cur)
        elif cur is not None:
            cur.append(r)
    assert len(segments) == len(manifest), (len(segments), len(manifest))
    out = {}
    for m, seg in zip(manifest, segments):
        vals = [float(r["Counter_Value"]) for r in seg if m["symbol"] in r["Kernel_Name"]][-m["launches"]:]
        out[m["name"]] = sum(vals) / max(1, len(vals))
    return out


def main():
    fetch_csv, write_csv, man = sys.argv[1:4]
    manifest = json.load(open(man))
    f, w = per_target(fetch_csv, manifest, "FETCH_SIZE"), per_target(write_csv, manifest, "WRITE_SIZE")
    cal = [n for n in f if n.startswith("calibration")][0]
    print(f"calibration: FETCH_SIZE {f[cal]:.0f} KiB for 262144 KiB read (x{262144 / f[cal]:.3f}), WRITE_SIZE {w[cal]:.0f} KiB for 262144 KiB written (x{262144 / w[cal]:.3f})")
    traffic = {}
    print("| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | corrected traffic MB |")
    print("|---|---|---|---|")
    for n in f:
        if n == cal:
            continue
        traffic[n] = int((2.0 * f[n] + w[n]) * 1024)
        print(f"| {n} | {f[n]:.0f} | {w[n]:.0f} | {traffic[n] / 1e6:.1f} |")
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
